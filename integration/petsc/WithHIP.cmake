# WithHIP.cmake -- the OpenCMISS CMake build's switch for the MI355X (gfx950) AIJ / Vec types on libpetschip.
#
# Goes beside option(WITH_CUDA) / if (WITH_CUDA) of the reference's CMakeLists.txt:8,36-45 :
#     include(WithHIP)          # after include(OpenCMISSPETScConfig), CMakeLists.txt:17 (it appends to PETSCCONF_HAVE_FLAGS,
#                               # which :21-22 write into petscconf.h through templates/petscconf.h.in)
# and, where the GPU sources are registered (vec CMakeLists.txt:606-611, mat :1072-1079):
#     if (PETSC_HAVE_HIP)
#       list (APPEND PETSCMAT_SRCS ${PETSCHIP_GLUE_SOURCES})
#     endif ()
# The glue is plain C and libpetschip is prebuilt by hipcc, so -- unlike CUDA (CMakeLists.txt:457-458,639-640,1084-1085:
# cuda_add_library) -- the libraries stay ordinary add_library targets; the new dependency is one shared library.
option(WITH_HIP "${PROJECT_NAME} - Use HIP (MI355X / gfx950 types MATAIJHIP, VECHIP on libpetschip)" OFF)
set(PETSCHIP_DIR "" CACHE PATH "Root of the libpetschip installation (include/petschip.h, lib/libpetschip.so)")
set(PETSC_HIP_ARCH "gfx950" CACHE STRING "GPU architecture libpetschip was built for")

if (WITH_HIP)
    if (NOT PETSC_HIP_ARCH STREQUAL "gfx950")
        message(FATAL_ERROR "WITH_HIP: libpetschip is built for gfx950 only (PETSC_HIP_ARCH=${PETSC_HIP_ARCH})")
    endif ()
    # the C ABI of libpetschip fixes PetscScalar = double, real, and PetscInt = int32_t (include/petschip.h)
    if (PETSC_USE_COMPLEX OR PETSC_USE_REAL_SINGLE OR PETSC_USE_REAL___FLOAT128 OR PETSC_USE_64BIT_INDICES)
        message(FATAL_ERROR "WITH_HIP needs real double-precision scalars and 32-bit indices")
    endif ()
    find_path(PETSCHIP_INCLUDE_DIR petschip.h HINTS ${PETSCHIP_DIR} ENV PETSCHIP_DIR PATH_SUFFIXES include)
    find_library(PETSCHIP_LIBRARY petschip HINTS ${PETSCHIP_DIR} ENV PETSCHIP_DIR PATH_SUFFIXES lib lib64 petsc_amd/lib)
    find_path(PETSCHIP_GLUE_DIR petschip_glue.c HINTS ${PETSCHIP_DIR} ENV PETSCHIP_DIR PATH_SUFFIXES integration/petsc share/petschip)
    if (PETSCHIP_INCLUDE_DIR AND PETSCHIP_LIBRARY AND PETSCHIP_GLUE_DIR)
        set(PETSC_HAVE_HIP TRUE)
        include_directories(${PETSCHIP_INCLUDE_DIR})
        set(PETSCHIP_GLUE_SOURCES ${PETSCHIP_GLUE_DIR}/petschip_glue.c)
        # the glue's in-tree registration and its dlopen entry point are both guarded by this define
        list(APPEND PETSCCONF_HAVE_FLAGS "#ifndef PETSC_HAVE_HIP\n#define PETSC_HAVE_HIP 1\n#endif\n\n")
        list(APPEND PETSCCONF_HAVE_FLAGS "#ifndef PETSC_HIP_ARCH\n#define PETSC_HIP_ARCH \"${PETSC_HIP_ARCH}\"\n#endif\n\n")
        list(APPEND PETSC_PACKAGE_LIBS ${PETSCHIP_LIBRARY})
        list(APPEND PETSC_CONFIGINFO_STRING "--with-hip=1 --with-hip-lib=[${PETSCHIP_LIBRARY}] --with-hip-include=[${PETSCHIP_INCLUDE_DIR}] --with-hip-arch=${PETSC_HIP_ARCH}")
        message(STATUS "Building PETSC with HIP (libpetschip: ${PETSCHIP_LIBRARY})...")
    else ()
        message(WARNING "Could not locate libpetschip (petschip.h, libpetschip.so, petschip_glue.c; set PETSCHIP_DIR). Disabling.")
    endif ()
endif ()
