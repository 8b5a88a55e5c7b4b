/* Minimal hand-written configuration header, used ONLY to let a compiler PARSE integration/petsc/petschip_glue.c against the PETSc 3.6.1
   headers (tests/test_glue_syntax.py: gcc -fsyntax-only).  It describes a real, double-precision, 32-bit-index, MPIUNI, C build on
   x86-64 Linux.  Nothing is compiled into an object, linked or run with it; it is not a way to build PETSc. */
#ifndef PETSCCONF_STUB_H
#define PETSCCONF_STUB_H
#define PETSC_ARCH "parse-only"
#define PETSC_DIR "/root/reference"
#define PETSC_LIB_DIR "/nonexistent"
#define PETSC_USE_REAL_DOUBLE 1
#define PETSC_USE_SCALAR_REAL 1
#define PETSC_CLANGUAGE_C 1
#define PETSC_HAVE_MPIUNI 1
#define PETSC_USE_BACKWARD_LOOP 1
#define PETSC_BLASLAPACK_UNDERSCORE 1
#define PETSC_HAVE_DYNAMIC_LIBRARIES 1
#define PETSC_USE_SHARED_LIBRARIES 1
#define PETSC_USE_EXTERN_CXX
#define PETSC_MEMALIGN 16
#define PETSC_LEVEL1_DCACHE_LINESIZE 64
#define PETSC_SIZEOF_INT 4
#define PETSC_SIZEOF_LONG 8
#define PETSC_SIZEOF_LONG_LONG 8
#define PETSC_SIZEOF_SHORT 2
#define PETSC_SIZEOF_CHAR 1
#define PETSC_SIZEOF_FLOAT 4
#define PETSC_SIZEOF_DOUBLE 8
#define PETSC_SIZEOF_SIZE_T 8
#define PETSC_SIZEOF_VOID_P 8
#define PETSC_SIZEOF_MPI_COMM 4
#define PETSC_SIZEOF_MPI_FINT 4
#define PETSC_BITS_PER_BYTE 8
#define PETSC_RETSIGTYPE void
#define PETSC_C_STATIC_INLINE static inline
#define PETSC_CXX_STATIC_INLINE static inline
#define PETSC_C_RESTRICT __restrict__
#define PETSC_CXX_RESTRICT __restrict__
#define PETSC_FUNCTION_NAME_C __func__
#define PETSC_FUNCTION_NAME_CXX __func__
#define PETSC_HAVE_BUILTIN_EXPECT 1
#define PETSC_HAVE_STDLIB_H 1
#define PETSC_HAVE_STRING_H 1
#define PETSC_HAVE_STDINT_H 1
#define PETSC_HAVE_UNISTD_H 1
#define PETSC_HAVE_FLOAT_H 1
#define PETSC_HAVE_LIMITS_H 1
#define PETSC_HAVE_MATH_H 1
#define PETSC_HAVE_SYS_TYPES_H 1
#define PETSC_HAVE_VISIBILITY_C 1
#define PETSC_HAVE_ISNAN 1
#define PETSC_HAVE_ISINF 1
#define PETSC_HAVE_ISNORMAL 1
#define PETSC_HAVE_LOG2 1
#define PETSC_HAVE_VA_COPY 1
#define PETSC_Prefetch(a,b,c)
#define PETSC_PREFETCH_HINT_NTA 0
#define PETSC_PREFETCH_HINT_T0 0
#define PETSC_PREFETCH_HINT_T1 0
#define PETSC_PREFETCH_HINT_T2 0
#define PETSC_UNUSED __attribute__((unused))
#define PETSC_DEPRECATED(why) __attribute__((deprecated))
#define PETSC_USE_ERRORCHECKING 1
#define PETSC_USE_LOG 1
#define PETSC_USE_INFO 1
#define PETSC_IS_COLOR_VALUE_TYPE short
#define MPIU_COLORING_VALUE MPI_UNSIGNED_SHORT
#define PETSC_VERSION_GIT "parse-only"
#define PETSC_VERSION_DATE_GIT "parse-only"
#endif
