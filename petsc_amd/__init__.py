"""petsc_amd -- MI355X-native sparse-solve hot path behind PETSc's AIJ / Vec / Jacobi interfaces.

capi      ctypes binding of libpetschip.so      (include/petschip.h, the kernel-level C ABI)
petsc     ctypes binding of libpetschip_host.so (include/petschip_host.h, the stand-alone mirror of the PETSc API)
The package has no CPU fallback: loading fails loudly when the HIP libraries are missing.
"""
