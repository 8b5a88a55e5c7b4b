/* harness/ksp_restated.h -- C interface of libpetschip_harness.so: the KSP slice of include/petscksp.h that tests/ and bench.py drive.
 * TEST AND BENCHMARK HARNESS, NOT PRODUCT (see ksp_restated.cpp): the product boundary is include/petschip.h (kernels) and
 * include/petschip_host.h (types, ops tables, assembly, PCJACOBI); inside a real PETSc these are PETSc's own unmodified routines. */
#ifndef PETSCHIP_KSP_RESTATED_H
#define PETSCHIP_KSP_RESTATED_H
#include "../include/petschip_host.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ---- KSP (include/petscksp.h) ---- */
PetscErrorCode KSPCreate(PetscComm comm, KSP *ksp);
PetscErrorCode KSPSetType(KSP ksp, const char *type);
PetscErrorCode KSPSetOperators(KSP ksp, Mat Amat, Mat Pmat);
PetscErrorCode KSPGetPC(KSP ksp, PC *pc);
PetscErrorCode KSPSetTolerances(KSP ksp, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt maxits);
PetscErrorCode KSPSetInitialGuessNonzero(KSP ksp, PetscBool flg);
PetscErrorCode KSPSetNormType(KSP ksp, KSPNormType normtype);
PetscErrorCode KSPGMRESSetRestart(KSP ksp, PetscInt restart);
PetscErrorCode KSPGMRESSetCGSRefinementType(KSP ksp, KSPGMRESCGSRefinementType type);
PetscErrorCode KSPGMRESSetHapTol(KSP ksp, PetscReal tol);
PetscErrorCode KSPCGSetSingleReduction(KSP ksp, PetscBool flg); /* -ksp_cg_single_reduction, cg.c:226-243,262-268 */
PetscErrorCode KSPSetResidualHistory(KSP ksp, PetscReal a[], PetscInt na, PetscBool reset);
PetscErrorCode KSPGetResidualHistory(KSP ksp, PetscReal **a, PetscInt *na);
PetscErrorCode KSPSetUp(KSP ksp);
PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x);
PetscErrorCode KSPGetIterationNumber(KSP ksp, PetscInt *its);
PetscErrorCode KSPGetConvergedReason(KSP ksp, KSPConvergedReason *reason);
PetscErrorCode KSPGetResidualNorm(KSP ksp, PetscReal *rnorm);
PetscErrorCode KSPDestroy(KSP *ksp);
#ifdef __cplusplus
}
#endif
#endif
