// libpetschip_harness.so -- TEST AND BENCHMARK HARNESS, NOT PRODUCT: the reference's unmodified callers of the hot path, restated op for
// op so that their call sequences run on the HIP types where PETSc itself cannot be built:
// KSPSolve_CG (src/ksp/ksp/impls/cg/cg.c:93-279), KSPSolve_PIPECG (impls/cg/pipecg/pipecg.c:31-169), KSPGMRESCycle and its helpers
// (impls/gmres/gmres.c:118-418, borthog2.c:36-111), KSPConvergedDefault (interface/iterativ.c:744-817), KSPSolve (interface/itfunc.c:499-611).
// Inside a real PETSc these ARE PETSc's own routines and this file does not exist.  It is layered on the public entry points of
// libpetschip_host.so only (include/petschip_host.h: Vec*, Mat*, PC*): no access to that library's structs, no numerical loop.
// Cites are file:line under the PETSc 3.6.1 reference tree.
#include "ksp_restated.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#define PETSC_ERR_SUP 56
#define PETSC_ERR_ARG_SIZ 60
#define PETSC_ERR_ARG_IDN 61
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_LIB 76
#define PETSC_ERR_ARG_NULL 85
#define PETSC_ERR_ARG_UNKNOWN_TYPE 86

static PetscErrorCode seterr(PetscErrorCode code, const char *fmt, ...)
{
  char    buf[640];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return PetscHipSetError(code, buf); /* the text PetscHipLastError returns lives in libpetschip_host.so */
}
#define SETERRQ(code, ...) return seterr((code), __VA_ARGS__)
#define CHKERRQ(ierr) do { if (ierr) return (ierr); } while (0)

/* ================================================================================================ */
/* KSP: the unmodified callers, restated op for op (cg.c:93-279, gmres.c:118-418, borthog2.c:36-111)   */
struct _p_KSP {
  PetscComm   comm;
  std::string type;
  PC          pc;
  Mat         amat, pmat;
  PetscReal   rtol, abstol, divtol, ttol, rnorm0, rnorm;
  PetscInt    max_it, its;
  bool        guess_zero, guess_zero_user;
  KSPNormType normtype;
  KSPConvergedReason reason;
  std::vector<Vec> work;
  Vec         vec_rhs, vec_sol;
  /* residual history */
  PetscReal  *hist; PetscInt hist_cap, hist_len; std::vector<PetscReal> hist_own;
  /* CG */
  bool        singlereduction;
  /* GMRES (gmresimpl.h:11-46) */
  PetscInt    max_k; KSPGMRESCGSRefinementType cgstype; PetscReal haptol;
  std::vector<PetscScalar> hh, rs, cc, ss, orthogwork, nrs;
  std::vector<Vec> vecs; PetscInt vv_allocated;
  bool        setup;
};
extern "C" PetscErrorCode KSPCreate(PetscComm comm, KSP *ksp)
{
  if (!comm || !ksp) SETERRQ(PETSC_ERR_ARG_NULL, "KSPCreate: null argument");
  KSP k = new _p_KSP();
  k->comm = comm; k->type = KSPGMRES; /* default KSP type is GMRES (itcreate.c) */
  k->pc = NULL; k->amat = k->pmat = NULL;
  k->rtol = 1.e-5; k->abstol = 1.e-50; k->divtol = 1.e4; k->max_it = 10000; /* itcreate.c:706-714 */
  k->ttol = 0; k->rnorm0 = 0; k->rnorm = 0; k->its = 0; k->guess_zero = k->guess_zero_user = true;
  k->normtype = KSP_NORM_DEFAULT; k->reason = KSP_CONVERGED_ITERATING;
  k->vec_rhs = k->vec_sol = NULL; k->hist = NULL; k->hist_cap = 0; k->hist_len = 0;
  k->singlereduction = false;
  k->max_k = 30; k->cgstype = KSP_GMRES_CGS_REFINE_NEVER; k->haptol = 1.0e-30; /* gmres.c:925-933 */
  k->vv_allocated = 0; k->setup = false;
  PetscErrorCode ierr = PCCreate(comm, &k->pc); CHKERRQ(ierr);
  *ksp = k;
  return 0;
}
static PetscErrorCode ksp_free_work(KSP k)
{
  for (auto &v : k->work) { PetscErrorCode ierr = VecDestroy(&v); CHKERRQ(ierr); }
  k->work.clear();
  for (auto &v : k->vecs) { PetscErrorCode ierr = VecDestroy(&v); CHKERRQ(ierr); }
  k->vecs.clear(); k->vv_allocated = 0; k->setup = false;
  return 0;
}
extern "C" PetscErrorCode KSPSetType(KSP k, const char *type)
{
  if (!k || !type) SETERRQ(PETSC_ERR_ARG_NULL, "KSPSetType: null argument");
  if (strcmp(type, KSPCG) && strcmp(type, KSPGMRES) && strcmp(type, KSPPIPECG)) SETERRQ(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unable to find requested KSP type %s", type);
  if (k->type != type) { PetscErrorCode ierr = ksp_free_work(k); CHKERRQ(ierr); }
  k->type = type;
  return 0;
}
extern "C" PetscErrorCode KSPSetOperators(KSP k, Mat Amat, Mat Pmat)
{
  if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "KSPSetOperators: null KSP");
  k->amat = Amat; k->pmat = Pmat;
  return PCSetOperators(k->pc, Amat, Pmat);
}
extern "C" PetscErrorCode KSPGetPC(KSP k, PC *pc) { if (!k || !pc) SETERRQ(PETSC_ERR_ARG_NULL, "KSPGetPC: null argument"); *pc = k->pc; return 0; }
extern "C" PetscErrorCode KSPSetTolerances(KSP k, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt maxits)
{
  if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "KSPSetTolerances: null KSP");
  if (rtol != PETSC_DEFAULT) { if (rtol < 0.0 || 1.0 <= rtol) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Relative tolerance %g must be non-negative and less than 1.0", rtol); k->rtol = rtol; }
  if (abstol != PETSC_DEFAULT) { if (abstol < 0.0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Absolute tolerance %g must be non-negative", abstol); k->abstol = abstol; }
  if (dtol != PETSC_DEFAULT) { if (dtol < 0.0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Divergence tolerance %g must be larger than 1.0", dtol); k->divtol = dtol; }
  if (maxits != PETSC_DEFAULT) { if (maxits < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Maximum number of iterations %d must be non-negative", maxits); k->max_it = maxits; }
  return 0;
}
extern "C" PetscErrorCode KSPSetInitialGuessNonzero(KSP k, PetscBool flg) { if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "null KSP"); k->guess_zero_user = !flg; return 0; }
extern "C" PetscErrorCode KSPSetNormType(KSP k, KSPNormType t) { if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "null KSP"); k->normtype = t; return 0; }
extern "C" PetscErrorCode KSPGMRESSetRestart(KSP k, PetscInt r)
{
  if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "null KSP");
  if (r < 1) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Restart must be positive"); /* gmres.c:736 */
  if (r != k->max_k) { PetscErrorCode ierr = ksp_free_work(k); CHKERRQ(ierr); }
  k->max_k = r;
  return 0;
}
extern "C" PetscErrorCode KSPGMRESSetCGSRefinementType(KSP k, KSPGMRESCGSRefinementType t) { if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "null KSP"); k->cgstype = t; return 0; }
extern "C" PetscErrorCode KSPGMRESSetHapTol(KSP k, PetscReal tol) { if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "null KSP"); if (tol < 0.0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Tolerance must be non-negative"); k->haptol = tol; return 0; }
extern "C" PetscErrorCode KSPCGSetSingleReduction(KSP k, PetscBool flg)
{
  if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "null KSP");
  if ((flg != 0) != k->singlereduction) { PetscErrorCode ierr = ksp_free_work(k); CHKERRQ(ierr); }
  k->singlereduction = flg != 0;
  return 0;
}
extern "C" PetscErrorCode KSPSetResidualHistory(KSP k, PetscReal a[], PetscInt na, PetscBool reset)
{
  if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "null KSP");
  (void)reset;
  if (a) { k->hist = a; k->hist_cap = na; }
  else { k->hist_own.assign((size_t)(na > 0 ? na : 10000), 0.0); k->hist = k->hist_own.data(); k->hist_cap = (PetscInt)k->hist_own.size(); }
  k->hist_len = 0;
  return 0;
}
extern "C" PetscErrorCode KSPGetResidualHistory(KSP k, PetscReal **a, PetscInt *na)
{
  if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "null KSP");
  if (a) *a = k->hist;
  if (na) *na = k->hist_len;
  return 0;
}
static inline void KSPLogResidualHistory(KSP k, PetscReal r) { if (k->hist && k->hist_len < k->hist_cap) k->hist[k->hist_len++] = r; }

extern "C" PetscErrorCode KSPSetUp(KSP k)
{
  if (!k) SETERRQ(PETSC_ERR_ARG_NULL, "KSPSetUp: null KSP");
  if (!k->amat) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "KSPSetUp: operators not set");
  if (!k->vec_sol) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "KSPSetUp: solution vector not set (call through KSPSolve)");
  PetscErrorCode ierr;
  if (k->normtype == KSP_NORM_DEFAULT) k->normtype = KSP_NORM_PRECONDITIONED; /* CG cg.c:424, GMRES gmres.c:915: left PC, preconditioned norm */
  { const char *pctype = NULL; ierr = PCGetType(k->pc, &pctype); CHKERRQ(ierr); if (!pctype) { ierr = PCSetType(k->pc, PCJACOBI); CHKERRQ(ierr); } }
  if (!k->setup) {
    if (k->type == KSPCG || k->type == KSPPIPECG) { /* KSPSetUp_CG cg.c:56-80: 3 work vectors (+2 for single reduction); KSPSetUp_PIPECG pipecg.c:13-20: 9 */
      const PetscInt nwork = k->type == KSPPIPECG ? 9 : (k->singlereduction ? 5 : 3);
      Vec           *w;
      ierr = VecDuplicateVecs(k->vec_sol, nwork, &w); CHKERRQ(ierr);
      k->work.assign(w, w + nwork);
      free(w);
    } else { /* KSPSetUp_GMRES gmres.c:39-97 */
      const PetscInt mk = k->max_k;
      k->hh.assign((size_t)(mk + 2) * (mk + 1), 0.0); k->rs.assign((size_t)mk + 2, 0.0);
      k->cc.assign((size_t)mk + 1, 0.0); k->ss.assign((size_t)mk + 1, 0.0);
      k->orthogwork.assign((size_t)mk + 2, 0.0); k->nrs.assign((size_t)mk + 2, 0.0);
      k->vecs.assign((size_t)(2 + 2 + mk), NULL); /* VEC_OFFSET + 2 + max_k */
      Vec *w;
      ierr = VecDuplicateVecs(k->vec_sol, 5, &w); CHKERRQ(ierr); /* :85-95 */
      for (int q = 0; q < 5 && q < (int)k->vecs.size(); q++) k->vecs[(size_t)q] = w[q];
      for (int q = (int)k->vecs.size(); q < 5; q++) VecDestroy(&w[q]);
      free(w);
      k->vv_allocated = std::min<PetscInt>(5, (PetscInt)k->vecs.size());
    }
    k->setup = true;
  }
  return PCSetUp(k->pc); /* itfunc.c:332 */
}

/* KSPConvergedDefault iterativ.c:744-817 */
static PetscErrorCode KSPConvergedDefault(KSP k, PetscInt n, PetscReal rnorm, KSPConvergedReason *reason)
{
  *reason = KSP_CONVERGED_ITERATING;
  if (k->normtype == KSP_NORM_NONE) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Use KSPConvergedSkip() with KSPNormType of KSP_NORM_NONE");
  if (!n) {
    if (!k->guess_zero) { /* :761-788 */
      PetscReal      snorm = 0;
      PetscErrorCode ierr;
      if (k->normtype == KSP_NORM_UNPRECONDITIONED) { ierr = VecNorm(k->vec_rhs, NORM_2, &snorm); CHKERRQ(ierr); }
      else {
        Vec z;
        ierr = VecDuplicate(k->vec_rhs, &z); CHKERRQ(ierr);
        ierr = PCApply(k->pc, k->vec_rhs, z); CHKERRQ(ierr);
        if (k->normtype == KSP_NORM_PRECONDITIONED) { ierr = VecNorm(z, NORM_2, &snorm); CHKERRQ(ierr); }
        else { PetscScalar nrm; ierr = VecDot(k->vec_rhs, z, &nrm); CHKERRQ(ierr); snorm = sqrt(fabs(nrm)); }
        ierr = VecDestroy(&z); CHKERRQ(ierr);
      }
      if (!snorm) snorm = rnorm;
      k->rnorm0 = snorm;
    } else k->rnorm0 = rnorm;
    k->ttol = std::max(k->rtol * k->rnorm0, k->abstol);
  }
  if (std::isnan(rnorm) || std::isinf(rnorm)) *reason = KSP_DIVERGED_NANORINF;
  else if (rnorm <= k->ttol) *reason = (rnorm < k->abstol) ? KSP_CONVERGED_ATOL : KSP_CONVERGED_RTOL;
  else if (rnorm >= k->divtol * k->rnorm0) *reason = KSP_DIVERGED_DTOL;
  return 0;
}
#define KSPCheckDot(k, beta) do { if (std::isnan(beta) || std::isinf(beta)) { (k)->reason = KSP_DIVERGED_NANORINF; return 0; } } while (0) /* kspimpl.h:302-309 */

/* KSPSolve_CG cg.c:93-279 */
static PetscErrorCode KSPSolve_CG(KSP ksp)
{
  PetscErrorCode ierr;
  PetscInt       i;
  PetscScalar    dpi = 0.0, a = 1.0, beta = 0.0, betaold = 1.0, b = 0, delta = 0, dpiold;
  PetscReal      dp = 0.0;
  Vec            X = ksp->vec_sol, B = ksp->vec_rhs, R = ksp->work[0], Z = ksp->work[1], P = ksp->work[2], S, W;
  Mat            Amat = ksp->amat;
  const bool     sr = ksp->singlereduction;
  if (sr) { S = ksp->work[3]; W = ksp->work[4]; } else { S = NULL; W = Z; }
  ksp->its = 0;
  if (!ksp->guess_zero) { ierr = MatMult(Amat, X, R); CHKERRQ(ierr); ierr = VecAYPX(R, -1.0, B); CHKERRQ(ierr); } /* :131-132 */
  else { ierr = VecCopy(B, R); CHKERRQ(ierr); }                                                                  /* :134 */
  switch (ksp->normtype) {
  case KSP_NORM_PRECONDITIONED:
    ierr = PCApply(ksp->pc, R, Z); CHKERRQ(ierr); ierr = VecNorm(Z, NORM_2, &dp); CHKERRQ(ierr); break;          /* :139-140 */
  case KSP_NORM_UNPRECONDITIONED:
    ierr = VecNorm(R, NORM_2, &dp); CHKERRQ(ierr); break;                                                        /* :143 */
  case KSP_NORM_NATURAL:
    ierr = PCApply(ksp->pc, R, Z); CHKERRQ(ierr);
    if (sr) { ierr = MatMult(Amat, Z, S); CHKERRQ(ierr); ierr = VecTDot(Z, S, &delta); CHKERRQ(ierr); }
    ierr = VecTDot(Z, R, &beta); CHKERRQ(ierr);
    KSPCheckDot(ksp, beta);
    dp = sqrt(fabs(beta)); break;                                                                                /* :146-153 */
  case KSP_NORM_NONE: dp = 0.0; break;
  default: SETERRQ(PETSC_ERR_SUP, "KSPSolve_CG: unsupported norm type %d", (int)ksp->normtype);
  }
  KSPLogResidualHistory(ksp, dp);
  ksp->rnorm = dp;
  ierr = KSPConvergedDefault(ksp, 0, dp, &ksp->reason); CHKERRQ(ierr);                                           /* :164 */
  if (ksp->reason) return 0;
  if (ksp->normtype != KSP_NORM_PRECONDITIONED && ksp->normtype != KSP_NORM_NATURAL) { ierr = PCApply(ksp->pc, R, Z); CHKERRQ(ierr); } /* :167-169 */
  if (ksp->normtype != KSP_NORM_NATURAL) {
    if (sr) { ierr = MatMult(Amat, Z, S); CHKERRQ(ierr); ierr = VecTDot(Z, S, &delta); CHKERRQ(ierr); }
    ierr = VecTDot(Z, R, &beta); CHKERRQ(ierr);                                                                  /* :175 */
    KSPCheckDot(ksp, beta);
  }
  i = 0;
  do {
    ksp->its = i + 1;
    if (beta == 0.0) { ksp->reason = KSP_CONVERGED_ATOL; break; }                                                /* :182-185 */
    else if ((i > 0) && (beta * betaold < 0.0)) { ksp->reason = KSP_DIVERGED_INDEFINITE_PC; break; }             /* :187-190 */
    if (!i) { ierr = VecCopy(Z, P); CHKERRQ(ierr); b = 0.0; }                                                    /* :194 */
    else { b = beta / betaold; ierr = VecAYPX(P, b, Z); CHKERRQ(ierr); }                                         /* :197,202 */
    dpiold = dpi;
    if (!sr || !i) {
      ierr = MatMult(Amat, P, W); CHKERRQ(ierr);                                                                 /* :206 */
      ierr = VecTDot(P, W, &dpi); CHKERRQ(ierr);                                                                 /* :207 */
    } else {
      ierr = VecAYPX(W, beta / betaold, S); CHKERRQ(ierr);                                                       /* :209 */
      dpi = delta - beta * beta * dpiold / (betaold * betaold);                                                  /* :210 */
    }
    betaold = beta;
    KSPCheckDot(ksp, beta);
    if ((dpi == 0.0) || ((i > 0) && (dpi * dpiold <= 0.0))) { ksp->reason = KSP_DIVERGED_INDEFINITE_MAT; break; } /* :215-219 */
    a = beta / dpi;                                                                                              /* :220 */
    ierr = VecAXPY(X, a, P); CHKERRQ(ierr);                                                                      /* :222 */
    ierr = VecAXPY(R, -a, W); CHKERRQ(ierr);                                                                     /* :223 */
    if (ksp->normtype == KSP_NORM_PRECONDITIONED) {
      ierr = PCApply(ksp->pc, R, Z); CHKERRQ(ierr);                                                              /* :225 */
      if (sr) { ierr = MatMult(Amat, Z, S); CHKERRQ(ierr); }
      ierr = VecNorm(Z, NORM_2, &dp); CHKERRQ(ierr);                                                             /* :229 */
    } else if (ksp->normtype == KSP_NORM_UNPRECONDITIONED) {
      ierr = VecNorm(R, NORM_2, &dp); CHKERRQ(ierr);                                                             /* :231 */
    } else if (ksp->normtype == KSP_NORM_NATURAL) {
      ierr = PCApply(ksp->pc, R, Z); CHKERRQ(ierr);
      if (sr) {
        PetscScalar tmp[2]; Vec vecs[2] = {S, R};
        ierr = MatMult(Amat, Z, S); CHKERRQ(ierr);
        ierr = VecMDot(Z, 2, vecs, tmp); CHKERRQ(ierr);
        delta = tmp[0]; beta = tmp[1];
      } else { ierr = VecTDot(Z, R, &beta); CHKERRQ(ierr); }
      KSPCheckDot(ksp, beta);
      dp = sqrt(fabs(beta));
    } else dp = 0.0;
    ksp->rnorm = dp;
    KSPLogResidualHistory(ksp, dp);
    ierr = KSPConvergedDefault(ksp, i + 1, dp, &ksp->reason); CHKERRQ(ierr);                                     /* :252 */
    if (ksp->reason) break;
    if (ksp->normtype != KSP_NORM_PRECONDITIONED && ksp->normtype != KSP_NORM_NATURAL) {
      ierr = PCApply(ksp->pc, R, Z); CHKERRQ(ierr);                                                              /* :256 */
      if (sr) { ierr = MatMult(Amat, Z, S); CHKERRQ(ierr); }
    }
    if (ksp->normtype != KSP_NORM_NATURAL) {
      if (sr) {
        PetscScalar tmp[2]; Vec vecs[2] = {S, R};
        ierr = VecMDot(Z, 2, vecs, tmp); CHKERRQ(ierr);                                                          /* :266 */
        delta = tmp[0]; beta = tmp[1];
      } else { ierr = VecTDot(Z, R, &beta); CHKERRQ(ierr); }                                                     /* :269 */
      KSPCheckDot(ksp, beta);
    }
    i++;
  } while (i < ksp->max_it);
  if (i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;                                                          /* :276 */
  return 0;
}

/* KSPSolve_PIPECG src/ksp/ksp/impls/cg/pipecg/pipecg.c:31-169 (Ghysels & Vanroose): one reduction per iteration (gamma, delta and the
   norm queued together, comb.c), begun before PCApply and MatMult and collected after them.  On the GPU the three local kernels are
   launched first, PCApply and MatMult are queued behind them, and the host collects the sums while those still run: the device never
   waits for the host.  The price is 4 more vector updates per iteration than KSPCG. */
static PetscErrorCode KSPSolve_PIPECG(KSP ksp)
{
  PetscErrorCode ierr;
  PetscInt       i;
  PetscScalar    alpha = 0.0, beta = 0.0, gamma = 0.0, gammaold = 0.0, delta = 0.0;
  PetscReal      dp = 0.0;
  Vec            X = ksp->vec_sol, B = ksp->vec_rhs, M = ksp->work[0], Z = ksp->work[1], P = ksp->work[2], N = ksp->work[3], W = ksp->work[4],
                 Q = ksp->work[5], U = ksp->work[6], R = ksp->work[7], S = ksp->work[8];
  Mat            Amat = ksp->amat;
  PetscComm      comm = ksp->comm;
  ksp->its = 0;
  if (!ksp->guess_zero) { ierr = MatMult(Amat, X, R); CHKERRQ(ierr); ierr = VecAYPX(R, -1.0, B); CHKERRQ(ierr); } /* :62-63 */
  else { ierr = VecCopy(B, R); CHKERRQ(ierr); }                                                                  /* :65 */
  ierr = PCApply(ksp->pc, R, U); CHKERRQ(ierr);                                                                  /* :68 */
  switch (ksp->normtype) {
  case KSP_NORM_PRECONDITIONED:
    ierr = VecNormBegin(U, NORM_2, &dp); CHKERRQ(ierr);                                                          /* :72 */
    ierr = PetscCommSplitReductionBegin(comm); CHKERRQ(ierr);
    ierr = MatMult(Amat, U, W); CHKERRQ(ierr);
    ierr = VecNormEnd(U, NORM_2, &dp); CHKERRQ(ierr);
    break;
  case KSP_NORM_UNPRECONDITIONED:
    ierr = VecNormBegin(R, NORM_2, &dp); CHKERRQ(ierr);                                                          /* :78 */
    ierr = PetscCommSplitReductionBegin(comm); CHKERRQ(ierr);
    ierr = MatMult(Amat, U, W); CHKERRQ(ierr);
    ierr = VecNormEnd(R, NORM_2, &dp); CHKERRQ(ierr);
    break;
  case KSP_NORM_NATURAL:
    ierr = VecDotBegin(R, U, &gamma); CHKERRQ(ierr);                                                             /* :84 */
    ierr = PetscCommSplitReductionBegin(comm); CHKERRQ(ierr);
    ierr = MatMult(Amat, U, W); CHKERRQ(ierr);
    ierr = VecDotEnd(R, U, &gamma); CHKERRQ(ierr);
    KSPCheckDot(ksp, gamma);
    dp = sqrt(fabs(gamma));
    break;
  case KSP_NORM_NONE:
    ierr = MatMult(Amat, U, W); CHKERRQ(ierr);
    dp = 0.0;
    break;
  default: SETERRQ(PETSC_ERR_SUP, "KSPSolve_PIPECG: unsupported norm type %d", (int)ksp->normtype);
  }
  KSPLogResidualHistory(ksp, dp);
  ksp->rnorm = dp;
  ierr = KSPConvergedDefault(ksp, 0, dp, &ksp->reason); CHKERRQ(ierr);                                           /* :101 */
  if (ksp->reason) return 0;
  i = 0;
  do {
    if (i > 0 && ksp->normtype == KSP_NORM_UNPRECONDITIONED) { ierr = VecNormBegin(R, NORM_2, &dp); CHKERRQ(ierr); }       /* :106-110 */
    else if (i > 0 && ksp->normtype == KSP_NORM_PRECONDITIONED) { ierr = VecNormBegin(U, NORM_2, &dp); CHKERRQ(ierr); }
    if (!(i == 0 && ksp->normtype == KSP_NORM_NATURAL)) { ierr = VecDotBegin(R, U, &gamma); CHKERRQ(ierr); }               /* :111-113 */
    ierr = VecDotBegin(W, U, &delta); CHKERRQ(ierr);                                                             /* :114 */
    ierr = PetscCommSplitReductionBegin(comm); CHKERRQ(ierr);
    ierr = PCApply(ksp->pc, W, M); CHKERRQ(ierr);                                                                /* :117 */
    ierr = MatMult(Amat, M, N); CHKERRQ(ierr);                                                                   /* :118 */
    if (i > 0 && ksp->normtype == KSP_NORM_UNPRECONDITIONED) { ierr = VecNormEnd(R, NORM_2, &dp); CHKERRQ(ierr); }
    else if (i > 0 && ksp->normtype == KSP_NORM_PRECONDITIONED) { ierr = VecNormEnd(U, NORM_2, &dp); CHKERRQ(ierr); }
    if (!(i == 0 && ksp->normtype == KSP_NORM_NATURAL)) { ierr = VecDotEnd(R, U, &gamma); CHKERRQ(ierr); }
    ierr = VecDotEnd(W, U, &delta); CHKERRQ(ierr);
    if (i > 0) {
      if (ksp->normtype == KSP_NORM_NATURAL) dp = sqrt(fabs(gamma));
      else if (ksp->normtype == KSP_NORM_NONE) dp = 0.0;
      ksp->rnorm = dp;
      KSPLogResidualHistory(ksp, dp);
      ierr = KSPConvergedDefault(ksp, i, dp, &ksp->reason); CHKERRQ(ierr);                                       /* :137 */
      if (ksp->reason) break;
    }
    if (i == 0) {
      alpha = gamma / delta;
      ierr = VecCopy(N, Z); CHKERRQ(ierr); ierr = VecCopy(M, Q); CHKERRQ(ierr);                                  /* :143-146 */
      ierr = VecCopy(U, P); CHKERRQ(ierr); ierr = VecCopy(W, S); CHKERRQ(ierr);
    } else {
      beta = gamma / gammaold;
      alpha = gamma / (delta - beta / alpha * gamma);                                                            /* :149 */
      ierr = VecAYPX(Z, beta, N); CHKERRQ(ierr); ierr = VecAYPX(Q, beta, M); CHKERRQ(ierr);                      /* :150-153 */
      ierr = VecAYPX(P, beta, U); CHKERRQ(ierr); ierr = VecAYPX(S, beta, W); CHKERRQ(ierr);
    }
    ierr = VecAXPY(X, alpha, P); CHKERRQ(ierr);                                                                  /* :155-158 */
    ierr = VecAXPY(U, -alpha, Q); CHKERRQ(ierr);
    ierr = VecAXPY(W, -alpha, Z); CHKERRQ(ierr);
    ierr = VecAXPY(R, -alpha, S); CHKERRQ(ierr);
    gammaold = gamma;
    i++;
    ksp->its = i;
  } while (i < ksp->max_it);
  if (i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}

/* ---- GMRES ---- */
#define VEC_OFFSET 2
#define VEC_TEMP ksp->vecs[0]
#define VEC_TEMP_MATOP ksp->vecs[1]
#define VEC_VV(i) ksp->vecs[(size_t)(VEC_OFFSET + (i))]
#define HH(a, b) (ksp->hh.data() + (size_t)(b) * (size_t)(ksp->max_k + 2) + (size_t)(a))
#define GRS(a) (ksp->rs.data() + (a))

/* KSPGMRESGetNewVectors gmres.c:424-450: delta_allocate = 10 more direction vectors */
static PetscErrorCode KSPGMRESGetNewVectors(KSP ksp, PetscInt it)
{
  PetscInt nalloc = 10;
  const PetscInt total = (PetscInt)ksp->vecs.size();
  (void)it;
  if (ksp->vv_allocated + nalloc > total) nalloc = total - ksp->vv_allocated;
  if (nalloc <= 0) return 0;
  Vec           *w;
  PetscErrorCode ierr = VecDuplicateVecs(ksp->vec_sol, nalloc, &w); CHKERRQ(ierr);
  for (PetscInt q = 0; q < nalloc; q++) ksp->vecs[(size_t)(ksp->vv_allocated + q)] = w[q];
  free(w);
  ksp->vv_allocated += nalloc;
  return 0;
}
/* KSPGMRESClassicalGramSchmidtOrthogonalization borthog2.c:36-111 */
static PetscErrorCode KSPGMRESClassicalGramSchmidt(KSP ksp, PetscInt it)
{
  PetscErrorCode ierr;
  PetscScalar   *hh = HH(0, it), *lhh = ksp->orthogwork.data();
  bool           refine = (ksp->cgstype == KSP_GMRES_CGS_REFINE_ALWAYS);
  for (PetscInt j = 0; j <= it; j++) hh[j] = 0.0;
  ierr = VecMDot(VEC_VV(it + 1), it + 1, &VEC_VV(0), lhh); CHKERRQ(ierr);                   /* :66 */
  for (PetscInt j = 0; j <= it; j++) { KSPCheckDot(ksp, lhh[j]); lhh[j] = -lhh[j]; }        /* :67-70 */
  ierr = VecMAXPY(VEC_VV(it + 1), it + 1, lhh, &VEC_VV(0)); CHKERRQ(ierr);                  /* :76 */
  for (PetscInt j = 0; j <= it; j++) hh[j] -= lhh[j];                                      /* :78-81 */
  if (ksp->cgstype == KSP_GMRES_CGS_REFINE_IFNEEDED) {                                      /* :87-97 */
    PetscReal hnrm = 0.0, wnrm;
    for (PetscInt j = 0; j <= it; j++) hnrm += lhh[j] * lhh[j];
    hnrm = sqrt(hnrm);
    ierr = VecNorm(VEC_VV(it + 1), NORM_2, &wnrm); CHKERRQ(ierr);
    if (wnrm < hnrm) refine = true;
  }
  if (refine) {                                                                             /* :99-108 */
    ierr = VecMDot(VEC_VV(it + 1), it + 1, &VEC_VV(0), lhh); CHKERRQ(ierr);
    for (PetscInt j = 0; j <= it; j++) lhh[j] = -lhh[j];
    ierr = VecMAXPY(VEC_VV(it + 1), it + 1, lhh, &VEC_VV(0)); CHKERRQ(ierr);
    for (PetscInt j = 0; j <= it; j++) hh[j] -= lhh[j];
  }
  return 0;
}
/* KSPGMRESUpdateHessenberg gmres.c:369-418 */
static void KSPGMRESUpdateHessenberg(KSP ksp, PetscInt it, bool hapend, PetscReal *res)
{
  PetscScalar *hh = HH(0, it), *cc = ksp->cc.data(), *ss = ksp->ss.data(), tt;
  for (PetscInt j = 1; j <= it; j++) {
    tt = *hh;
    *hh = *cc * tt + *ss * *(hh + 1);
    hh++;
    *hh = *cc++ * *hh - (*ss++ * tt);
  }
  if (!hapend) {
    tt = sqrt(*hh * *hh + *(hh + 1) * *(hh + 1));
    if (tt == 0.0) { ksp->reason = KSP_DIVERGED_NULL; return; }
    *cc = *hh / tt;
    *ss = *(hh + 1) / tt;
    *GRS(it + 1) = -(*ss * *GRS(it));
    *GRS(it) = *cc * *GRS(it);
    *hh = *cc * *hh + *ss * *(hh + 1);
    *res = fabs(*GRS(it + 1));
  } else *res = 0.0;
}
/* KSPGMRESBuildSoln gmres.c:316-363 (left preconditioning: KSPUnwindPreconditioner is a no-op, itres.c:114-116) */
static PetscErrorCode KSPGMRESBuildSoln(KSP ksp, PetscInt it)
{
  PetscErrorCode ierr;
  PetscScalar   *nrs = ksp->nrs.data(), tt;
  if (it < 0) return 0;
  if (*HH(it, it) != 0.0) nrs[it] = *GRS(it) / *HH(it, it);
  else { ksp->reason = KSP_DIVERGED_BREAKDOWN; return 0; }
  for (PetscInt ii = 1; ii <= it; ii++) {
    const PetscInt k = it - ii;
    tt = *GRS(k);
    for (PetscInt j = k + 1; j <= it; j++) tt = tt - *HH(k, j) * nrs[j];
    if (*HH(k, k) == 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; return 0; }
    nrs[k] = tt / *HH(k, k);
  }
  ierr = VecSet(VEC_TEMP, 0.0); CHKERRQ(ierr);
  ierr = VecMAXPY(VEC_TEMP, it + 1, nrs, &VEC_VV(0)); CHKERRQ(ierr);
  ierr = VecAXPY(ksp->vec_sol, 1.0, VEC_TEMP); CHKERRQ(ierr);
  return 0;
}
/* KSPGMRESCycle gmres.c:118-214 */
static PetscErrorCode KSPGMRESCycle(PetscInt *itcount, KSP ksp)
{
  PetscReal      res_norm, res, hapbnd, tt;
  PetscErrorCode ierr;
  PetscInt       it = 0;
  const PetscInt max_k = ksp->max_k;
  bool           hapend = false;
  ierr = VecNormalize(VEC_VV(0), &res_norm); CHKERRQ(ierr);                                 /* :127 */
  res = res_norm;
  *GRS(0) = res_norm;
  ksp->rnorm = res;
  KSPLogResidualHistory(ksp, res);
  if (!res) { if (itcount) *itcount = 0; ksp->reason = KSP_CONVERGED_ATOL; return 0; }      /* :138-143 */
  ierr = KSPConvergedDefault(ksp, ksp->its, res, &ksp->reason); CHKERRQ(ierr);
  while (!ksp->reason && it < max_k && ksp->its < ksp->max_it) {
    if (it) KSPLogResidualHistory(ksp, res);
    if (ksp->vv_allocated <= it + VEC_OFFSET + 1) { ierr = KSPGMRESGetNewVectors(ksp, it + 1); CHKERRQ(ierr); } /* :152-154 */
    /* KSP_PCApplyBAorAB, left PC: MatMult then PCApply (precon.c:722-734) */
    ierr = MatMult(ksp->amat, VEC_VV(it), VEC_TEMP_MATOP); CHKERRQ(ierr);
    ierr = PCApply(ksp->pc, VEC_TEMP_MATOP, VEC_VV(1 + it)); CHKERRQ(ierr);
    ierr = KSPGMRESClassicalGramSchmidt(ksp, it); CHKERRQ(ierr);                            /* :158 */
    if (ksp->reason) break;
    ierr = VecNormalize(VEC_VV(it + 1), &tt); CHKERRQ(ierr);                                /* :162 */
    *HH(it + 1, it) = tt;
    hapbnd = fabs(tt / *GRS(it));                                                           /* :169-174 */
    if (hapbnd > ksp->haptol) hapbnd = ksp->haptol;
    if (tt < hapbnd) hapend = true;
    KSPGMRESUpdateHessenberg(ksp, it, hapend, &res);                                        /* :175 */
    it++;
    ksp->its++;
    ksp->rnorm = res;
    if (ksp->reason) break;
    ierr = KSPConvergedDefault(ksp, ksp->its, res, &ksp->reason); CHKERRQ(ierr);
    if (hapend && !ksp->reason) { ksp->reason = KSP_DIVERGED_BREAKDOWN; break; }            /* :186-194 */
  }
  if (it && (ksp->reason || ksp->its >= ksp->max_it)) KSPLogResidualHistory(ksp, res);      /* :198-201 */
  if (itcount) *itcount = it;
  return KSPGMRESBuildSoln(ksp, it - 1);                                                    /* :212 */
}
/* KSPSolve_GMRES gmres.c:218-246 with KSPInitialResidual itres.c:39-70 (left PC) */
static PetscErrorCode KSPSolve_GMRES(KSP ksp)
{
  PetscErrorCode ierr;
  PetscInt       its, itcount = 0;
  const bool     guess_zero = ksp->guess_zero;
  ksp->its = 0;
  ksp->reason = KSP_CONVERGED_ITERATING;
  while (!ksp->reason) {
    if (!ksp->guess_zero) {
      ierr = MatMult(ksp->amat, ksp->vec_sol, VEC_TEMP); CHKERRQ(ierr);
      ierr = VecCopy(ksp->vec_rhs, VEC_TEMP_MATOP); CHKERRQ(ierr);
      ierr = VecAXPY(VEC_TEMP_MATOP, -1.0, VEC_TEMP); CHKERRQ(ierr);
      ierr = PCApply(ksp->pc, VEC_TEMP_MATOP, VEC_VV(0)); CHKERRQ(ierr);
    } else {
      ierr = VecCopy(ksp->vec_rhs, VEC_TEMP_MATOP); CHKERRQ(ierr);
      ierr = PCApply(ksp->pc, ksp->vec_rhs, VEC_VV(0)); CHKERRQ(ierr);
    }
    ierr = KSPGMRESCycle(&its, ksp); CHKERRQ(ierr);
    itcount += its;
    if (itcount >= ksp->max_it) { if (!ksp->reason) ksp->reason = KSP_DIVERGED_ITS; break; }
    ksp->guess_zero = false;
  }
  ksp->guess_zero = guess_zero;
  return 0;
}

/* KSPSolve itfunc.c:499-611 */
extern "C" PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x)
{
  if (!ksp || !b || !x) SETERRQ(PETSC_ERR_ARG_NULL, "KSPSolve: null argument");
  if (b == x) SETERRQ(PETSC_ERR_ARG_IDN, "b and x must be different vectors"); /* itfunc.c:513-519 would duplicate; not supported here */
  PetscErrorCode ierr;
  ksp->vec_rhs = b; ksp->vec_sol = x;
  if (!ksp->amat) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "KSPSolve: call KSPSetOperators() first");
  { PetscInt am = 0, an = 0, bn = 0, xn = 0;
    ierr = MatGetLocalSize(ksp->amat, &am, &an); CHKERRQ(ierr); ierr = VecGetLocalSize(b, &bn); CHKERRQ(ierr); ierr = VecGetLocalSize(x, &xn); CHKERRQ(ierr);
    if (am != bn || an != xn) SETERRQ(PETSC_ERR_ARG_SIZ, "KSPSolve: matrix and vector local sizes do not conform"); }
  ierr = KSPSetUp(ksp); CHKERRQ(ierr);
  ksp->guess_zero = ksp->guess_zero_user;
  if (ksp->guess_zero) { ierr = VecSet(x, 0.0); CHKERRQ(ierr); } /* itfunc.c:580 */
  ksp->hist_len = 0;
  ksp->reason = KSP_CONVERGED_ITERATING;
  if (ksp->type == KSPCG) ierr = KSPSolve_CG(ksp);
  else if (ksp->type == KSPPIPECG) ierr = KSPSolve_PIPECG(ksp);
  else ierr = KSPSolve_GMRES(ksp);
  CHKERRQ(ierr);
  if (!ksp->reason) SETERRQ(PETSC_ERR_LIB, "Internal error, solver returned without setting converged reason"); /* itfunc.c:607 */
  return 0;
}
extern "C" PetscErrorCode KSPGetIterationNumber(KSP k, PetscInt *its) { if (!k || !its) SETERRQ(PETSC_ERR_ARG_NULL, "null argument"); *its = k->its; return 0; }
extern "C" PetscErrorCode KSPGetConvergedReason(KSP k, KSPConvergedReason *r) { if (!k || !r) SETERRQ(PETSC_ERR_ARG_NULL, "null argument"); *r = k->reason; return 0; }
extern "C" PetscErrorCode KSPGetResidualNorm(KSP k, PetscReal *r) { if (!k || !r) SETERRQ(PETSC_ERR_ARG_NULL, "null argument"); *r = k->rnorm; return 0; }
extern "C" PetscErrorCode KSPDestroy(KSP *ksp)
{
  if (!ksp || !*ksp) return 0;
  PetscErrorCode ierr = ksp_free_work(*ksp); CHKERRQ(ierr);
  ierr = PCDestroy(&(*ksp)->pc); CHKERRQ(ierr);
  delete *ksp;
  *ksp = NULL;
  return 0;
}
