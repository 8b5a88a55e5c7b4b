/*
 * petschip_glue.c -- the PETSc 3.6.1 side of the drop-in: registers MATSEQAIJHIP / MATMPIAIJHIP / MATAIJHIP and VECSEQHIP /
 * VECMPIHIP / VECHIP and fills their ops tables with calls into libpetschip.so (include/petschip.h).
 *
 * NOT BUILT IN THIS REPOSITORY'S IMAGE: PETSc 3.6.1 cannot be configured there (Python-2 configure; the CMake build needs Fortran,
 * MPI, BLAS/LAPACK).  It is the file a PETSc maintainer adds; the stand-alone mirror
 * petsc_amd/csrc/host/petsc_host.cpp has the same structure and is what the tests and the benchmark run.
 *
 * Build (with a configured PETSc):  mpicc -fPIC -shared -I$PETSC_DIR/include -I$PETSC_DIR/$PETSC_ARCH/include -Iinclude \
 *                                   integration/petsc/petschip_glue.c -Lpetsc_amd/lib -lpetschip -o libpetschip_glue.so
 * Use:    ./ex2 -dll_append /path/libpetschip_glue.so -mat_type aijhip -vec_type hip -ksp_type cg -pc_type jacobi
 *
 * Recipes followed (file:line in the PETSc 3.6.1 tree):
 *   MatCreate_SeqAIJCUSP   src/mat/impls/aij/seq/seqcusp/aijcusp.cu:565-598      MatCreate_MPIAIJCUSP  src/mat/impls/aij/mpi/mpicusp/mpiaijcusp.cu:196-226
 *   VecCreate_SeqCUSP ops  src/vec/vec/impls/seq/seqcusp/veccusp.cu:1952-1983    VecCreate_MPICUSP     src/vec/vec/impls/mpi/mpicusp/mpicusp.cu:188-250
 */
#include <petsc/private/vecimpl.h>
#include <petsc/private/matimpl.h>
#include <../src/vec/vec/impls/dvecimpl.h>
#include <../src/vec/vec/impls/mpi/pvecimpl.h>
#include <../src/mat/impls/aij/seq/aij.h>
#include <../src/mat/impls/aij/mpi/mpiaij.h>
#include "petschip.h"

/* libpetschip's PetscInt is int32_t and its PetscScalar is double (include/petschip.h "Conventions"): refuse any other configuration */
typedef char petschip_needs_32bit_PetscInt[(sizeof(PetscInt) == sizeof(int32_t)) ? 1 : -1];
typedef char petschip_needs_double_PetscScalar[(sizeof(PetscScalar) == sizeof(double)) ? 1 : -1];
typedef char petschip_needs_int_PetscMPIInt[(sizeof(PetscMPIInt) == sizeof(int32_t)) ? 1 : -1];

/* This file is written to work OUT OF TREE (a shared library opened by -dll_append): it calls only PETSC_EXTERN symbols of libpetsc.
   The routines a subtype normally chains to (VecDestroy_Seq, VecCreate_MPI_Private, MatDestroy_SeqAIJ, MatAssemblyEnd_MPIAIJ, ...) are
   PETSC_INTERN (hidden visibility, src/vec/vec/impls/dvecimpl.h:27-56, src/mat/impls/aij/seq/aij.h:295-314) or file-static
   (VecPlaceArray_MPI, pbvec.c:39), so the constructors below run the PUBLIC base constructor (VecCreate_Seq / VecCreate_MPI /
   MatCreate_SeqAIJ / MatCreate_MPIAIJ), keep a copy of the ops table it filled, and chain through those saved pointers. */
PETSC_EXTERN PetscErrorCode VecCreate_MPI(Vec);   /* src/vec/vec/impls/mpi/pbvec.c; declared like this by vecregall.c:6 */

#define VECSEQHIP    "seqhip"
#define VECMPIHIP    "mpihip"
#define VECHIP       "hip"
#define MATSEQAIJHIP "seqaijhip"
#define MATMPIAIJHIP "mpiaijhip"
#define MATAIJHIP    "aijhip"

#define CHKERRPHIP(call) do { int _rc = (call); if (_rc) SETERRQ1(PETSC_COMM_SELF,(PetscErrorCode)_rc,"libpetschip: %s",phip_last_error()); } while (0)

/* ---- one context and one communicator per process ------------------------------------------------------------ */
static phip_ctx  hipctx  = NULL;
static phip_comm hipcomm = NULL; /* mirrors PETSC_COMM_WORLD: one rank per GPU */
static PetscBool hipsync = PETSC_FALSE; /* -vec_hip_sync: wait for the device after every operation (role of -cusp_synchronize, src/sys/objects/init.c:674-676) */

static PetscErrorCode PetscHIPFinalize(void)
{
  PetscFunctionBegin;
  if (hipcomm) { phip_comm_destroy(hipcomm); hipcomm = NULL; }
  if (hipctx)  { phip_finalize(hipctx); hipctx = NULL; }
  PetscFunctionReturn(0);
}

static PetscErrorCode PetscHIPInitialize(void)
{
  PetscErrorCode ierr;
  PetscMPIInt    rank,size,lrank;
  MPI_Comm       node;
  char           id[128];

  PetscFunctionBegin;
  if (hipctx) PetscFunctionReturn(0);
  ierr = MPI_Comm_rank(PETSC_COMM_WORLD,&rank);CHKERRQ(ierr);
  ierr = MPI_Comm_size(PETSC_COMM_WORLD,&size);CHKERRQ(ierr);
#if defined(PETSC_HAVE_MPIUNI) || !defined(MPI_COMM_TYPE_SHARED)
  lrank = 0; node = MPI_COMM_NULL; (void)node;                   /* one process (MPIUNI has no MPI_Comm_split_type), or an MPI-2 library: device 0 */
  if (size > 1) SETERRQ(PETSC_COMM_WORLD,PETSC_ERR_SUP,"the HIP types need MPI_Comm_split_type (MPI-3) to give every rank of a node its own GPU");
#else
  ierr = MPI_Comm_split_type(PETSC_COMM_WORLD,MPI_COMM_TYPE_SHARED,rank,MPI_INFO_NULL,&node);CHKERRQ(ierr);
  ierr = MPI_Comm_rank(node,&lrank);CHKERRQ(ierr);
  ierr = MPI_Comm_free(&node);CHKERRQ(ierr);
#endif
  CHKERRPHIP(phip_init((int)lrank,&hipctx));                     /* one rank per GPU of the node */
  if (size == 1) CHKERRPHIP(phip_comm_create(hipctx,"self",NULL,1,0,&hipcomm));
  else {
    if (!rank) CHKERRPHIP(phip_comm_unique_id(id));
    ierr = MPI_Bcast(id,128,MPI_BYTE,0,PETSC_COMM_WORLD);CHKERRQ(ierr);
    CHKERRPHIP(phip_comm_create(hipctx,"rccl",id,size,rank,&hipcomm));
  }
  ierr = PetscOptionsGetBool(NULL,"-vec_hip_sync",&hipsync,NULL);CHKERRQ(ierr);
  ierr = PetscRegisterFinalize(PetscHIPFinalize);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* ---- Vec ----------------------------------------------------------------------------------------------------- */
typedef enum {VHIP_BOTH=0,VHIP_GPU=1,VHIP_CPU=2} VecHIPValid;   /* role of PetscCUSPFlag, include/petsc/private/petscimpl.h:749 */

/* by-products of the op that last wrote the vector (include/petschip.h "by-product reductions") */
typedef struct {
  int              kind;                       /* 0 none, 1 pointwise mult, 2 matmult */
  PetscBool        live,havehost;
  phip_ticket      ticket;
  PetscObjectState selfstate,partnerstate[2];
  PetscObjectId    partner[2];
  PetscScalar      host[3];
} VecHIPByProduct;

/* v->data: MUST begin with the layout of Vec_Seq / Vec_MPI (VECHEADER first, vecimpl.h:175-178) because inherited host ops cast it */
typedef struct { Vec_MPI mpi; double *d; VecHIPValid valid; PetscBool ismpi; VecHIPByProduct bp; struct _VecOps base; /* the Seq/MPI table the base constructor filled */ } Vec_HIP;
#define VHIP(v) ((Vec_HIP*)(v)->data)

static PetscErrorCode VecHIPSyncToGPU(Vec v)       /* role of VecCUSPCopyToGPU veccusp.cu:90 */
{
  Vec_HIP *h = VHIP(v);
  PetscFunctionBegin;
  if (h->valid == VHIP_CPU) { CHKERRPHIP(phip_memcpy_h2d(hipctx,h->d,h->mpi.array,sizeof(PetscScalar)*v->map->n)); h->valid = VHIP_BOTH; }
  PetscFunctionReturn(0);
}
static PetscErrorCode VecHIPSyncToCPU(Vec v)       /* role of VecCUSPCopyFromGPU veccusp.cu:152 */
{
  Vec_HIP *h = VHIP(v);
  PetscFunctionBegin;
  if (h->valid == VHIP_GPU) { CHKERRPHIP(phip_memcpy_d2h(hipctx,h->mpi.array,h->d,sizeof(PetscScalar)*v->map->n)); h->valid = VHIP_BOTH; }
  PetscFunctionReturn(0);
}
PetscErrorCode VecHIPGetArrayRead(Vec v,const PetscScalar **a)  { PetscErrorCode ierr; PetscFunctionBegin; ierr = VecHIPSyncToGPU(v);CHKERRQ(ierr); *a = VHIP(v)->d; PetscFunctionReturn(0); }
PetscErrorCode VecHIPRestoreArrayRead(Vec v,const PetscScalar **a) { PetscFunctionBegin; *a = NULL; PetscFunctionReturn(0); }
PetscErrorCode VecHIPGetArrayWrite(Vec v,PetscScalar **a)       { PetscFunctionBegin; *a = VHIP(v)->d; PetscFunctionReturn(0); } /* contents will be overwritten */
PetscErrorCode VecHIPGetArrayReadWrite(Vec v,PetscScalar **a)   { PetscErrorCode ierr; PetscFunctionBegin; ierr = VecHIPSyncToGPU(v);CHKERRQ(ierr); *a = VHIP(v)->d; PetscFunctionReturn(0); }
PetscErrorCode VecHIPRestoreArrayWrite(Vec v,PetscScalar **a)   /* device copy is now the valid one; bump the state (veccusp.cu:2011-2013) */
{
  PetscErrorCode ierr;
  PetscFunctionBegin;
  VHIP(v)->valid = VHIP_GPU; *a = NULL;
  if (hipsync) CHKERRPHIP(phip_sync(hipctx));                    /* WaitForGPU, cuspvecimpl.h:66 */
  ierr = PetscObjectStateIncrease((PetscObject)v);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
/* host accessors dispatched because v->petscnative is PETSC_FALSE (rvector.c:1661-1663,1711-1715,1861-1863,1894-1898) */
static PetscErrorCode VecGetArray_HIP(Vec v,PetscScalar **a)        { PetscErrorCode ierr; PetscFunctionBegin; ierr = VecHIPSyncToCPU(v);CHKERRQ(ierr); *a = VHIP(v)->mpi.array; PetscFunctionReturn(0); }
static PetscErrorCode VecRestoreArray_HIP(Vec v,PetscScalar **a)    { PetscFunctionBegin; VHIP(v)->valid = VHIP_CPU; PetscFunctionReturn(0); }
static PetscErrorCode VecGetArrayRead_HIP(Vec v,const PetscScalar **a) { PetscErrorCode ierr; PetscFunctionBegin; ierr = VecHIPSyncToCPU(v);CHKERRQ(ierr); *a = VHIP(v)->mpi.array; PetscFunctionReturn(0); }
static PetscErrorCode VecRestoreArrayRead_HIP(Vec v,const PetscScalar **a) { PetscFunctionBegin; PetscFunctionReturn(0); }

#define VEC_RW(y,py)  ierr = VecHIPGetArrayReadWrite(y,&py);CHKERRQ(ierr)
#define VEC_R(x,px)   ierr = VecHIPGetArrayRead(x,&px);CHKERRQ(ierr)
#define VEC_W(w,pw)   ierr = VecHIPGetArrayWrite(w,&pw);CHKERRQ(ierr)
#define VEC_DONE(y,py) ierr = VecHIPRestoreArrayWrite(y,&py);CHKERRQ(ierr)

/* ---- by-product cache ---- */
static PetscErrorCode VecHIPByProductFetch(Vec w,PetscBool *ok)
{
  VecHIPByProduct *b = &VHIP(w)->bp;
  PetscFunctionBegin;
  *ok = PETSC_FALSE;
  if (!b->havehost) {
    if (phip_ticket_wait(hipctx,&b->ticket,b->host)) { b->live = PETSC_FALSE; PetscFunctionReturn(0); }   /* expires identically on every rank */
    if (VHIP(w)->ismpi) CHKERRPHIP(phip_comm_allreduce_host(hipcomm,b->host,b->ticket.nvals,0));             /* the ranks meet on the host */
    b->havehost = PETSC_TRUE;
  }
  *ok = PETSC_TRUE;
  PetscFunctionReturn(0);
}
static PetscErrorCode VecHIPByProductDot(Vec x,Vec y,PetscScalar *z,PetscBool *hit)
{
  Vec              pair[2][2];
  PetscInt         q,i;
  PetscErrorCode   ierr;
  PetscObjectState sw,so;
  PetscObjectId    ido;

  PetscFunctionBegin;
  *hit = PETSC_FALSE;
  pair[0][0] = x; pair[0][1] = y; pair[1][0] = y; pair[1][1] = x;
  for (q=0; q<2 && !*hit; q++) {
    Vec              w = pair[q][0],o = pair[q][1];
    VecHIPByProduct *b = &VHIP(w)->bp;
    ierr = PetscObjectStateGet((PetscObject)w,&sw);CHKERRQ(ierr);
    ierr = PetscObjectStateGet((PetscObject)o,&so);CHKERRQ(ierr);
    ierr = PetscObjectGetId((PetscObject)o,&ido);CHKERRQ(ierr);
    if (!b->kind || !b->live || b->selfstate != sw) continue;
    for (i=0; i<2; i++) {
      if (b->partner[i] != ido || b->partnerstate[i] != so) continue;
      ierr = VecHIPByProductFetch(w,hit);CHKERRQ(ierr);
      if (*hit) *z = b->host[b->kind == 1 ? 1+i : 0];
      break;
    }
  }
  PetscFunctionReturn(0);
}
static PetscErrorCode VecHIPByProductRecord(Vec w,int kind,const phip_ticket *t,Vec p0,Vec p1)
{
  VecHIPByProduct *b = &VHIP(w)->bp;
  PetscErrorCode   ierr;
  PetscFunctionBegin;
  b->kind = kind; b->live = PETSC_TRUE; b->havehost = PETSC_FALSE; b->ticket = *t;
  ierr = PetscObjectStateGet((PetscObject)w,&b->selfstate);CHKERRQ(ierr);   /* call AFTER the write accessor was restored */
  ierr = PetscObjectGetId((PetscObject)p0,&b->partner[0]);CHKERRQ(ierr);
  ierr = PetscObjectStateGet((PetscObject)p0,&b->partnerstate[0]);CHKERRQ(ierr);
  b->partner[1] = 0; b->partnerstate[1] = 0;
  if (p1) { ierr = PetscObjectGetId((PetscObject)p1,&b->partner[1]);CHKERRQ(ierr); ierr = PetscObjectStateGet((PetscObject)p1,&b->partnerstate[1]);CHKERRQ(ierr); }
  PetscFunctionReturn(0);
}

/* ---- the _VecOps slots (include/petsc/private/vecimpl.h:20-96) ---- */
static PetscErrorCode VecDot_HIP(Vec x,Vec y,PetscScalar *z)                  /* bvec1.c:12-30 / pbvec.c:9-19 */
{
  PetscErrorCode     ierr;
  const PetscScalar *px,*py;
  PetscBool          hit;
  PetscFunctionBegin;
  ierr = VecHIPByProductDot(x,y,z,&hit);CHKERRQ(ierr);
  if (hit) PetscFunctionReturn(0);
  VEC_R(x,px); VEC_R(y,py);
  if (!VHIP(x)->ismpi) CHKERRPHIP(phip_vec_dot(hipctx,x->map->n,px,py,z));
  else {                                                      /* VecDot_MPI pbvec.c:9-19: local dot, then the ranks' values are summed */
    const double *ys[1];
    ys[0] = py;
    CHKERRPHIP(phip_vec_reduce_local_host(hipctx,0,x->map->n,px,1,ys,z));
    CHKERRPHIP(phip_comm_allreduce_host(hipcomm,z,1,0));       /* shared-memory ring of the node; MPI_Allreduce(MPI_IN_PLACE,z,1,MPIU_SCALAR,MPIU_SUM,comm) serves as well */
  }
  ierr = PetscLogFlops(2.0*x->map->n-1);CHKERRQ(ierr);     /* bvec1.c:27 */
  PetscFunctionReturn(0);
}
static PetscErrorCode VecNorm_HIP(Vec x,NormType type,PetscReal *z)           /* bvec2.c:233-271 / pvec2.c:49-85 */
{
  PetscErrorCode     ierr;
  const PetscScalar *px;
  VecHIPByProduct   *b = &VHIP(x)->bp;
  PetscObjectState   sx;
  PetscBool          hit = PETSC_FALSE;
  PetscFunctionBegin;
  ierr = PetscObjectStateGet((PetscObject)x,&sx);CHKERRQ(ierr);
  if ((type == NORM_2 || type == NORM_FROBENIUS) && b->kind == 1 && b->live && b->selfstate == sx) {
    ierr = VecHIPByProductFetch(x,&hit);CHKERRQ(ierr);
    if (hit) { *z = PetscSqrtReal(b->host[0]); PetscFunctionReturn(0); }
  }
  VEC_R(x,px);
  if (!VHIP(x)->ismpi) CHKERRPHIP(phip_vec_norm(hipctx,x->map->n,px,(int)type,z));
  else {                                                      /* VecNorm_MPI pvec2.c:49-85 */
    double r[2];
    int    kind = (type == NORM_1) ? 2 : (type == NORM_INFINITY) ? 3 : (type == NORM_1_AND_2) ? 4 : 1;
    CHKERRPHIP(phip_vec_reduce_local_host(hipctx,kind,x->map->n,px,1,NULL,r));
    CHKERRPHIP(phip_comm_allreduce_host(hipcomm,r,kind == 4 ? 2 : 1,kind == 3 ? 1 : 0));
    if (kind == 1) z[0] = PetscSqrtReal(r[0]); else if (kind == 4) { z[0] = r[0]; z[1] = PetscSqrtReal(r[1]); } else z[0] = r[0];
  }
  PetscFunctionReturn(0);
}
static PetscErrorCode VecMDot_HIP(Vec x,PetscInt nv,const Vec y[],PetscScalar *z)   /* dvec2.c:98-294 / pvec2.c:10-25 */
{
  PetscErrorCode     ierr;
  const PetscScalar *px,*py[PHIP_MAX_MVEC];
  PetscInt           k;
  PetscFunctionBegin;
  if (nv > PHIP_MAX_MVEC) SETERRQ2(PETSC_COMM_SELF,PETSC_ERR_SUP,"VecMDot of %D vectors (max %d)",nv,PHIP_MAX_MVEC);
  VEC_R(x,px);
  for (k=0; k<nv; k++) { VEC_R(y[k],py[k]); }
  if (!VHIP(x)->ismpi) CHKERRPHIP(phip_vec_mdot(hipctx,x->map->n,px,nv,py,z));
  else {                                                      /* VecMDot_MPI pvec2.c:10-25: one all-reduce of nv scalars */
    CHKERRPHIP(phip_vec_reduce_local_host(hipctx,0,x->map->n,px,nv,py,z));
    CHKERRPHIP(phip_comm_allreduce_host(hipcomm,z,nv,0));
  }
  ierr = PetscLogFlops(PetscMax(nv*(2.0*x->map->n-1),0.0));CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecAXPY_HIP(Vec y,PetscScalar a,Vec x)                  /* bvec1.c:76-95 */
{
  PetscErrorCode ierr; const PetscScalar *px; PetscScalar *py;
  PetscFunctionBegin;
  if (a == 0.0) PetscFunctionReturn(0);
  VEC_R(x,px); VEC_RW(y,py);
  CHKERRPHIP(phip_vec_axpy(hipctx,y->map->n,py,a,px));
  VEC_DONE(y,py);
  ierr = PetscLogFlops(2.0*y->map->n);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecAYPX_HIP(Vec y,PetscScalar a,Vec x)                  /* dvec2.c:644-686 */
{
  PetscErrorCode ierr; const PetscScalar *px; PetscScalar *py;
  PetscFunctionBegin;
  VEC_R(x,px); VEC_RW(y,py);
  CHKERRPHIP(phip_vec_aypx(hipctx,y->map->n,py,a,px));
  VEC_DONE(y,py);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecWAXPY_HIP(Vec w,PetscScalar a,Vec x,Vec y)           /* dvec2.c:697-730 */
{
  PetscErrorCode ierr; const PetscScalar *px,*py; PetscScalar *pw;
  PetscFunctionBegin;
  VEC_R(x,px); VEC_R(y,py); VEC_W(w,pw);
  CHKERRPHIP(phip_vec_waxpy(hipctx,w->map->n,pw,a,px,py));
  VEC_DONE(w,pw);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecMAXPY_HIP(Vec y,PetscInt nv,const PetscScalar *alpha,Vec *x)   /* dvec2.c:570-638 */
{
  PetscErrorCode ierr; const PetscScalar *px[PHIP_MAX_MVEC]; PetscScalar *py; PetscInt k,k0;
  PetscFunctionBegin;
  VEC_RW(y,py);
  for (k0=0; k0<nv; k0+=PHIP_MAX_MVEC) {
    const PetscInt g = PetscMin(nv-k0,PHIP_MAX_MVEC);
    for (k=0; k<g; k++) { VEC_R(x[k0+k],px[k]); }
    CHKERRPHIP(phip_vec_maxpy(hipctx,y->map->n,py,g,alpha+k0,px));
  }
  VEC_DONE(y,py);
  ierr = PetscLogFlops(nv*2.0*y->map->n);CHKERRQ(ierr);   /* dvec2.c:582 */
  PetscFunctionReturn(0);
}
static PetscErrorCode VecPointwiseMult_HIP(Vec w,Vec x,Vec y)                 /* bvec2.c:84-110 */
{
  PetscErrorCode ierr; const PetscScalar *px,*py; PetscScalar *pw;
  PetscFunctionBegin;
  if (w != x && w != y && x != y) {                            /* PCApply_Jacobi's shape (jacobi.c:288): fuse the reductions KSPCG asks for next */
    phip_ticket t;
    VEC_R(x,px); VEC_R(y,py); VEC_W(w,pw);
    CHKERRPHIP(phip_vec_pointwise_mult_reduce(hipctx,w->map->n,pw,px,py,1,&t));
    VEC_DONE(w,pw);
    ierr = VecHIPByProductRecord(w,1,&t,x,y);CHKERRQ(ierr);
  } else {
    VEC_R(x,px); VEC_R(y,py);
    if (w == x || w == y) { VEC_RW(w,pw); } else { VEC_W(w,pw); }
    CHKERRPHIP(phip_vec_pointwise_mult(hipctx,w->map->n,pw,px,py));
    VEC_DONE(w,pw);
  }
  ierr = PetscLogFlops(w->map->n);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecScale_HIP(Vec x,PetscScalar a) { PetscErrorCode ierr; PetscScalar *px; PetscFunctionBegin; VEC_RW(x,px); CHKERRPHIP(phip_vec_scale(hipctx,x->map->n,px,a)); VEC_DONE(x,px); PetscFunctionReturn(0); }  /* bvec1.c:55-72 */
static PetscErrorCode VecSet_HIP(Vec x,PetscScalar a)   { PetscErrorCode ierr; PetscScalar *px; PetscFunctionBegin; VEC_W(x,px); CHKERRPHIP(phip_vec_set(hipctx,x->map->n,px,a)); VEC_DONE(x,px); PetscFunctionReturn(0); }     /* dvec2.c:551-566 */
static PetscErrorCode VecCopy_HIP(Vec x,Vec y)                                 /* bvec2.c:192-207 */
{
  PetscErrorCode ierr; const PetscScalar *px; PetscScalar *py;
  PetscFunctionBegin;
  if (x == y) PetscFunctionReturn(0);
  VEC_R(x,px); VEC_W(y,py);
  CHKERRPHIP(phip_vec_copy(hipctx,x->map->n,px,py));
  VEC_DONE(y,py);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecAXPBY_HIP(Vec y,PetscScalar a,PetscScalar b,Vec x)    /* bvec1.c:99-133 */
{
  PetscErrorCode ierr; const PetscScalar *px; PetscScalar *py;
  PetscFunctionBegin;
  VEC_R(x,px); VEC_RW(y,py);
  CHKERRPHIP(phip_vec_axpby(hipctx,y->map->n,py,a,b,px));
  VEC_DONE(y,py);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecAXPBYPCZ_HIP(Vec z,PetscScalar a,PetscScalar b,PetscScalar c,Vec x,Vec y)   /* bvec1.c:137-165 */
{
  PetscErrorCode ierr; const PetscScalar *px,*py; PetscScalar *pz;
  PetscFunctionBegin;
  VEC_R(x,px); VEC_R(y,py); VEC_RW(z,pz);
  CHKERRPHIP(phip_vec_axpbypcz(hipctx,z->map->n,pz,a,b,c,px,py));
  VEC_DONE(z,pz);
  ierr = PetscLogFlops(5.0*z->map->n);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecPointwiseDivide_HIP(Vec w,Vec x,Vec y)                /* bvec2.c:114-132 */
{
  PetscErrorCode ierr; const PetscScalar *px,*py; PetscScalar *pw;
  PetscFunctionBegin;
  VEC_R(x,px); VEC_R(y,py);
  if (w == x || w == y) { VEC_RW(w,pw); } else { VEC_W(w,pw); }
  CHKERRPHIP(phip_vec_pointwise_divide(hipctx,w->map->n,pw,px,py));
  VEC_DONE(w,pw);
  ierr = PetscLogFlops(w->map->n);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecSwap_HIP(Vec x,Vec y)                                 /* bvec2.c:211-227 */
{
  PetscErrorCode ierr; PetscScalar *px,*py;
  PetscFunctionBegin;
  if (x == y) PetscFunctionReturn(0);
  VEC_RW(x,px); VEC_RW(y,py);
  CHKERRPHIP(phip_vec_swap(hipctx,x->map->n,px,py));
  VEC_DONE(x,px); VEC_DONE(y,py);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecReciprocal_HIP(Vec x)                                 /* vinv.c:1213-1227 */
{
  PetscErrorCode ierr; PetscScalar *px;
  PetscFunctionBegin;
  VEC_RW(x,px);
  CHKERRPHIP(phip_vec_reciprocal(hipctx,x->map->n,px));
  VEC_DONE(x,px);
  PetscFunctionReturn(0);
}
static PetscErrorCode VecDotNorm2_HIP(Vec s,Vec t,PetscScalar *dp,PetscScalar *nm)   /* vinv.c:1388-1390: dp = s.t, nm = t.t, one fused pass */
{
  PetscErrorCode ierr; Vec ys[2]; PetscScalar z[2];
  PetscFunctionBegin;
  ys[0] = s; ys[1] = t;
  ierr = VecMDot_HIP(t,2,ys,z);CHKERRQ(ierr);
  *dp = z[0]; *nm = z[1];
  PetscFunctionReturn(0);
}
static PetscErrorCode VecConjugate_HIP(Vec x) { PetscFunctionBegin; PetscFunctionReturn(0); }   /* real scalars */
/* placearray / replacearray / resetarray (veccusp.cu:1974-1976): bring the host copy up to date, let the Seq/MPI routine swap the host
   pointer (bvec2.c:687-735, pbvec.c:60-97), and mark the host copy as the valid one so the next device op uploads it */
static PetscErrorCode VecPlaceArray_HIP(Vec v,const PetscScalar *a)
{
  PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = VecHIPSyncToCPU(v);CHKERRQ(ierr);
  ierr = (*VHIP(v)->base.placearray)(v,a);CHKERRQ(ierr);         /* VecPlaceArray_Seq bvec2.c:687 / VecPlaceArray_MPI pbvec.c:39 */
  VHIP(v)->valid = VHIP_CPU;
  PetscFunctionReturn(0);
}
static PetscErrorCode VecResetArray_HIP(Vec v)
{
  PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = VecHIPSyncToCPU(v);CHKERRQ(ierr);
  ierr = (*VHIP(v)->base.resetarray)(v);CHKERRQ(ierr);
  VHIP(v)->valid = VHIP_CPU;
  PetscFunctionReturn(0);
}
static PetscErrorCode VecReplaceArray_HIP(Vec v,const PetscScalar *a)
{
  PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = VecHIPSyncToCPU(v);CHKERRQ(ierr);
  if (!VHIP(v)->base.replacearray) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_SUP,"VecReplaceArray on VECMPIHIP"); /* Vec_MPI has no replacearray either (pbvec.c:101-171) */
  ierr = (*VHIP(v)->base.replacearray)(v,a);CHKERRQ(ierr);
  VHIP(v)->valid = VHIP_CPU;
  PetscFunctionReturn(0);
}

static PetscErrorCode VecDestroy_HIP(Vec v)
{
  PetscErrorCode ierr;
  Vec_HIP       *h = VHIP(v);
  PetscFunctionBegin;
  if (h->d) CHKERRPHIP(phip_free(hipctx,h->d));
  h->d = NULL;
  ierr = (*h->base.destroy)(v);CHKERRQ(ierr);                    /* VecDestroy_Seq bvec2.c:739 / VecDestroy_MPI pdvec.c:11: frees the host array and v->data */
  PetscFunctionReturn(0);
}
static PetscErrorCode VecDuplicate_HIP(Vec win,Vec *V)      /* keeps KSPCreateVecs -> VecDuplicateVecs (iterativ.c:993) on the device */
{
  PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = VecCreate(PetscObjectComm((PetscObject)win),V);CHKERRQ(ierr);
  ierr = PetscLayoutReference(win->map,&(*V)->map);CHKERRQ(ierr);
  ierr = VecSetType(*V,((PetscObject)win)->type_name);CHKERRQ(ierr);
  ierr = PetscObjectListDuplicate(((PetscObject)win)->olist,&((PetscObject)(*V))->olist);CHKERRQ(ierr);
  ierr = PetscFunctionListDuplicate(((PetscObject)win)->qlist,&((PetscObject)(*V))->qlist);CHKERRQ(ierr);
  (*V)->stash.ignorenegidx = win->stash.ignorenegidx;
  PetscFunctionReturn(0);
}

static PetscErrorCode VecCreate_HIP_Private(Vec V,PetscBool ismpi,const char *type)
{
  PetscErrorCode ierr;
  Vec_HIP       *h;
  PetscFunctionBegin;
  ierr = PetscHIPInitialize();CHKERRQ(ierr);
  /* the public base constructor: layout, zeroed host array, Seq/MPI ops table (bvec3.c:VecCreate_Seq -> VecCreate_Seq_Private bvec2.c:873;
     pbvec.c:VecCreate_MPI -> VecCreate_MPI_Private pbvec.c:183) */
  if (ismpi) { ierr = VecCreate_MPI(V);CHKERRQ(ierr); } else { ierr = VecCreate_Seq(V);CHKERRQ(ierr); }
  /* grow the data struct: Vec_Seq / Vec_MPI both start with VECHEADER, Vec_HIP embeds the larger one */
  ierr = PetscNewLog(V,&h);CHKERRQ(ierr);
  ierr = PetscMemcpy(&h->mpi,V->data,ismpi ? sizeof(Vec_MPI) : sizeof(Vec_Seq));CHKERRQ(ierr);
  ierr = PetscFree(V->data);CHKERRQ(ierr);
  V->data = (void*)h;
  ierr = PetscMemcpy(&h->base,V->ops,sizeof(struct _VecOps));CHKERRQ(ierr);
  h->ismpi = ismpi; h->valid = VHIP_BOTH;
  CHKERRPHIP(phip_malloc(hipctx,sizeof(PetscScalar)*V->map->n,(void**)&h->d));
  CHKERRPHIP(phip_vec_set(hipctx,V->map->n,h->d,0.0));
  V->petscnative = PETSC_FALSE;                                   /* host array accessors go through the ops below */
  V->ops->getarray = VecGetArray_HIP;          V->ops->restorearray = VecRestoreArray_HIP;
  V->ops->getarrayread = VecGetArrayRead_HIP;  V->ops->restorearrayread = VecRestoreArrayRead_HIP;
  V->ops->dot = VecDot_HIP;     V->ops->tdot = VecDot_HIP;      V->ops->norm = VecNorm_HIP;   V->ops->mdot = VecMDot_HIP;  V->ops->mtdot = VecMDot_HIP;
  V->ops->dot_local = VecDot_HIP; V->ops->tdot_local = VecDot_HIP; V->ops->norm_local = VecNorm_HIP; V->ops->mdot_local = VecMDot_HIP;
  V->ops->axpy = VecAXPY_HIP;   V->ops->aypx = VecAYPX_HIP;     V->ops->waxpy = VecWAXPY_HIP; V->ops->maxpy = VecMAXPY_HIP;
  V->ops->pointwisemult = VecPointwiseMult_HIP; V->ops->scale = VecScale_HIP; V->ops->set = VecSet_HIP; V->ops->copy = VecCopy_HIP;
  V->ops->axpby = VecAXPBY_HIP; V->ops->axpbypcz = VecAXPBYPCZ_HIP; V->ops->pointwisedivide = VecPointwiseDivide_HIP; V->ops->swap = VecSwap_HIP;
  V->ops->reciprocal = VecReciprocal_HIP; V->ops->dotnorm2 = VecDotNorm2_HIP; V->ops->conjugate = VecConjugate_HIP;
  V->ops->placearray = VecPlaceArray_HIP; V->ops->resetarray = VecResetArray_HIP; V->ops->replacearray = VecReplaceArray_HIP;
  V->ops->duplicate = VecDuplicate_HIP;        V->ops->destroy = VecDestroy_HIP;
  /* every other slot keeps the Seq/MPI routine, which reaches the data through VecGetArray[Read] and therefore through the accessors above */
  ierr = PetscObjectChangeTypeName((PetscObject)V,type);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
PETSC_EXTERN PetscErrorCode VecCreate_SeqHIP(Vec V)
{
  PetscErrorCode ierr; PetscMPIInt size;
  PetscFunctionBegin;
  ierr = MPI_Comm_size(PetscObjectComm((PetscObject)V),&size);CHKERRQ(ierr);
  if (size > 1) SETERRQ(PETSC_COMM_SELF,PETSC_ERR_ARG_WRONG,"Cannot create VECSEQHIP on more than one process");
  ierr = VecCreate_HIP_Private(V,PETSC_FALSE,VECSEQHIP);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
PETSC_EXTERN PetscErrorCode VecCreate_MPIHIP(Vec V) { PetscErrorCode ierr; PetscFunctionBegin; ierr = VecCreate_HIP_Private(V,PETSC_TRUE,VECMPIHIP);CHKERRQ(ierr); PetscFunctionReturn(0); }
PETSC_EXTERN PetscErrorCode VecCreate_HIP(Vec V)            /* mpicusp.cu:237-250 */
{
  PetscErrorCode ierr; PetscMPIInt size;
  PetscFunctionBegin;
  ierr = MPI_Comm_size(PetscObjectComm((PetscObject)V),&size);CHKERRQ(ierr);
  ierr = VecSetType(V,size == 1 ? VECSEQHIP : VECMPIHIP);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* ---- Mat ----------------------------------------------------------------------------------------------------- */
typedef PetscErrorCode (*MatAssemblyEndFn)(Mat,MatAssemblyType);
typedef PetscErrorCode (*MatDestroyFn)(Mat);
typedef PetscErrorCode (*MatScaleFn)(Mat,PetscScalar);
typedef PetscErrorCode (*MatDiagonalScaleFn)(Mat,Vec,Vec);
typedef PetscErrorCode (*MatZeroEntriesFn)(Mat);
typedef enum {MAT_HIP_AUTO=0,MAT_HIP_CSR=1,MAT_HIP_SELL=2} MatHIPStorageFormat;   /* phip_mat_set_format; precedent MatCUSPStorageFormat, petscmat.h */
static const char *const MatHIPStorageFormats[] = {"AUTO","CSR","SELL","MatHIPStorageFormat","MAT_HIP_",0};
typedef struct { phip_mat dev,tdev; PetscObjectState state,nonzerostate; MatAssemblyEndFn base_assemblyend; MatDestroyFn base_destroy;
                 MatScaleFn base_scale; MatDiagonalScaleFn base_diagonalscale; MatZeroEntriesFn base_zeroentries;
                 MatHIPStorageFormat format; PetscBool valuedict,diagmask,hotcolumns,optset; } Mat_SeqAIJHIP; /* B->spptr of the Seq type */
typedef struct { phip_halo halo; PetscObjectState nonzerostate; MatAssemblyEndFn base_assemblyend; MatDestroyFn base_destroy; } Mat_MPIAIJHIP;      /* Mat_MPIAIJ.spptr (mpiaij.h:77) */

static PetscErrorCode MatHIPCopyToGPU(Mat A)                /* role of MatCUSPCopyToGPU aijcusp.cu:110-208 */
{
  Mat_SeqAIJ       *a = (Mat_SeqAIJ*)A->data;
  Mat_SeqAIJHIP    *h = (Mat_SeqAIJHIP*)A->spptr;
  PetscObjectState  st;
  PetscErrorCode    ierr;
  PetscFunctionBegin;
  ierr = PetscObjectStateGet((PetscObject)A,&st);CHKERRQ(ierr);
  if (h->dev && h->nonzerostate == A->nonzerostate) {            /* same pattern: values only */
    if (h->state != st) CHKERRPHIP(phip_mat_update_values(h->dev,a->a));
  } else {
    if (h->dev) CHKERRPHIP(phip_mat_destroy(h->dev));
    h->tdev = NULL;                                              /* the transposed copy went with it */
    CHKERRPHIP(phip_mat_create(hipctx,A->rmap->n,A->cmap->n,a->i,a->j,a->a,a->compressedrow.use,a->compressedrow.nrows,
                               a->compressedrow.i,a->compressedrow.rindex,&h->dev));
    h->nonzerostate = A->nonzerostate;
    if (h->optset) {                                             /* -mat_hip_* (MatSetFromOptions_SeqAIJHIP) */
      if (h->format != MAT_HIP_AUTO) CHKERRPHIP(phip_mat_set_format(h->dev,(int)h->format));
      CHKERRPHIP(phip_mat_set_value_dict(h->dev,(int)h->valuedict));
      CHKERRPHIP(phip_mat_set_diag_mask(h->dev,(int)h->diagmask));
      CHKERRPHIP(phip_mat_set_hot_columns(h->dev,(int)h->hotcolumns));
    }
  }
  h->state = st;
  PetscFunctionReturn(0);
}
/* -mat_hip_storage_format auto|csr|sell, -mat_hip_value_dict, -mat_hip_diag_mask, -mat_hip_hot_columns: which of the device layouts of
   include/petschip.h the products may use (all give the same bits).  Precedent: MatSetFromOptions_SeqAIJCUSP, aijcusp.cu:82-105. */
static PetscErrorCode MatSetFromOptions_SeqAIJHIP(PetscOptions *PetscOptionsObject,Mat A)
{
  Mat_SeqAIJHIP *h = (Mat_SeqAIJHIP*)A->spptr;
  PetscErrorCode ierr;
  PetscBool      f0,f1,f2,f3;
  PetscFunctionBegin;
  ierr = PetscOptionsHead(PetscOptionsObject,"SeqAIJHIP options");CHKERRQ(ierr);
  ierr = PetscOptionsEnum("-mat_hip_storage_format","device layout of the SpMV: CSR row blocks, or the sliced layout of stencil matrices","phip_mat_set_format",
                          MatHIPStorageFormats,(PetscEnum)h->format,(PetscEnum*)&h->format,&f0);CHKERRQ(ierr);
  ierr = PetscOptionsBool("-mat_hip_value_dict","name (diagonal, value) pairs by one byte when the matrix holds <= 256 of them","phip_mat_set_value_dict",h->valuedict,&h->valuedict,&f1);CHKERRQ(ierr);
  ierr = PetscOptionsBool("-mat_hip_diag_mask","hold a row as a 32-bit set of pairs when there are <= 32","phip_mat_set_diag_mask",h->diagmask,&h->diagmask,&f2);CHKERRQ(ierr);
  ierr = PetscOptionsBool("-mat_hip_hot_columns","gather x in descending order of column use (power-law matrices)","phip_mat_set_hot_columns",h->hotcolumns,&h->hotcolumns,&f3);CHKERRQ(ierr);
  ierr = PetscOptionsTail();CHKERRQ(ierr);
  if (f0 || f1 || f2 || f3) h->optset = PETSC_TRUE;
  PetscFunctionReturn(0);
}
static PetscErrorCode MatMult_SeqAIJHIP(Mat A,Vec xx,Vec yy)  /* aij.c:1272-1327 */
{
  PetscErrorCode ierr; const PetscScalar *x; PetscScalar *y; Mat_SeqAIJ *a = (Mat_SeqAIJ*)A->data;
  PetscFunctionBegin;
  ierr = MatHIPCopyToGPU(A);CHKERRQ(ierr);
  VEC_R(xx,x); VEC_W(yy,y);
  if (A->rmap->n == A->cmap->n) {                                /* by-product sum x_i y_i for the VecTDot(P,W) of cg.c:207 */
    phip_ticket t;
    CHKERRPHIP(phip_mat_mult_dot(((Mat_SeqAIJHIP*)A->spptr)->dev,x,y,x,1,&t));
    VEC_DONE(yy,y);
    ierr = VecHIPByProductRecord(yy,2,&t,xx,NULL);CHKERRQ(ierr);
  } else {
    CHKERRPHIP(phip_mat_mult(((Mat_SeqAIJHIP*)A->spptr)->dev,x,y));
    VEC_DONE(yy,y);
  }
  ierr = PetscLogFlops(2.0*a->nz - a->nonzerorowcnt);CHKERRQ(ierr);   /* aij.c:1323 */
  PetscFunctionReturn(0);
}
static PetscErrorCode MatMultAdd_SeqAIJHIP(Mat A,Vec xx,Vec yy,Vec zz)   /* aij.c:1440-1499 */
{
  PetscErrorCode ierr; const PetscScalar *x,*y; PetscScalar *z;
  PetscFunctionBegin;
  ierr = MatHIPCopyToGPU(A);CHKERRQ(ierr);
  VEC_R(xx,x);
  if (yy == zz) { VEC_RW(zz,z); y = z; } else { VEC_R(yy,y); VEC_W(zz,z); }
  CHKERRPHIP(phip_mat_multadd(((Mat_SeqAIJHIP*)A->spptr)->dev,x,y,z));
  VEC_DONE(zz,z);
  ierr = PetscLogFlops(2.0*((Mat_SeqAIJ*)A->data)->nz);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
/* transposed products: the first call builds the transposed device copy from the host CSR (phip_mat_transpose_setup); it lives and dies
   with the device matrix and follows phip_mat_update_values */
static PetscErrorCode MatHIPNeedTranspose(Mat A)
{
  Mat_SeqAIJ *a = (Mat_SeqAIJ*)A->data; Mat_SeqAIJHIP *h = (Mat_SeqAIJHIP*)A->spptr; PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = MatHIPCopyToGPU(A);CHKERRQ(ierr);
  if (h->tdev != h->dev) { CHKERRPHIP(phip_mat_transpose_setup(h->dev,a->i,a->j,a->a)); h->tdev = h->dev; }
  PetscFunctionReturn(0);
}
static PetscErrorCode MatMultTransposeAdd_SeqAIJHIP(Mat A,Vec xx,Vec yy,Vec zz)   /* aij.c:1207-1254 */
{
  PetscErrorCode ierr; const PetscScalar *x,*y; PetscScalar *z;
  PetscFunctionBegin;
  ierr = MatHIPNeedTranspose(A);CHKERRQ(ierr);
  VEC_R(xx,x);
  if (yy == zz) { VEC_RW(zz,z); y = z; } else { VEC_R(yy,y); VEC_W(zz,z); }
  CHKERRPHIP(phip_mat_mult_transpose_add(((Mat_SeqAIJHIP*)A->spptr)->dev,x,y,z));
  VEC_DONE(zz,z);
  ierr = PetscLogFlops(2.0*((Mat_SeqAIJ*)A->data)->nz);CHKERRQ(ierr);      /* aij.c:1250 */
  PetscFunctionReturn(0);
}
static PetscErrorCode MatMultTranspose_SeqAIJHIP(Mat A,Vec xx,Vec yy)       /* aij.c:1258-1266 */
{
  PetscErrorCode ierr; const PetscScalar *x; PetscScalar *y;
  PetscFunctionBegin;
  ierr = MatHIPNeedTranspose(A);CHKERRQ(ierr);
  VEC_R(xx,x); VEC_W(yy,y);
  CHKERRPHIP(phip_mat_mult_transpose(((Mat_SeqAIJHIP*)A->spptr)->dev,x,y));
  VEC_DONE(yy,y);
  ierr = PetscLogFlops(2.0*((Mat_SeqAIJ*)A->data)->nz);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
static PetscErrorCode MatGetDiagonal_SeqAIJHIP(Mat A,Vec v)  /* aij.c:1169-1202 */
{
  PetscErrorCode ierr; PetscScalar *pv;
  PetscFunctionBegin;
  ierr = MatHIPCopyToGPU(A);CHKERRQ(ierr);
  VEC_W(v,pv);
  CHKERRPHIP(phip_mat_getdiagonal(((Mat_SeqAIJHIP*)A->spptr)->dev,pv));
  VEC_DONE(v,pv);
  PetscFunctionReturn(0);
}
/* Value updates that keep the nonzero pattern (include/petschip.h): the inherited SeqAIJ routine updates the host CSR, the master copy
   (every inherited op reads a->a directly), and the SAME update runs on the device copy, so that no value crosses PCIe: the state the
   interface wrapper is about to set (it increases the object state once after the op, matrix.c:4850,4903,5327) is recorded as already
   uploaded.  A device copy that is not current is left alone: the next product re-sends the values as before. */
static PetscBool MatHIPDeviceCurrent(Mat A)
{
  Mat_SeqAIJHIP *h = (Mat_SeqAIJHIP*)A->spptr; PetscObjectState st;
  if (PetscObjectStateGet((PetscObject)A,&st)) return PETSC_FALSE;
  return (PetscBool)(h->dev && h->nonzerostate == A->nonzerostate && h->state == st);
}
static PetscErrorCode MatScale_SeqAIJHIP(Mat A,PetscScalar alpha)            /* aij.c:2475-2488 */
{
  Mat_SeqAIJHIP *h = (Mat_SeqAIJHIP*)A->spptr; PetscErrorCode ierr; PetscBool cur = MatHIPDeviceCurrent(A);
  PetscFunctionBegin;
  ierr = (*h->base_scale)(A,alpha);CHKERRQ(ierr);
  if (cur) { CHKERRPHIP(phip_mat_scale(h->dev,alpha)); h->state++; }
  PetscFunctionReturn(0);
}
static PetscErrorCode MatDiagonalScale_SeqAIJHIP(Mat A,Vec ll,Vec rr)        /* aij.c:2200-2235 */
{
  Mat_SeqAIJHIP *h = (Mat_SeqAIJHIP*)A->spptr; PetscErrorCode ierr; PetscBool cur = MatHIPDeviceCurrent(A),lhip = PETSC_FALSE,rhip = PETSC_FALSE;
  const PetscScalar *l = NULL,*r = NULL;
  PetscFunctionBegin;
  ierr = (*h->base_diagonalscale)(A,ll,rr);CHKERRQ(ierr);
  if (ll) { ierr = PetscObjectTypeCompareAny((PetscObject)ll,&lhip,VECSEQHIP,VECMPIHIP,"");CHKERRQ(ierr); }
  if (rr) { ierr = PetscObjectTypeCompareAny((PetscObject)rr,&rhip,VECSEQHIP,VECMPIHIP,"");CHKERRQ(ierr); }
  if (cur && (!ll || lhip) && (!rr || rhip)) {                   /* both vectors live on the device: scale the device copy where it is */
    if (ll) { VEC_R(ll,l); }
    if (rr) { VEC_R(rr,r); }
    CHKERRPHIP(phip_mat_diagonal_scale(h->dev,l,r));
    h->state++;
  }
  PetscFunctionReturn(0);
}
static PetscErrorCode MatZeroEntries_SeqAIJHIP(Mat A)                        /* aij.c:1058-1067 */
{
  Mat_SeqAIJHIP *h = (Mat_SeqAIJHIP*)A->spptr; PetscErrorCode ierr; PetscBool cur = MatHIPDeviceCurrent(A);
  PetscFunctionBegin;
  ierr = (*h->base_zeroentries)(A);CHKERRQ(ierr);
  if (cur) { CHKERRPHIP(phip_mat_zero_entries(h->dev)); h->state++; }
  PetscFunctionReturn(0);
}
/* MatShift: MatShift_Basic (gcreate.c:6-24) goes through MatSetValues + assembly, after which MatHIPCopyToGPU re-sends the values; the
   device-side form (phip_mat_shift) is bound by the stand-alone mirror (petsc_host.cpp: MatShift_AIJHIP), which owns its assembly state. */
static PetscErrorCode MatAssemblyEnd_SeqAIJHIP(Mat A,MatAssemblyType mode)   /* aijcusp.cu:451-462 */
{
  PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = (*((Mat_SeqAIJHIP*)A->spptr)->base_assemblyend)(A,mode);CHKERRQ(ierr);   /* MatAssemblyEnd_SeqAIJ aij.c:968-1024 */
  if (mode == MAT_FLUSH_ASSEMBLY) PetscFunctionReturn(0);
  ierr = MatHIPCopyToGPU(A);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
static PetscErrorCode MatCreateVecs_AIJHIP(Mat mat,Vec *right,Vec *left)     /* aijcusp.cu:287-309 */
{
  PetscErrorCode ierr;
  PetscFunctionBegin;
  if (right) {
    ierr = VecCreate(PetscObjectComm((PetscObject)mat),right);CHKERRQ(ierr);
    ierr = VecSetSizes(*right,mat->cmap->n,PETSC_DETERMINE);CHKERRQ(ierr);
    ierr = VecSetType(*right,VECHIP);CHKERRQ(ierr);
    ierr = PetscLayoutReference(mat->cmap,&(*right)->map);CHKERRQ(ierr);
  }
  if (left) {
    ierr = VecCreate(PetscObjectComm((PetscObject)mat),left);CHKERRQ(ierr);
    ierr = VecSetSizes(*left,mat->rmap->n,PETSC_DETERMINE);CHKERRQ(ierr);
    ierr = VecSetType(*left,VECHIP);CHKERRQ(ierr);
    ierr = PetscLayoutReference(mat->rmap,&(*left)->map);CHKERRQ(ierr);
  }
  PetscFunctionReturn(0);
}
static PetscErrorCode MatDestroy_SeqAIJHIP(Mat A)            /* aijcusp.cu:529-557 */
{
  PetscErrorCode ierr; Mat_SeqAIJHIP *h = (Mat_SeqAIJHIP*)A->spptr; MatDestroyFn base = h->base_destroy;
  PetscFunctionBegin;
  if (h->dev) CHKERRPHIP(phip_mat_destroy(h->dev));
  ierr = PetscFree(A->spptr);CHKERRQ(ierr);                  /* allocated with PetscNew, so MatDestroy's own PetscFree would also be fine */
  ierr = (*base)(A);CHKERRQ(ierr);                           /* MatDestroy_SeqAIJ aij.c:1030 */
  PetscFunctionReturn(0);
}
PETSC_EXTERN PetscErrorCode MatCreate_SeqAIJHIP(Mat B)
{
  PetscErrorCode ierr; Mat_SeqAIJHIP *h;
  PetscFunctionBegin;
  ierr = PetscHIPInitialize();CHKERRQ(ierr);
  ierr = MatCreate_SeqAIJ(B);CHKERRQ(ierr);                                  /* host CSR + every non-accelerated op */
  ((Mat_SeqAIJ*)B->data)->inode.use = PETSC_FALSE;                           /* aijcusp.cu:573 */
  ierr = PetscNewLog(B,&h);CHKERRQ(ierr);
  B->spptr = h;
  h->base_assemblyend = B->ops->assemblyend; h->base_destroy = B->ops->destroy;
  h->base_scale = B->ops->scale; h->base_diagonalscale = B->ops->diagonalscale; h->base_zeroentries = B->ops->zeroentries;
  B->ops->scale = MatScale_SeqAIJHIP; B->ops->diagonalscale = MatDiagonalScale_SeqAIJHIP; B->ops->zeroentries = MatZeroEntries_SeqAIJHIP;
  h->format = MAT_HIP_AUTO; h->valuedict = h->diagmask = h->hotcolumns = PETSC_TRUE; h->optset = PETSC_FALSE;
  B->ops->setfromoptions = MatSetFromOptions_SeqAIJHIP;
  B->ops->mult = MatMult_SeqAIJHIP;               B->ops->multadd = MatMultAdd_SeqAIJHIP;
  B->ops->getdiagonal = MatGetDiagonal_SeqAIJHIP; B->ops->assemblyend = MatAssemblyEnd_SeqAIJHIP;
  B->ops->destroy = MatDestroy_SeqAIJHIP;         B->ops->getvecs = MatCreateVecs_AIJHIP;
  B->ops->multtranspose = MatMultTranspose_SeqAIJHIP; B->ops->multtransposeadd = MatMultTransposeAdd_SeqAIJHIP;
  /* shift, setvalues ... stay the SeqAIJ host routines: they bump the object state, and MatHIPCopyToGPU re-sends the values (only)
     before the next product; scale / diagonalscale / zeroentries run on both copies (above) */
  ierr = PetscObjectChangeTypeName((PetscObject)B,MATSEQAIJHIP);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* MPIAIJHIP: preallocation creates the two blocks as MATSEQAIJHIP (mpiaijcusp.cu:11-50) */
static PetscErrorCode MatMPIAIJSetPreallocation_MPIAIJHIP(Mat B,PetscInt d_nz,const PetscInt d_nnz[],PetscInt o_nz,const PetscInt o_nnz[])
{
  Mat_MPIAIJ *b = (Mat_MPIAIJ*)B->data; PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = PetscLayoutSetUp(B->rmap);CHKERRQ(ierr);
  ierr = PetscLayoutSetUp(B->cmap);CHKERRQ(ierr);
  if (!B->preallocated) {
    ierr = MatCreate(PETSC_COMM_SELF,&b->A);CHKERRQ(ierr);
    ierr = MatSetSizes(b->A,B->rmap->n,B->cmap->n,B->rmap->n,B->cmap->n);CHKERRQ(ierr);
    ierr = MatSetType(b->A,MATSEQAIJHIP);CHKERRQ(ierr);
    ierr = PetscLogObjectParent((PetscObject)B,(PetscObject)b->A);CHKERRQ(ierr);
    ierr = MatCreate(PETSC_COMM_SELF,&b->B);CHKERRQ(ierr);
    ierr = MatSetSizes(b->B,B->rmap->n,B->cmap->N,B->rmap->n,B->cmap->N);CHKERRQ(ierr);
    ierr = MatSetType(b->B,MATSEQAIJHIP);CHKERRQ(ierr);
    ierr = PetscLogObjectParent((PetscObject)B,(PetscObject)b->B);CHKERRQ(ierr);
  }
  ierr = MatSeqAIJSetPreallocation(b->A,d_nz,d_nnz);CHKERRQ(ierr);
  ierr = MatSeqAIJSetPreallocation(b->B,o_nz,o_nnz);CHKERRQ(ierr);
  B->preallocated = PETSC_TRUE;
  PetscFunctionReturn(0);
}
static PetscErrorCode MatAssemblyEnd_MPIAIJHIP(Mat A,MatAssemblyType mode)
{
  Mat_MPIAIJ    *a = (Mat_MPIAIJ*)A->data; Mat_MPIAIJHIP *h = (Mat_MPIAIJHIP*)a->spptr; PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = (*h->base_assemblyend)(A,mode);CHKERRQ(ierr);                         /* MatAssemblyEnd_MPIAIJ mpiaij.c:703-773: stash, A, MatSetUpMultiply, B */
  if (mode == MAT_FLUSH_ASSEMBLY) PetscFunctionReturn(0);
  if (!h->halo || h->nonzerostate != A->nonzerostate) {
    /* harvest the plan the core just built (VecScatter_MPI_General, vecimpl.h:245-273; neighbours ascending, receives contiguous) */
    VecScatter_MPI_General *to = (VecScatter_MPI_General*)a->Mvctx->todata,*from = (VecScatter_MPI_General*)a->Mvctx->fromdata;
    PetscInt                ec;
    if (h->halo) CHKERRPHIP(phip_halo_destroy(h->halo));
    CHKERRPHIP(phip_halo_create(hipcomm,to->n,to->procs,to->starts,to->indices,from->n,from->procs,from->starts,&h->halo));
    ierr = VecGetLocalSize(a->lvec,&ec);CHKERRQ(ierr);
    ierr = VecDestroy(&a->lvec);CHKERRQ(ierr);                                 /* lvec (mmaij.c:104) becomes a device vector */
    ierr = VecCreate(PETSC_COMM_SELF,&a->lvec);CHKERRQ(ierr);
    ierr = VecSetSizes(a->lvec,ec,ec);CHKERRQ(ierr);
    ierr = VecSetType(a->lvec,VECSEQHIP);CHKERRQ(ierr);
    h->nonzerostate = A->nonzerostate;
  }
  PetscFunctionReturn(0);
}
static PetscErrorCode MatMult_MPIAIJHIP(Mat A,Vec xx,Vec yy)   /* mpiaij.c:984-998 with the scatter replaced */
{
  Mat_MPIAIJ    *a = (Mat_MPIAIJ*)A->data; Mat_MPIAIJHIP *h = (Mat_MPIAIJHIP*)a->spptr; PetscErrorCode ierr;
  const PetscScalar *x; PetscScalar *y,*l; phip_ticket t; PetscBool square = (PetscBool)(A->rmap->n == A->cmap->n);
  PetscFunctionBegin;
  ierr = MatHIPCopyToGPU(a->A);CHKERRQ(ierr); ierr = MatHIPCopyToGPU(a->B);CHKERRQ(ierr);
  VEC_R(xx,x); VEC_W(yy,y); VEC_W(a->lvec,l);
  CHKERRPHIP(phip_halo_begin(h->halo,x,l));                                    /* pack + grouped ncclSend/ncclRecv on the communication stream */
  if (square) CHKERRPHIP(phip_mat_mult_dot(((Mat_SeqAIJHIP*)a->A->spptr)->dev,x,y,x,0,&t));
  else        CHKERRPHIP(phip_mat_mult(((Mat_SeqAIJHIP*)a->A->spptr)->dev,x,y));    /* diagonal block overlaps the exchange */
  CHKERRPHIP(phip_halo_end(h->halo));
  if (square) CHKERRPHIP(phip_mat_multadd_dot(((Mat_SeqAIJHIP*)a->B->spptr)->dev,l,y,x,1,&t));
  else        CHKERRPHIP(phip_mat_multadd(((Mat_SeqAIJHIP*)a->B->spptr)->dev,l,y,y));
  VEC_DONE(a->lvec,l); VEC_DONE(yy,y);
  if (square) { ierr = VecHIPByProductRecord(yy,2,&t,xx,NULL);CHKERRQ(ierr); }
  PetscFunctionReturn(0);
}
static PetscErrorCode MatMultAdd_MPIAIJHIP(Mat A,Vec xx,Vec yy,Vec zz)   /* mpiaij.c:1014-1025 */
{
  Mat_MPIAIJ    *a = (Mat_MPIAIJ*)A->data; Mat_MPIAIJHIP *h = (Mat_MPIAIJHIP*)a->spptr; PetscErrorCode ierr;
  const PetscScalar *x,*y; PetscScalar *z,*l;
  PetscFunctionBegin;
  ierr = MatHIPCopyToGPU(a->A);CHKERRQ(ierr); ierr = MatHIPCopyToGPU(a->B);CHKERRQ(ierr);
  VEC_R(xx,x); VEC_W(a->lvec,l);
  if (yy == zz) { VEC_RW(zz,z); y = z; } else { VEC_R(yy,y); VEC_W(zz,z); }
  CHKERRPHIP(phip_halo_begin(h->halo,x,l));
  CHKERRPHIP(phip_mat_multadd(((Mat_SeqAIJHIP*)a->A->spptr)->dev,x,y,z));
  CHKERRPHIP(phip_halo_end(h->halo));
  CHKERRPHIP(phip_mat_multadd(((Mat_SeqAIJHIP*)a->B->spptr)->dev,l,z,z));
  VEC_DONE(a->lvec,l); VEC_DONE(zz,z);
  PetscFunctionReturn(0);
}
/* mpiaij.c:1029-1056 / 1102-1117: lvec = B^T x; reverse scatter with ADD_VALUES begins; the diagonal block's transposed product runs
   meanwhile; the scatter's end adds the ghost contributions (yy == NULL: MatMultTranspose) */
static PetscErrorCode MatMultTransposeAdd_MPIAIJHIP(Mat A,Vec xx,Vec yy,Vec zz)
{
  Mat_MPIAIJ    *a = (Mat_MPIAIJ*)A->data; Mat_MPIAIJHIP *h = (Mat_MPIAIJHIP*)a->spptr; PetscErrorCode ierr;
  const PetscScalar *x,*y = NULL; PetscScalar *z,*l;
  PetscFunctionBegin;
  ierr = MatHIPNeedTranspose(a->A);CHKERRQ(ierr); ierr = MatHIPNeedTranspose(a->B);CHKERRQ(ierr);
  VEC_R(xx,x); VEC_W(a->lvec,l);
  if (yy && yy == zz) { VEC_RW(zz,z); y = z; } else { if (yy) { VEC_R(yy,y); } VEC_W(zz,z); }
  CHKERRPHIP(phip_mat_mult_transpose(((Mat_SeqAIJHIP*)a->B->spptr)->dev,x,l));
  CHKERRPHIP(phip_halo_reverse_begin(h->halo,l));
  if (yy) CHKERRPHIP(phip_mat_mult_transpose_add(((Mat_SeqAIJHIP*)a->A->spptr)->dev,x,y,z));
  else    CHKERRPHIP(phip_mat_mult_transpose(((Mat_SeqAIJHIP*)a->A->spptr)->dev,x,z));
  CHKERRPHIP(phip_halo_reverse_end(h->halo,z));
  VEC_DONE(a->lvec,l); VEC_DONE(zz,z);
  PetscFunctionReturn(0);
}
static PetscErrorCode MatMultTranspose_MPIAIJHIP(Mat A,Vec xx,Vec yy) { return MatMultTransposeAdd_MPIAIJHIP(A,xx,NULL,yy); }
static PetscErrorCode MatDestroy_MPIAIJHIP(Mat A)
{
  Mat_MPIAIJ *a = (Mat_MPIAIJ*)A->data; Mat_MPIAIJHIP *h = (Mat_MPIAIJHIP*)a->spptr; PetscErrorCode ierr; MatDestroyFn base = h->base_destroy;
  PetscFunctionBegin;
  if (h->halo) CHKERRPHIP(phip_halo_destroy(h->halo));
  ierr = PetscFree(a->spptr);CHKERRQ(ierr);
  ierr = (*base)(A);CHKERRQ(ierr);                           /* MatDestroy_MPIAIJ mpiaij.c:1158 */
  PetscFunctionReturn(0);
}
PETSC_EXTERN PetscErrorCode MatCreate_MPIAIJHIP(Mat B)       /* mpiaijcusp.cu:196-226 */
{
  PetscErrorCode ierr; Mat_MPIAIJ *a; Mat_MPIAIJHIP *h;
  PetscFunctionBegin;
  ierr = PetscHIPInitialize();CHKERRQ(ierr);
  ierr = MatCreate_MPIAIJ(B);CHKERRQ(ierr);
  a = (Mat_MPIAIJ*)B->data;
  ierr = PetscNewLog(B,&h);CHKERRQ(ierr);
  a->spptr = h;
  h->base_assemblyend = B->ops->assemblyend; h->base_destroy = B->ops->destroy;
  ierr = PetscObjectComposeFunction((PetscObject)B,"MatMPIAIJSetPreallocation_C",MatMPIAIJSetPreallocation_MPIAIJHIP);CHKERRQ(ierr);
  B->ops->mult = MatMult_MPIAIJHIP;  B->ops->assemblyend = MatAssemblyEnd_MPIAIJHIP;
  B->ops->getvecs = MatCreateVecs_AIJHIP; B->ops->destroy = MatDestroy_MPIAIJHIP;
  B->ops->multadd = MatMultAdd_MPIAIJHIP; B->ops->multtranspose = MatMultTranspose_MPIAIJHIP; B->ops->multtransposeadd = MatMultTransposeAdd_MPIAIJHIP;
  /* getdiagonal is inherited (MatGetDiagonal(a->A), mpiaij.c:1125-1135) and lands in MatGetDiagonal_SeqAIJHIP; scale / diagonalscale / shift
     stay MatScale_MPIAIJ etc., which call the blocks' host routines (mpiaij.c:1139-1148, 2040-2069, 2589-2601) */
  ierr = PetscObjectChangeTypeName((PetscObject)B,MATMPIAIJHIP);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

/* ---- registration: run by PetscDLLibraryOpen on -dll_append (src/sys/dll/dl.c:175-191) ------------------------- */
PETSC_EXTERN PetscErrorCode PetscDLLibraryRegister_petschip_glue(void)
{
  PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = VecRegister(VECSEQHIP,VecCreate_SeqHIP);CHKERRQ(ierr);
  ierr = VecRegister(VECMPIHIP,VecCreate_MPIHIP);CHKERRQ(ierr);
  ierr = VecRegister(VECHIP,VecCreate_HIP);CHKERRQ(ierr);
  ierr = MatRegisterBaseName(MATAIJHIP,MATSEQAIJHIP,MATMPIAIJHIP);CHKERRQ(ierr);   /* matreg.c:184 */
  ierr = MatRegister(MATSEQAIJHIP,MatCreate_SeqAIJHIP);CHKERRQ(ierr);
  ierr = MatRegister(MATMPIAIJHIP,MatCreate_MPIAIJHIP);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
/* the loader derives the symbol from the file name (dl.c:175-191: lib<basename>.so -> PetscDLLibraryRegister_<basename>): the same
   registration for a library that was linked together with the kernels and installed as libpetschip.so */
PETSC_EXTERN PetscErrorCode PetscDLLibraryRegister_petschip(void) { return PetscDLLibraryRegister_petschip_glue(); }
