// libpetschip_host.so -- stand-alone mirror of the PETSc 3.6.1 API slice that drives the sparse-solve hot path
// (include/petschip_host.h).  Host logic only: every numerical operation is a call into libpetschip.so (HIP).
// Cites are file:line under the PETSc 3.6.1 reference tree.
#include "../../../include/petschip_host.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#define PETSC_ERR_MEM 55
#define PETSC_ERR_SUP 56
#define PETSC_ERR_ORDER 58
#define PETSC_ERR_ARG_SIZ 60
#define PETSC_ERR_ARG_IDN 61
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_ERR_FILE_OPEN 65
#define PETSC_ERR_FILE_READ 66
#define PETSC_ERR_FILE_WRITE 67
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_ARG_INCOMP 75
#define PETSC_ERR_LIB 76
#define PETSC_ERR_PLIB 77 /* inconsistent data in this library (petscerror.h) */
#define PETSC_ERR_FILE_UNEXPECTED 79
#define PETSC_ERR_ARG_NULL 85
#define PETSC_ERR_ARG_UNKNOWN_TYPE 86
#define PETSC_ERR_ARG_TYPENOTSET 89

#define MAT_FILE_CLASSID 1211216 /* include/petscmat.h, bin/PetscBinaryIO.py:129-163 */
#define VEC_FILE_CLASSID 1211214

static thread_local char g_herr[640] = "";
static PetscErrorCode seterr(PetscErrorCode code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_herr, sizeof(g_herr), fmt, ap);
  va_end(ap);
  return code;
}
extern "C" const char *PetscHipLastError(void) { return g_herr; }

#define SETERRQ(code, ...) return seterr((code), __VA_ARGS__)
#define CHKERRQ(ierr) do { if (ierr) return (ierr); } while (0)
#define CHKPHIP(call) do { int _rc = (call); if (_rc) return seterr(_rc, "%s", phip_last_error()); } while (0)

/* ================================================================================================ */
struct _p_PetscComm {
  phip_ctx  ctx;
  phip_comm comm;
  int       rank, size;
  bool      is_world;
  double   *d_red; /* device scratch for distributed reductions: PHIP_MAX_MVEC doubles */
  _p_PetscComm *self;
  /* by-product reductions (include/petschip.h): kind 1 = VecPointwiseMult, kind 2 = MatMult.  A kind switches itself off after 8
     productions in a row that nobody consumed (GMRES never asks for them) and on again when a request comes that it would have served */
  bool      host_reduce; /* distributed reductions combine on the host (phip_comm_allreduce_host) instead of ncclAllReduce on the device */
  bool      bp_allowed;
  bool      bp_on[4];
  int       bp_streak[4];
  _p_PetscComm *root; /* the world communicator that owns the switches (self shares them) */
  struct SplitReduction *split; /* VecDotBegin/VecNormBegin ... queue (comb.c), made on first use */
};
#define BP(c) ((c)->root)

extern "C" PetscErrorCode PetscHipInitialize(int device, const char *backend, const void *id128, int nranks, int rank, PetscComm *world)
{
  if (!world || !backend) SETERRQ(PETSC_ERR_ARG_NULL, "PetscHipInitialize: null argument");
  PetscComm w = new _p_PetscComm();
  memset(w, 0, sizeof(*w));
  CHKPHIP(phip_init(device, &w->ctx));
  CHKPHIP(phip_comm_create(w->ctx, backend, id128, nranks, rank, &w->comm));
  w->rank = rank; w->size = nranks; w->is_world = true;
  CHKPHIP(phip_malloc(w->ctx, sizeof(double) * PHIP_MAX_MVEC, (void **)&w->d_red));
  PetscComm s = new _p_PetscComm();
  memset(s, 0, sizeof(*s));
  s->ctx = w->ctx; s->rank = 0; s->size = 1; s->is_world = false; s->d_red = w->d_red;
  CHKPHIP(phip_comm_create(w->ctx, "self", NULL, 1, 0, &s->comm));
  s->self = s; w->self = s;
  w->root = w; s->root = w;
  w->bp_allowed = true; w->bp_on[1] = w->bp_on[2] = w->bp_on[3] = true;
  w->host_reduce = !(getenv("PETSCHIP_DEVICE_ALLREDUCE") && atoi(getenv("PETSCHIP_DEVICE_ALLREDUCE")));
  *world = w;
  return 0;
}

/* Where the ranks' partial results of VecDot/VecNorm/VecMDot (and of the by-products) meet: PETSC_TRUE (default) on the host, through
   the node's shared-memory ring (include/petschip.h, phip_comm_allreduce_host); PETSC_FALSE on the device (ncclAllReduce, then one
   read).  Collective: every rank must make the same choice. */
extern "C" PetscErrorCode PetscHipSetHostReductions(PetscComm comm, PetscBool flg)
{
  if (!comm) SETERRQ(PETSC_ERR_ARG_NULL, "PetscHipSetHostReductions: null communicator");
  BP(comm)->host_reduce = flg != 0;
  return 0;
}

extern "C" PetscErrorCode PetscHipSetByProducts(PetscComm comm, PetscBool flg)
{
  if (!comm) SETERRQ(PETSC_ERR_ARG_NULL, "PetscHipSetByProducts: null communicator");
  BP(comm)->bp_allowed = flg != 0;
  BP(comm)->bp_on[1] = BP(comm)->bp_on[2] = BP(comm)->bp_on[3] = flg != 0;
  BP(comm)->bp_streak[1] = BP(comm)->bp_streak[2] = BP(comm)->bp_streak[3] = 0;
  return 0;
}

/* Set-up collectives supplied by the embedding application (host MPI in PETSc, gloo in the CPU tests); device < 0 gives a
   planning-only context (include/petschip.h) */
extern "C" PetscErrorCode PetscHipInitializeWithCallbacks(int device, int nranks, int rank, phip_allgather_fn allgather, phip_alltoallv_i32_fn alltoallv, void *user, PetscComm *world)
{
  if (!world) SETERRQ(PETSC_ERR_ARG_NULL, "PetscHipInitializeWithCallbacks: null argument");
  PetscComm w = new _p_PetscComm();
  memset(w, 0, sizeof(*w));
  CHKPHIP(phip_init(device, &w->ctx));
  CHKPHIP(phip_comm_create_callbacks(w->ctx, nranks, rank, allgather, alltoallv, user, &w->comm));
  w->rank = rank; w->size = nranks; w->is_world = true;
  CHKPHIP(phip_malloc(w->ctx, sizeof(double) * PHIP_MAX_MVEC, (void **)&w->d_red));
  PetscComm s = new _p_PetscComm();
  memset(s, 0, sizeof(*s));
  s->ctx = w->ctx; s->rank = 0; s->size = 1; s->is_world = false; s->d_red = w->d_red;
  CHKPHIP(phip_comm_create(w->ctx, "self", NULL, 1, 0, &s->comm));
  s->self = s; w->self = s;
  w->root = w; s->root = w;
  w->bp_allowed = false; /* set-up-only communicator: no device traffic, no by-products */
  w->host_reduce = true; /* reductions through the application's all-gather */
  *world = w;
  return 0;
}

static bool ctx_has_device(phip_ctx ctx)
{
  int dev = -1;
  phip_device_info(ctx, &dev, NULL, NULL, NULL, 0);
  return dev >= 0;
}

static void split_free(PetscComm c);
extern "C" PetscErrorCode PetscHipFinalize(PetscComm *world)
{
  if (!world || !*world) return 0;
  PetscComm w = *world;
  split_free(w->self); split_free(w);
  phip_comm_destroy(w->self->comm);
  phip_comm_destroy(w->comm);
  phip_free(w->ctx, w->d_red);
  phip_finalize(w->ctx);
  delete w->self;
  delete w;
  *world = NULL;
  return 0;
}

extern "C" PetscErrorCode PetscCommGetSelf(PetscComm world, PetscComm *self)
{
  if (!world || !self) SETERRQ(PETSC_ERR_ARG_NULL, "PetscCommGetSelf: null argument");
  *self = world->self;
  return 0;
}
extern "C" PetscErrorCode PetscCommRankSize(PetscComm c, int *rank, int *size)
{
  if (!c) SETERRQ(PETSC_ERR_ARG_NULL, "PetscCommRankSize: null communicator");
  if (rank) *rank = c->rank;
  if (size) *size = c->size;
  return 0;
}
extern "C" PetscErrorCode PetscCommGetContext(PetscComm c, phip_ctx *ctx)
{
  if (!c || !ctx) SETERRQ(PETSC_ERR_ARG_NULL, "PetscCommGetContext: null argument");
  *ctx = c->ctx;
  return 0;
}

extern "C" PetscErrorCode PetscCommGetHipComm(PetscComm c, phip_comm *hipcomm)
{
  if (!c || !hipcomm) SETERRQ(PETSC_ERR_ARG_NULL, "PetscCommGetHipComm: null argument");
  *hipcomm = c->comm;
  return 0;
}

/* harness: rows of the synthetic stencil matrices (include/petschip_host.h) */
extern "C" PetscErrorCode PetscHipStencilRows(int kind, PetscInt nx, PetscInt ny, PetscInt nz, int64_t rstart, int64_t rend, PetscInt *ai, PetscInt *aj, PetscScalar *aa)
{
  if (!ai) SETERRQ(PETSC_ERR_ARG_NULL, "PetscHipStencilRows: null row pointer");
  if (kind != 5 && kind != 7 && kind != 27) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "PetscHipStencilRows: kind %d", kind);
  if (nx < 1 || ny < 1 || nz < 1 || (kind == 5 && nz != 1)) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "PetscHipStencilRows: grid %d x %d x %d", nx, ny, nz);
  const int64_t N = (int64_t)nx * ny * nz;
  if (rstart < 0 || rend < rstart || rend > N || N > 2147483647ll) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "PetscHipStencilRows: rows [%lld, %lld) of %lld", (long long)rstart, (long long)rend, (long long)N);
  const bool   box = kind == 27;
  const double dg = kind == 27 ? 26.0 : (kind == 7 ? 6.0 : 4.0);
  int64_t      w = 0;
  ai[0] = 0;
  for (int64_t r = rstart; r < rend; r++) {
    const int i = (int)(r % nx), j = (int)((r / nx) % ny), k = (int)(r / ((int64_t)nx * ny));
    for (int dz = -1; dz <= 1; dz++) {
      if (k + dz < 0 || k + dz >= nz) continue;
      for (int dy = -1; dy <= 1; dy++) {
        if (j + dy < 0 || j + dy >= ny) continue;
        for (int dx = -1; dx <= 1; dx++) {
          if (i + dx < 0 || i + dx >= nx) continue;
          if (!box && (dx != 0) + (dy != 0) + (dz != 0) > 1) continue;
          if (aj) { aj[w] = (PetscInt)(r + dx + (int64_t)nx * (dy + (int64_t)ny * dz)); aa[w] = (dx || dy || dz) ? -1.0 : dg; }
          w++;
        }
      }
    }
    if (w > 2147483647ll) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "PetscHipStencilRows: more than 2^31-1 entries");
    ai[r - rstart + 1] = (PetscInt)w;
  }
  return 0;
}

/* PetscSplitOwnership, src/sys/utils/psplit.c:75-97 */
extern "C" PetscErrorCode PetscSplitOwnership(PetscComm comm, PetscInt *n, PetscInt *N)
{
  if (!comm || !n || !N) SETERRQ(PETSC_ERR_ARG_NULL, "PetscSplitOwnership: null argument");
  if (*N == PETSC_DECIDE && *n == PETSC_DECIDE) SETERRQ(PETSC_ERR_ARG_INCOMP, "Both n and N cannot be PETSC_DECIDE");
  if (*N == PETSC_DECIDE) {
    std::vector<PetscInt> all((size_t)comm->size);
    CHKPHIP(phip_comm_allgather_host(comm->comm, n, all.data(), sizeof(PetscInt)));
    PetscInt s = 0;
    for (int r = 0; r < comm->size; r++) s += all[(size_t)r];
    *N = s;
  } else if (*n == PETSC_DECIDE) {
    *n = *N / comm->size + ((*N % comm->size) > comm->rank); /* psplit.c:88 */
  }
  return 0;
}

/* PetscLayoutSetUp, src/vec/is/utils/pmap.c:133-158: ranges = prefix sums of the local sizes */
static PetscErrorCode layout_setup(PetscComm comm, PetscInt *n, PetscInt *N, std::vector<PetscInt> &range)
{
  PetscErrorCode ierr = PetscSplitOwnership(comm, n, N); CHKERRQ(ierr);
  std::vector<PetscInt> all((size_t)comm->size);
  CHKPHIP(phip_comm_allgather_host(comm->comm, n, all.data(), sizeof(PetscInt)));
  range.assign((size_t)comm->size + 1, 0);
  for (int r = 0; r < comm->size; r++) range[(size_t)r + 1] = range[(size_t)r] + all[(size_t)r];
  if (range[(size_t)comm->size] != *N) SETERRQ(PETSC_ERR_ARG_SIZ, "Sum of local lengths %d does not equal global length %d", range[(size_t)comm->size], *N);
  return 0;
}

/* ================================================================================================ */
/* Vec                                                                                              */
struct _VecOps { /* the slots of include/petsc/private/vecimpl.h:20-96 that this slice fills */
  PetscErrorCode (*duplicate)(Vec, Vec *);
  PetscErrorCode (*dot)(Vec, Vec, PetscScalar *);
  PetscErrorCode (*tdot)(Vec, Vec, PetscScalar *);
  PetscErrorCode (*mdot)(Vec, PetscInt, const Vec[], PetscScalar *);
  PetscErrorCode (*norm)(Vec, NormType, PetscReal *);
  PetscErrorCode (*scale)(Vec, PetscScalar);
  PetscErrorCode (*copy)(Vec, Vec);
  PetscErrorCode (*set)(Vec, PetscScalar);
  PetscErrorCode (*swap)(Vec, Vec);
  PetscErrorCode (*axpy)(Vec, PetscScalar, Vec);
  PetscErrorCode (*axpby)(Vec, PetscScalar, PetscScalar, Vec);
  PetscErrorCode (*maxpy)(Vec, PetscInt, const PetscScalar *, Vec *);
  PetscErrorCode (*aypx)(Vec, PetscScalar, Vec);
  PetscErrorCode (*waxpy)(Vec, PetscScalar, Vec, Vec);
  PetscErrorCode (*axpbypcz)(Vec, PetscScalar, PetscScalar, PetscScalar, Vec, Vec);
  PetscErrorCode (*pointwisemult)(Vec, Vec, Vec);
  PetscErrorCode (*pointwisedivide)(Vec, Vec, Vec);
  PetscErrorCode (*reciprocal)(Vec);
  PetscErrorCode (*destroy)(Vec);
};

#include <atomic>
static std::atomic<uint64_t> g_next_object_id{1};

struct _p_Vec {
  uint64_t    id = g_next_object_id++;
  PetscComm   comm;
  std::string type;
  PetscInt    n, N;
  std::vector<PetscInt> range;
  bool        setup;
  double     *d;          /* device array */
  std::vector<double> h;  /* host mirror handed out by VecGetArray */
  double     *placed = NULL;        /* VecPlaceArray: the caller's array stands in for the host mirror until VecResetArray ... */
  std::vector<double> unplaced;     /* ... which brings these values back (the vector's own array, untouched meanwhile: bvec2.c:687-735) */
  int         host_out;   /* 0 none, 1 read access outstanding, 2 write access outstanding */
  uint64_t    state;      /* PetscObjectState: bumped on every write (rvector.c:1865 et al.) */
  std::vector<PetscInt> stash_i; std::vector<double> stash_v; std::vector<int> stash_mode;
  _VecOps     ops;
  /* by-products of the operation that last wrote this vector: kind 1 (w = x .* y): {sum w^2, sum w x, sum w y}, partners x, y;
     kind 2 (y = A x): {sum x y}, partner x.  Valid while the object states still match (cf. the norm cache, rvector.c:237-247). */
  struct {
    int         kind;
    bool        live, consumed, have_host;
    phip_ticket ticket;
    uint64_t    self_state;
    uint64_t    partner[2]; /* object ids (not pointers: a destroyed partner's address may be reused) */
    uint64_t    partner_state[2];
    double      host[3];
  } bp;
};

static std::map<std::string, PetscErrorCode (*)(Vec)> &VecList()
{
  static std::map<std::string, PetscErrorCode (*)(Vec)> l;
  return l;
}
static void VecRegisterAll();

extern "C" PetscErrorCode VecRegister(const char *sname, PetscErrorCode (*ctor)(Vec))
{
  if (!sname || !ctor) SETERRQ(PETSC_ERR_ARG_NULL, "VecRegister: null argument");
  VecList()[sname] = ctor;
  return 0;
}

extern "C" PetscErrorCode VecCreate(PetscComm comm, Vec *v)
{
  if (!comm || !v) SETERRQ(PETSC_ERR_ARG_NULL, "VecCreate: null argument");
  Vec x = new _p_Vec();
  x->comm = comm; x->n = PETSC_DECIDE; x->N = PETSC_DECIDE; x->setup = false; x->d = NULL; x->host_out = 0; x->state = 0;
  memset(&x->ops, 0, sizeof(x->ops));
  memset(&x->bp, 0, sizeof(x->bp));
  *v = x;
  return 0;
}

extern "C" PetscErrorCode VecSetSizes(Vec v, PetscInt n, PetscInt N)
{
  if (!v) SETERRQ(PETSC_ERR_ARG_NULL, "VecSetSizes: null vector");
  if (N >= 0 && n > N) SETERRQ(PETSC_ERR_ARG_INCOMP, "Local size %d cannot be larger than global size %d", n, N);
  if (v->setup) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Cannot change/reset vector sizes after the type has been set");
  v->n = n; v->N = N;
  return 0;
}

#define VCTX(v) ((v)->comm->ctx)
#define VecCheckReady(v, name) do { if (!(v)) SETERRQ(PETSC_ERR_ARG_NULL, name ": null vector"); if (!(v)->setup) SETERRQ(PETSC_ERR_ARG_TYPENOTSET, name ": vector type/sizes not set"); if ((v)->host_out == 2) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, name ": VecGetArray() is outstanding; call VecRestoreArray() first"); } while (0)
#define VecCheckSame(x, y, name) do { if ((x)->n != (y)->n || (x)->N != (y)->N) SETERRQ(PETSC_ERR_ARG_INCOMP, name ": incompatible vector lengths (local %d vs %d, global %d vs %d)", (x)->n, (y)->n, (x)->N, (y)->N); } while (0)

/* ---- by-product bookkeeping ---- */
/* called by a producing op before it overwrites w: w's previous by-products, if nobody asked for them, count against the kind */
static bool bp_want(Vec w, int kind)
{
  PetscComm r = BP(w->comm);
  if (!r->bp_allowed) return false;
  if (w->bp.kind == kind && w->bp.live) {
    if (w->bp.consumed) r->bp_streak[kind] = 0;
    else if (++r->bp_streak[kind] >= 8) r->bp_on[kind] = false;
  }
  return r->bp_on[kind];
}
static void bp_record(Vec w, int kind, bool live, const phip_ticket *t, Vec p0, Vec p1)
{
  w->bp.kind = kind; w->bp.live = live; w->bp.consumed = false; w->bp.have_host = false;
  if (t) w->bp.ticket = *t;
  w->bp.self_state = w->state + 1; /* the interface wrapper bumps the state once after the op */
  w->bp.partner[0] = p0 ? p0->id : 0; w->bp.partner_state[0] = p0 ? p0->state : 0;
  w->bp.partner[1] = p1 ? p1->id : 0; w->bp.partner_state[1] = p1 ? p1->state : 0;
}
/* bring the (global) values of w's by-products to the host: spin on the pinned slot (one rank) or all-reduce the device slot */
static bool bp_fetch(Vec w)
{
  if (w->bp.have_host) return true;
  const int n = w->bp.ticket.nvals;
  if (w->comm->size == 1) {
    if (phip_ticket_wait(VCTX(w), &w->bp.ticket, w->bp.host)) { w->bp.live = false; return false; }
  } else if (BP(w->comm)->host_reduce) {
    if (phip_ticket_wait(VCTX(w), &w->bp.ticket, w->bp.host)) { w->bp.live = false; return false; } /* expired identically on every rank */
    if (phip_comm_allreduce_host(w->comm->comm, w->bp.host, n, 0)) { w->bp.live = false; return false; }
  } else {
    double *d;
    if (phip_ticket_values(VCTX(w), &w->bp.ticket, &d)) { w->bp.live = false; return false; } /* expired identically on every rank */
    if (phip_comm_allreduce(w->comm->comm, d, n, 0)) { w->bp.live = false; return false; }
    if (phip_read_scalars(VCTX(w), d, n, w->bp.host)) { w->bp.live = false; return false; }
  }
  w->bp.have_host = true; w->bp.consumed = true;
  return true;
}
/* is x . y among the by-products of x or of y?  idx of the value in host[]: kind 1: 1 + partner index, kind 2: 0 */
static bool bp_try_dot(Vec x, Vec y, PetscScalar *val)
{
  Vec pair[2][2] = {{x, y}, {y, x}};
  for (int q = 0; q < 2; q++) {
    Vec w = pair[q][0], o = pair[q][1];
    if (!w->bp.kind || w->bp.self_state != w->state) continue;
    for (int i = 0; i < 2; i++) {
      if (w->bp.partner[i] != o->id || w->bp.partner_state[i] != o->state) continue;
      if (!w->bp.live) { /* it would have been served: switch the kind back on */
        PetscComm r = BP(w->comm);
        if (r->bp_allowed) { r->bp_on[w->bp.kind] = true; r->bp_streak[w->bp.kind] = 0; }
        return false;
      }
      if (!bp_fetch(w)) return false;
      *val = w->bp.host[w->bp.kind == 1 ? 1 + i : 0];
      return true;
    }
  }
  return false;
}
static bool bp_try_norm2(Vec w, PetscReal *val)
{
  if ((w->bp.kind != 1 && w->bp.kind != 3) || w->bp.self_state != w->state) return false; /* kind 3: sum w^2 alone (VecMAXPY) */
  if (!w->bp.live) {
    PetscComm r = BP(w->comm);
    if (r->bp_allowed) { r->bp_on[w->bp.kind] = true; r->bp_streak[w->bp.kind] = 0; }
    return false;
  }
  if (!bp_fetch(w)) return false;
  *val = sqrt(w->bp.host[0]);
  return true;
}

/* ---- HIP implementations of the ops: Seq versions, then the MPI reductions ---- */
static PetscErrorCode VecDuplicate_HIP(Vec v, Vec *w);
static PetscErrorCode VecDestroy_HIP(Vec v) { if (v->d) CHKPHIP(phip_free(VCTX(v), v->d)); v->d = NULL; return 0; }
static PetscErrorCode VecDot_SeqHIP(Vec x, Vec y, PetscScalar *z) { CHKPHIP(phip_vec_dot(VCTX(x), x->n, x->d, y->d, z)); return 0; }
static PetscErrorCode VecNorm_SeqHIP(Vec x, NormType t, PetscReal *z) { CHKPHIP(phip_vec_norm(VCTX(x), x->n, x->d, (int)t, z)); return 0; }
static PetscErrorCode VecMDot_SeqHIP(Vec x, PetscInt nv, const Vec y[], PetscScalar *z)
{
  std::vector<const double *> p((size_t)nv);
  for (PetscInt k = 0; k < nv; k++) p[(size_t)k] = y[k]->d;
  CHKPHIP(phip_vec_mdot(VCTX(x), x->n, x->d, nv, p.data(), z));
  return 0;
}
/* The reductions of the MPI type: local kernel, then the ranks' values are combined (VecDot_MPI pbvec.c:9-19, VecMDot_MPI pvec2.c:10-25,
   VecNorm_MPI pvec2.c:49-85).  Host route: the local kernel publishes to pinned memory and the ranks meet in the node's shared-memory
   ring.  Device route: the local result stays on the device, ncclAllReduce, one read.  kind as in phip_vec_reduce_local. */
static PetscErrorCode mpi_reduce(Vec x, int kind, PetscInt nv, const double *const *p, int nres, int op, double *r)
{
  PetscComm cm = x->comm;
  if (BP(cm)->host_reduce) {
    CHKPHIP(phip_vec_reduce_local_host(VCTX(x), kind, x->n, x->d, nv, p, r));
    CHKPHIP(phip_comm_allreduce_host(cm->comm, r, nres, op));
  } else {
    CHKPHIP(phip_vec_reduce_local(VCTX(x), kind, x->n, x->d, nv, p, cm->d_red));
    CHKPHIP(phip_comm_allreduce(cm->comm, cm->d_red, nres, op));
    CHKPHIP(phip_read_scalars(VCTX(x), cm->d_red, nres, r));
  }
  return 0;
}
static PetscErrorCode VecDot_MPIHIP(Vec x, Vec y, PetscScalar *z)
{
  const double *p[1] = {y->d};
  return mpi_reduce(x, 0, 1, p, 1, 0, z);
}
/* one all-reduce of nv scalars */
static PetscErrorCode VecMDot_MPIHIP(Vec x, PetscInt nv, const Vec y[], PetscScalar *z)
{
  if (nv > PHIP_MAX_MVEC) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "VecMDot: nv=%d exceeds %d", nv, PHIP_MAX_MVEC);
  std::vector<const double *> p((size_t)nv);
  for (PetscInt k = 0; k < nv; k++) p[(size_t)k] = y[k]->d;
  return mpi_reduce(x, 0, nv, p.data(), nv, 0, z);
}
static PetscErrorCode VecNorm_MPIHIP(Vec x, NormType t, PetscReal *z)
{
  double         r[2];
  PetscErrorCode ierr;
  switch (t) {
  case NORM_2: case NORM_FROBENIUS:
    ierr = mpi_reduce(x, 1, 1, NULL, 1, 0, r); CHKERRQ(ierr);
    *z = sqrt(r[0]);
    return 0;
  case NORM_1: return mpi_reduce(x, 2, 1, NULL, 1, 0, z);
  case NORM_INFINITY: return mpi_reduce(x, 3, 1, NULL, 1, 1, z);
  case NORM_1_AND_2:
    ierr = mpi_reduce(x, 4, 1, NULL, 2, 0, r); CHKERRQ(ierr);
    z[0] = r[0]; z[1] = sqrt(r[1]);
    return 0;
  }
  SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "VecNorm: unknown NormType %d", (int)t);
}
static PetscErrorCode VecScale_HIP(Vec x, PetscScalar a) { CHKPHIP(phip_vec_scale(VCTX(x), x->n, x->d, a)); return 0; }
static PetscErrorCode VecCopy_HIP(Vec x, Vec y) { CHKPHIP(phip_vec_copy(VCTX(x), x->n, x->d, y->d)); return 0; }
static PetscErrorCode VecSet_HIP(Vec x, PetscScalar a) { CHKPHIP(phip_vec_set(VCTX(x), x->n, x->d, a)); return 0; }
static PetscErrorCode VecSwap_HIP(Vec x, Vec y) { CHKPHIP(phip_vec_swap(VCTX(x), x->n, x->d, y->d)); return 0; }
static PetscErrorCode VecAXPY_HIP(Vec y, PetscScalar a, Vec x) { CHKPHIP(phip_vec_axpy(VCTX(y), y->n, y->d, a, x->d)); return 0; }
static PetscErrorCode VecAXPBY_HIP(Vec y, PetscScalar a, PetscScalar b, Vec x) { CHKPHIP(phip_vec_axpby(VCTX(y), y->n, y->d, a, b, x->d)); return 0; }
static PetscErrorCode VecMAXPY_HIP(Vec y, PetscInt nv, const PetscScalar *al, Vec *x)
{
  std::vector<const double *> p((size_t)nv);
  for (PetscInt k = 0; k < nv; k++) p[(size_t)k] = x[k]->d;
  /* the orthogonalisation's VecMAXPY is followed by VecNormalize (gmres.c:170-176): its sum of squares leaves with the update */
  if (nv >= 1) {
    const bool want = bp_want(y, 3);
    if (want) {
      phip_ticket t;
      CHKPHIP(phip_vec_maxpy_reduce(VCTX(y), y->n, y->d, nv, al, p.data(), y->comm->size == 1 || BP(y->comm)->host_reduce, &t));
      bp_record(y, 3, true, &t, NULL, NULL);
      return 0;
    }
    bp_record(y, 3, false, NULL, NULL, NULL);
  }
  CHKPHIP(phip_vec_maxpy(VCTX(y), y->n, y->d, nv, al, p.data()));
  return 0;
}
static PetscErrorCode VecAYPX_HIP(Vec y, PetscScalar a, Vec x) { CHKPHIP(phip_vec_aypx(VCTX(y), y->n, y->d, a, x->d)); return 0; }
static PetscErrorCode VecWAXPY_HIP(Vec w, PetscScalar a, Vec x, Vec y) { CHKPHIP(phip_vec_waxpy(VCTX(w), w->n, w->d, a, x->d, y->d)); return 0; }
static PetscErrorCode VecAXPBYPCZ_HIP(Vec z, PetscScalar a, PetscScalar b, PetscScalar g, Vec x, Vec y) { CHKPHIP(phip_vec_axpbypcz(VCTX(z), z->n, z->d, a, b, g, x->d, y->d)); return 0; }
static PetscErrorCode VecPointwiseMult_HIP(Vec w, Vec x, Vec y)
{
  if (w != x && w != y && x != y) {
    const bool want = bp_want(w, 1);
    if (want) {
      phip_ticket t;
      CHKPHIP(phip_vec_pointwise_mult_reduce(VCTX(w), w->n, w->d, x->d, y->d, w->comm->size == 1 || BP(w->comm)->host_reduce, &t));
      bp_record(w, 1, true, &t, x, y);
      return 0;
    }
    bp_record(w, 1, false, NULL, x, y);
  }
  CHKPHIP(phip_vec_pointwise_mult(VCTX(w), w->n, w->d, x->d, y->d));
  return 0;
}
static PetscErrorCode VecPointwiseDivide_HIP(Vec w, Vec x, Vec y) { CHKPHIP(phip_vec_pointwise_divide(VCTX(w), w->n, w->d, x->d, y->d)); return 0; }
static PetscErrorCode VecReciprocal_HIP(Vec x) { CHKPHIP(phip_vec_reciprocal(VCTX(x), x->n, x->d)); return 0; }

static PetscErrorCode vec_alloc(Vec v)
{
  PetscErrorCode ierr = layout_setup(v->comm, &v->n, &v->N, v->range); CHKERRQ(ierr);
  CHKPHIP(phip_malloc(VCTX(v), sizeof(double) * (size_t)v->n, (void **)&v->d));
  if (ctx_has_device(VCTX(v))) CHKPHIP(phip_vec_set(VCTX(v), v->n, v->d, 0.0)); /* new vectors are zeroed (bvec2.c:873 VecCreate_Seq_Private + calloc) */
  v->setup = true;
  return 0;
}

static void vec_fill_ops(Vec v, bool mpi)
{
  v->ops.duplicate = VecDuplicate_HIP; v->ops.destroy = VecDestroy_HIP;
  v->ops.dot = mpi ? VecDot_MPIHIP : VecDot_SeqHIP; v->ops.tdot = v->ops.dot; /* real scalars: TDot == Dot */
  v->ops.mdot = mpi ? VecMDot_MPIHIP : VecMDot_SeqHIP;
  v->ops.norm = mpi ? VecNorm_MPIHIP : VecNorm_SeqHIP;
  v->ops.scale = VecScale_HIP; v->ops.copy = VecCopy_HIP; v->ops.set = VecSet_HIP; v->ops.swap = VecSwap_HIP;
  v->ops.axpy = VecAXPY_HIP; v->ops.axpby = VecAXPBY_HIP; v->ops.maxpy = VecMAXPY_HIP; v->ops.aypx = VecAYPX_HIP;
  v->ops.waxpy = VecWAXPY_HIP; v->ops.axpbypcz = VecAXPBYPCZ_HIP; v->ops.pointwisemult = VecPointwiseMult_HIP;
  v->ops.pointwisedivide = VecPointwiseDivide_HIP; v->ops.reciprocal = VecReciprocal_HIP;
}

/* constructors: the analogue of VecCreate_SeqCUSP veccusp.cu:1714+ / VecCreate_MPICUSP mpicusp.cu:188-233 / VecCreate_CUSP :237-250 */
static PetscErrorCode VecCreate_SeqHIP(Vec v)
{
  if (v->comm->size > 1) SETERRQ(PETSC_ERR_ARG_WRONG, "Cannot create VECSEQHIP on more than one process");
  vec_fill_ops(v, false);
  v->type = VECSEQHIP;
  return vec_alloc(v);
}
static PetscErrorCode VecCreate_MPIHIP(Vec v)
{
  vec_fill_ops(v, true);
  v->type = VECMPIHIP;
  return vec_alloc(v);
}
static PetscErrorCode VecCreate_HIP(Vec v) { return (v->comm->size == 1) ? VecCreate_SeqHIP(v) : VecCreate_MPIHIP(v); }

static void VecRegisterAll()
{
  static bool done = false;
  if (done) return;
  done = true;
  VecRegister(VECSEQHIP, VecCreate_SeqHIP);
  VecRegister(VECMPIHIP, VecCreate_MPIHIP);
  VecRegister(VECHIP, VecCreate_HIP);
}

extern "C" PetscErrorCode VecSetType(Vec v, const char *type)
{
  if (!v || !type) SETERRQ(PETSC_ERR_ARG_NULL, "VecSetType: null argument");
  VecRegisterAll();
  auto it = VecList().find(type);
  if (it == VecList().end()) SETERRQ(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown vector type: %s", type); /* vecreg.c:55 */
  if (v->setup) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "VecSetType: type already set to %s", v->type.c_str());
  if (v->n == PETSC_DECIDE && v->N == PETSC_DECIDE) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "VecSetType: call VecSetSizes() first");
  return (*it->second)(v);
}
extern "C" PetscErrorCode VecGetType(Vec v, const char **type)
{
  if (!v || !type) SETERRQ(PETSC_ERR_ARG_NULL, "VecGetType: null argument");
  *type = v->type.c_str();
  return 0;
}

static PetscErrorCode VecDuplicate_HIP(Vec v, Vec *w)
{
  Vec x = new _p_Vec();
  x->comm = v->comm; x->type = v->type; x->n = v->n; x->N = v->N; x->range = v->range; x->setup = true; x->host_out = 0; x->state = 0;
  x->ops = v->ops; x->d = NULL;
  memset(&x->bp, 0, sizeof(x->bp));
  CHKPHIP(phip_malloc(VCTX(v), sizeof(double) * (size_t)v->n, (void **)&x->d));
  if (ctx_has_device(VCTX(v))) CHKPHIP(phip_vec_set(VCTX(v), x->n, x->d, 0.0));
  *w = x;
  return 0;
}
extern "C" PetscErrorCode VecDuplicate(Vec v, Vec *w)
{
  VecCheckReady(v, "VecDuplicate");
  if (!w) SETERRQ(PETSC_ERR_ARG_NULL, "VecDuplicate: null output");
  return (*v->ops.duplicate)(v, w);
}
/* VecDuplicateVecs_Default, src/vec/vec/interface/vector.c:830: m separate duplicates */
extern "C" PetscErrorCode VecDuplicateVecs(Vec v, PetscInt m, Vec **V)
{
  VecCheckReady(v, "VecDuplicateVecs");
  if (m <= 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "m must be > 0: m = %d", m);
  Vec *a = (Vec *)calloc((size_t)m, sizeof(Vec));
  for (PetscInt i = 0; i < m; i++) { PetscErrorCode ierr = VecDuplicate(v, &a[i]); CHKERRQ(ierr); }
  *V = a;
  return 0;
}
extern "C" PetscErrorCode VecDestroy(Vec *v)
{
  if (!v || !*v) return 0;
  if ((*v)->ops.destroy) { PetscErrorCode ierr = (*(*v)->ops.destroy)(*v); CHKERRQ(ierr); }
  delete *v;
  *v = NULL;
  return 0;
}
extern "C" PetscErrorCode VecDestroyVecs(PetscInt m, Vec **V)
{
  if (!V || !*V) return 0;
  for (PetscInt i = 0; i < m; i++) { PetscErrorCode ierr = VecDestroy(&(*V)[i]); CHKERRQ(ierr); }
  free(*V);
  *V = NULL;
  return 0;
}
extern "C" PetscErrorCode VecGetSize(Vec v, PetscInt *N) { VecCheckReady(v, "VecGetSize"); *N = v->N; return 0; }
extern "C" PetscErrorCode VecGetLocalSize(Vec v, PetscInt *n) { VecCheckReady(v, "VecGetLocalSize"); *n = v->n; return 0; }
extern "C" PetscErrorCode VecGetOwnershipRange(Vec v, PetscInt *low, PetscInt *high)
{
  VecCheckReady(v, "VecGetOwnershipRange");
  if (low) *low = v->range[(size_t)v->comm->rank];
  if (high) *high = v->range[(size_t)v->comm->rank + 1];
  return 0;
}

extern "C" PetscErrorCode VecGetArrayRead(Vec v, const PetscScalar **a)
{
  VecCheckReady(v, "VecGetArrayRead");
  v->h.resize((size_t)v->n + 1);
  double *host = v->placed ? v->placed : v->h.data();
  CHKPHIP(phip_memcpy_d2h(VCTX(v), host, v->d, sizeof(double) * (size_t)v->n));
  v->host_out = 1;
  *a = host;
  return 0;
}
extern "C" PetscErrorCode VecRestoreArrayRead(Vec v, const PetscScalar **a)
{
  if (!v) SETERRQ(PETSC_ERR_ARG_NULL, "VecRestoreArrayRead: null vector");
  v->host_out = 0;
  if (a) *a = NULL;
  return 0;
}
extern "C" PetscErrorCode VecGetArray(Vec v, PetscScalar **a)
{
  const PetscScalar *r;
  PetscErrorCode     ierr = VecGetArrayRead(v, &r); CHKERRQ(ierr);
  v->host_out = 2;
  *a = v->placed ? v->placed : v->h.data();
  return 0;
}
extern "C" PetscErrorCode VecRestoreArray(Vec v, PetscScalar **a)
{
  if (!v) SETERRQ(PETSC_ERR_ARG_NULL, "VecRestoreArray: null vector");
  if (v->host_out != 2) SETERRQ(PETSC_ERR_ORDER, "VecRestoreArray: no VecGetArray() outstanding");
  CHKPHIP(phip_memcpy_h2d(VCTX(v), v->d, v->placed ? v->placed : v->h.data(), sizeof(double) * (size_t)v->n));
  v->host_out = 0;
  v->state++; /* rvector.c:1865 */
  if (a) *a = NULL;
  return 0;
}
/* VecPlaceArray / VecResetArray (rvector.c:1955-2030; ops->placearray/resetarray, veccusp.cu:1974-1976): the vector takes the values of
   the caller's array and hands that array out from VecGetArray until the reset, which restores the vector's own values */
extern "C" PetscErrorCode VecPlaceArray(Vec v, const PetscScalar array[])
{
  VecCheckReady(v, "VecPlaceArray");
  if (!array && v->n) SETERRQ(PETSC_ERR_ARG_NULL, "VecPlaceArray: null array");
  if (v->placed) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "VecPlaceArray() was already called on this vector, without a call to VecResetArray()"); /* bvec2.c:693 */
  if (v->host_out) SETERRQ(PETSC_ERR_ORDER, "VecPlaceArray: VecGetArray() is outstanding");
  v->unplaced.resize((size_t)v->n + 1);
  CHKPHIP(phip_memcpy_d2h(VCTX(v), v->unplaced.data(), v->d, sizeof(double) * (size_t)v->n));
  v->placed = const_cast<PetscScalar *>(array);
  CHKPHIP(phip_memcpy_h2d(VCTX(v), v->d, array, sizeof(double) * (size_t)v->n));
  v->state++;
  return 0;
}
extern "C" PetscErrorCode VecResetArray(Vec v)
{
  VecCheckReady(v, "VecResetArray");
  if (v->host_out) SETERRQ(PETSC_ERR_ORDER, "VecResetArray: VecGetArray() is outstanding");
  if (!v->placed) return 0;
  /* what the device holds now belongs to the placed array (the reference computes in it in place) */
  CHKPHIP(phip_memcpy_d2h(VCTX(v), v->placed, v->d, sizeof(double) * (size_t)v->n));
  CHKPHIP(phip_memcpy_h2d(VCTX(v), v->d, v->unplaced.data(), sizeof(double) * (size_t)v->n));
  v->placed = NULL; v->unplaced.clear();
  v->state++;
  return 0;
}

/* PetscRandom, type rand48 (src/sys/classes/random/impls/rand48/rand48.c:6-42, interface randomc.c:335-350): seed 0x12345678 + 76543*rank,
   srand48(seed) at type set-up, value = width*drand48() + low when an interval is set.  erand48 on the object's own state gives
   drand48's sequence without its process-wide state (ranks may be threads here). */
struct _p_PetscRandom {
  PetscComm      comm;
  unsigned long  seed;
  unsigned short x[3];
  double         low, width;
  bool           iset;
};
extern "C" PetscErrorCode PetscRandomSeed(PetscRandom r)
{
  if (!r) SETERRQ(PETSC_ERR_ARG_NULL, "PetscRandomSeed: null object");
  r->x[0] = 0x330E; r->x[1] = (unsigned short)(r->seed & 0xFFFF); r->x[2] = (unsigned short)((r->seed >> 16) & 0xFFFF); /* srand48(3) */
  return 0;
}
extern "C" PetscErrorCode PetscRandomCreate(PetscComm comm, PetscRandom *r)
{
  if (!comm || !r) SETERRQ(PETSC_ERR_ARG_NULL, "PetscRandomCreate: null argument");
  PetscRandom q = new _p_PetscRandom();
  q->comm = comm; q->low = 0.0; q->width = 1.0; q->iset = false;
  q->seed = 0x12345678ul + 76543ul * (unsigned long)comm->rank; /* randomc.c:347 */
  *r = q;
  return PetscRandomSeed(q); /* PetscRandomSetType, randreg.c:51 */
}
extern "C" PetscErrorCode PetscRandomSetSeed(PetscRandom r, unsigned long seed) { if (!r) SETERRQ(PETSC_ERR_ARG_NULL, "PetscRandomSetSeed: null object"); r->seed = seed; return 0; }
extern "C" PetscErrorCode PetscRandomSetInterval(PetscRandom r, PetscScalar low, PetscScalar high)
{
  if (!r) SETERRQ(PETSC_ERR_ARG_NULL, "PetscRandomSetInterval: null object");
  if (low >= high) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "only low < high: Instead %g %g", low, high); /* randomc.c PetscRandomSetInterval */
  r->low = low; r->width = high - low; r->iset = true;
  return 0;
}
extern "C" PetscErrorCode PetscRandomGetValue(PetscRandom r, PetscScalar *val)
{
  if (!r || !val) SETERRQ(PETSC_ERR_ARG_NULL, "PetscRandomGetValue: null argument");
  const double u = erand48(r->x);
  *val = r->iset ? r->width * u + r->low : u; /* rand48.c:40-41 */
  return 0;
}
extern "C" PetscErrorCode PetscRandomDestroy(PetscRandom *r) { if (r && *r) { delete *r; *r = NULL; } return 0; }
/* VecSetRandom vector.c:1205-1231 -> VecSetRandom_Seq bvec2.c:  xx[i] = PetscRandomGetValue, i ascending over the local part; the values are
   made on the host and sent to the device once (as VecSetRandom_SeqCUSP's host fallback would) */
extern "C" PetscErrorCode VecSetRandom(Vec x, PetscRandom rctx)
{
  VecCheckReady(x, "VecSetRandom");
  PetscRandom    own = NULL;
  PetscErrorCode ierr;
  if (!rctx) { ierr = PetscRandomCreate(x->comm, &own); CHKERRQ(ierr); rctx = own; }
  PetscScalar *a;
  ierr = VecGetArray(x, &a); CHKERRQ(ierr);
  for (PetscInt i = 0; i < x->n; i++) { ierr = PetscRandomGetValue(rctx, &a[i]); CHKERRQ(ierr); }
  ierr = VecRestoreArray(x, &a); CHKERRQ(ierr);
  return PetscRandomDestroy(&own);
}

/* The device pointer, for callers that launch work of their own (role of VecCUSPGetArrayRead/Write, veccusp.cu:1993-2040).  Elementwise
   operations still waiting in the context are launched first (phip_stream flushes them), so that whatever the caller queues on
   phip_stream(), or does after phip_sync(), sees them.  Write access: the vector's by-products are dropped when the pointer is handed
   out and its state is bumped again when the caller gives it back (VecHipRestoreArrayWrite), as VecRestoreArray does (rvector.c:1865):
   values cached against the old state can no longer be served while the caller writes. */
extern "C" PetscErrorCode VecHipGetArrayRead(Vec v, const PetscScalar **d_a)
{
  VecCheckReady(v, "VecHipGetArrayRead");
  if (!d_a) SETERRQ(PETSC_ERR_ARG_NULL, "VecHipGetArrayRead: null pointer");
  (void)phip_stream(VCTX(v));
  *d_a = v->d;
  return 0;
}
extern "C" PetscErrorCode VecHipGetArrayWrite(Vec v, PetscScalar **d_a)
{
  VecCheckReady(v, "VecHipGetArrayWrite");
  if (!d_a) SETERRQ(PETSC_ERR_ARG_NULL, "VecHipGetArrayWrite: null pointer");
  (void)phip_stream(VCTX(v));
  *d_a = v->d;
  v->state++; v->bp.kind = 0; v->bp.live = false;
  return 0;
}
extern "C" PetscErrorCode VecHipRestoreArrayWrite(Vec v, PetscScalar **d_a)
{
  VecCheckReady(v, "VecHipRestoreArrayWrite");
  if (d_a) *d_a = NULL;
  v->state++; v->bp.kind = 0; v->bp.live = false;
  return 0;
}

extern "C" PetscErrorCode VecSetValues(Vec v, PetscInt ni, const PetscInt ix[], const PetscScalar y[], InsertMode iora)
{
  VecCheckReady(v, "VecSetValues");
  const PetscInt lo = v->range[(size_t)v->comm->rank], hi = v->range[(size_t)v->comm->rank + 1];
  for (PetscInt k = 0; k < ni; k++) {
    if (ix[k] < 0) continue; /* negative indices are ignored (bvec2.c VecSetValues_Seq) */
    if (ix[k] >= v->N) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Out of range index value %d maximum %d", ix[k], v->N);
    if (ix[k] < lo || ix[k] >= hi) SETERRQ(PETSC_ERR_SUP, "VecSetValues: entry %d is owned by another process; only locally owned entries are supported", ix[k]);
    v->stash_i.push_back(ix[k] - lo); v->stash_v.push_back(y[k]); v->stash_mode.push_back((int)iora);
  }
  return 0;
}
extern "C" PetscErrorCode VecAssemblyBegin(Vec v) { VecCheckReady(v, "VecAssemblyBegin"); return 0; }
extern "C" PetscErrorCode VecAssemblyEnd(Vec v)
{
  VecCheckReady(v, "VecAssemblyEnd");
  if (v->stash_i.empty()) return 0;
  PetscScalar   *a;
  PetscErrorCode ierr = VecGetArray(v, &a); CHKERRQ(ierr);
  for (size_t k = 0; k < v->stash_i.size(); k++) {
    if (v->stash_mode[k] == ADD_VALUES) a[v->stash_i[k]] += v->stash_v[k]; else a[v->stash_i[k]] = v->stash_v[k];
  }
  v->stash_i.clear(); v->stash_v.clear(); v->stash_mode.clear();
  return VecRestoreArray(v, &a);
}

/* ---- interface wrappers (argument checks + dispatch + state bump), rvector.c / vector.c ---- */
extern "C" PetscErrorCode VecSet(Vec x, PetscScalar alpha)
{
  VecCheckReady(x, "VecSet");
  PetscErrorCode ierr = (*x->ops.set)(x, alpha); CHKERRQ(ierr);
  x->state++;
  return 0;
}
extern "C" PetscErrorCode VecCopy(Vec x, Vec y)
{
  VecCheckReady(x, "VecCopy"); VecCheckReady(y, "VecCopy");
  if (x == y) return 0; /* vector.c:1643 */
  VecCheckSame(x, y, "VecCopy");
  PetscErrorCode ierr = (*x->ops.copy)(x, y); CHKERRQ(ierr);
  y->state++;
  return 0;
}
extern "C" PetscErrorCode VecSwap(Vec x, Vec y)
{
  VecCheckReady(x, "VecSwap"); VecCheckReady(y, "VecSwap"); VecCheckSame(x, y, "VecSwap");
  PetscErrorCode ierr = (*x->ops.swap)(x, y); CHKERRQ(ierr);
  x->state++; y->state++;
  return 0;
}
extern "C" PetscErrorCode VecScale(Vec x, PetscScalar alpha)
{
  VecCheckReady(x, "VecScale");
  if (alpha != 1.0) { /* rvector.c:514 */
    PetscErrorCode ierr = (*x->ops.scale)(x, alpha); CHKERRQ(ierr);
    x->state++;
  }
  return 0;
}
extern "C" PetscErrorCode VecAXPY(Vec y, PetscScalar alpha, Vec x)
{
  VecCheckReady(x, "VecAXPY"); VecCheckReady(y, "VecAXPY"); VecCheckSame(x, y, "VecAXPY");
  if (x == y) SETERRQ(PETSC_ERR_ARG_IDN, "x and y cannot be the same vector"); /* rvector.c:638 */
  PetscErrorCode ierr = (*y->ops.axpy)(y, alpha, x); CHKERRQ(ierr);
  y->state++;
  return 0;
}
extern "C" PetscErrorCode VecAYPX(Vec y, PetscScalar alpha, Vec x)
{
  VecCheckReady(x, "VecAYPX"); VecCheckReady(y, "VecAYPX"); VecCheckSame(x, y, "VecAYPX");
  if (x == y) SETERRQ(PETSC_ERR_ARG_IDN, "x and y must be different vectors");
  PetscErrorCode ierr = (*y->ops.aypx)(y, alpha, x); CHKERRQ(ierr);
  y->state++;
  return 0;
}
extern "C" PetscErrorCode VecAXPBY(Vec y, PetscScalar alpha, PetscScalar beta, Vec x)
{
  VecCheckReady(x, "VecAXPBY"); VecCheckReady(y, "VecAXPBY"); VecCheckSame(x, y, "VecAXPBY");
  PetscErrorCode ierr = (*y->ops.axpby)(y, alpha, beta, x); CHKERRQ(ierr);
  y->state++;
  return 0;
}
extern "C" PetscErrorCode VecWAXPY(Vec w, PetscScalar alpha, Vec x, Vec y)
{
  VecCheckReady(w, "VecWAXPY"); VecCheckReady(x, "VecWAXPY"); VecCheckReady(y, "VecWAXPY"); VecCheckSame(x, y, "VecWAXPY"); VecCheckSame(w, x, "VecWAXPY");
  if (w == y) SETERRQ(PETSC_ERR_SUP, "Result vector w cannot be same as input vector y, suggest VecAXPY()"); /* rvector.c:758 */
  if (w == x) SETERRQ(PETSC_ERR_SUP, "Result vector w cannot be same as input vector x, suggest VecAYPX()");
  PetscErrorCode ierr = (*w->ops.waxpy)(w, alpha, x, y); CHKERRQ(ierr);
  w->state++;
  return 0;
}
extern "C" PetscErrorCode VecAXPBYPCZ(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y)
{
  VecCheckReady(z, "VecAXPBYPCZ"); VecCheckReady(x, "VecAXPBYPCZ"); VecCheckReady(y, "VecAXPBYPCZ"); VecCheckSame(x, y, "VecAXPBYPCZ"); VecCheckSame(z, x, "VecAXPBYPCZ");
  PetscErrorCode ierr = (*z->ops.axpbypcz)(z, alpha, beta, gamma, x, y); CHKERRQ(ierr);
  z->state++;
  return 0;
}
extern "C" PetscErrorCode VecMAXPY(Vec y, PetscInt nv, const PetscScalar alpha[], Vec x[])
{
  VecCheckReady(y, "VecMAXPY");
  if (!nv) return 0;                                                         /* rvector.c:1271 */
  if (nv < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Number of vectors (given %d) cannot be negative", nv);
  for (PetscInt k = 0; k < nv; k++) { VecCheckReady(x[k], "VecMAXPY"); VecCheckSame(y, x[k], "VecMAXPY"); }
  PetscErrorCode ierr = (*y->ops.maxpy)(y, nv, alpha, x); CHKERRQ(ierr);
  y->state++;
  return 0;
}
extern "C" PetscErrorCode VecPointwiseMult(Vec w, Vec x, Vec y)
{
  VecCheckReady(w, "VecPointwiseMult"); VecCheckReady(x, "VecPointwiseMult"); VecCheckReady(y, "VecPointwiseMult"); VecCheckSame(x, y, "VecPointwiseMult"); VecCheckSame(w, x, "VecPointwiseMult");
  PetscErrorCode ierr = (*w->ops.pointwisemult)(w, x, y); CHKERRQ(ierr);
  w->state++;
  return 0;
}
extern "C" PetscErrorCode VecPointwiseDivide(Vec w, Vec x, Vec y)
{
  VecCheckReady(w, "VecPointwiseDivide"); VecCheckReady(x, "VecPointwiseDivide"); VecCheckReady(y, "VecPointwiseDivide"); VecCheckSame(x, y, "VecPointwiseDivide"); VecCheckSame(w, x, "VecPointwiseDivide");
  PetscErrorCode ierr = (*w->ops.pointwisedivide)(w, x, y); CHKERRQ(ierr);
  w->state++;
  return 0;
}
extern "C" PetscErrorCode VecReciprocal(Vec x)
{
  VecCheckReady(x, "VecReciprocal");
  PetscErrorCode ierr = (*x->ops.reciprocal)(x); CHKERRQ(ierr);
  x->state++;
  return 0;
}
extern "C" PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *val)
{
  VecCheckReady(x, "VecDot"); VecCheckReady(y, "VecDot"); VecCheckSame(x, y, "VecDot");
  if (!val) SETERRQ(PETSC_ERR_ARG_NULL, "VecDot: null result");
  if (bp_try_dot(x, y, val)) return 0;
  return (*x->ops.dot)(x, y, val);
}
extern "C" PetscErrorCode VecTDot(Vec x, Vec y, PetscScalar *val)
{
  VecCheckReady(x, "VecTDot"); VecCheckReady(y, "VecTDot"); VecCheckSame(x, y, "VecTDot");
  if (!val) SETERRQ(PETSC_ERR_ARG_NULL, "VecTDot: null result");
  if (bp_try_dot(x, y, val)) return 0; /* real scalars: TDot == Dot */
  return (*x->ops.tdot)(x, y, val);
}
extern "C" PetscErrorCode VecMDot(Vec x, PetscInt nv, const Vec y[], PetscScalar val[])
{
  VecCheckReady(x, "VecMDot");
  if (!nv) return 0;                                                         /* rvector.c:1226 */
  if (nv < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Number of vectors (given %d) cannot be negative", nv);
  for (PetscInt k = 0; k < nv; k++) { VecCheckReady(y[k], "VecMDot"); VecCheckSame(x, y[k], "VecMDot"); }
  return (*x->ops.mdot)(x, nv, y, val);
}
/* VecMTDot rvector.c:1178-1207: no conjugation -- the same kernel for real scalars (the CUSP types alias it too) */
extern "C" PetscErrorCode VecMTDot(Vec x, PetscInt nv, const Vec y[], PetscScalar val[]) { return VecMDot(x, nv, y, val); }
/* VecDotNorm2 vinv.c:1369-1413 (ops->dotnorm2, veccusp.cu:1973): dp = sum s_i t_i, nm = sum t_i^2 (NOT its square root), one pass and
   one reduction: the fused multi-dot with x = t, y = {s, t} */
extern "C" PetscErrorCode VecDotNorm2(Vec s, Vec t, PetscScalar *dp, PetscReal *nm)
{
  VecCheckReady(s, "VecDotNorm2"); VecCheckReady(t, "VecDotNorm2");
  if (!dp || !nm) SETERRQ(PETSC_ERR_ARG_NULL, "VecDotNorm2: null result");
  if (s->N != t->N) SETERRQ(PETSC_ERR_ARG_INCOMP, "Incompatible vector global lengths");
  if (s->n != t->n) SETERRQ(PETSC_ERR_ARG_INCOMP, "Incompatible vector local lengths");
  const Vec      ys[2] = {s, t};
  PetscScalar    z[2];
  PetscErrorCode ierr = (*t->ops.mdot)(t, 2, ys, z); CHKERRQ(ierr);
  *dp = z[0]; *nm = z[1];
  return 0;
}
/* ---- split reductions: VecDotBegin/End, VecTDotBegin/End, VecNormBegin/End, PetscCommSplitReductionBegin (src/vec/vec/utils/comb.c) ----
   The reference queues the local results and combines all of them in one MPI_Allreduce at the first End (or, non-blocking, at
   PetscCommSplitReductionBegin), so that the callers in between -- KSPPIPECG's PCApply and MatMult -- hide the reduction's latency.
   Here every Begin launches its local kernel at once, without waiting; the kernels that the caller queues afterwards run behind it, and
   by the first End the results usually sit in pinned host memory already: one wait, one host all-reduce of all queued values. */
struct SplitOp { Vec x, y; int kind; NormType nt; phip_ticket t; int nres; double res[2]; };
struct SplitReduction { std::vector<SplitOp> ops; size_t ended = 0; bool applied = false; };
static void split_free(PetscComm c) { delete c->split; c->split = NULL; }
static SplitReduction *split_of(PetscComm c) { if (!c->split) c->split = new SplitReduction(); return c->split; }
static PetscErrorCode split_begin(Vec x, Vec y, int kind, NormType nt)
{
  SplitReduction *sr = split_of(x->comm);
  if (sr->applied) { /* comb.c:  a Begin after the Ends of the previous round starts a new round */
    if (sr->ended != sr->ops.size()) SETERRQ(PETSC_ERR_ORDER, "Called before all VecxxxEnd() called");
    sr->ops.clear(); sr->ended = 0; sr->applied = false;
  }
  SplitOp op;
  op.x = x; op.y = y; op.kind = kind; op.nt = nt; op.nres = (kind == 4) ? 2 : 1;
  const double *p[1] = {y ? y->d : NULL};
  CHKPHIP(phip_vec_reduce_begin(VCTX(x), kind, x->n, x->d, 1, p, &op.t));
  sr->ops.push_back(op);
  return 0;
}
static PetscErrorCode split_apply(PetscComm c) /* PetscSplitReductionApply, comb.c: one reduction for everything queued */
{
  SplitReduction *sr = split_of(c);
  if (sr->applied) return 0;
  std::vector<double> sum, mx;
  for (SplitOp &op : sr->ops) {
    CHKPHIP(phip_ticket_wait(c->ctx, &op.t, op.res));
    for (int k = 0; k < op.nres; k++) (op.kind == 3 ? mx : sum).push_back(op.res[k]);
  }
  if (c->size > 1) {
    for (size_t o = 0; o < sum.size(); o += PHIP_MAX_MVEC) CHKPHIP(phip_comm_allreduce_host(c->comm, sum.data() + o, (int32_t)std::min<size_t>(PHIP_MAX_MVEC, sum.size() - o), 0));
    for (size_t o = 0; o < mx.size(); o += PHIP_MAX_MVEC) CHKPHIP(phip_comm_allreduce_host(c->comm, mx.data() + o, (int32_t)std::min<size_t>(PHIP_MAX_MVEC, mx.size() - o), 1));
  }
  size_t is = 0, im = 0;
  for (SplitOp &op : sr->ops)
    for (int k = 0; k < op.nres; k++) op.res[k] = (op.kind == 3) ? mx[im++] : sum[is++];
  sr->applied = true;
  return 0;
}
static PetscErrorCode split_end(Vec x, Vec y, int kind, SplitOp **out)
{
  SplitReduction *sr = split_of(x->comm);
  PetscErrorCode  ierr = split_apply(x->comm); CHKERRQ(ierr);
  if (sr->ended >= sr->ops.size()) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Called VecxxxEnd() more times then VecxxxBegin()");
  SplitOp *op = &sr->ops[sr->ended];
  if (op->x != x || op->y != y || (op->kind == 0) != (kind == 0)) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Called VecxxxEnd() in a different order or with a different vector than VecxxxBegin()");
  sr->ended++;
  *out = op;
  return 0;
}
extern "C" PetscErrorCode VecDotBegin(Vec x, Vec y, PetscScalar *result)
{
  (void)result;
  VecCheckReady(x, "VecDotBegin"); VecCheckReady(y, "VecDotBegin"); VecCheckSame(x, y, "VecDotBegin");
  return split_begin(x, y, 0, NORM_2);
}
extern "C" PetscErrorCode VecDotEnd(Vec x, Vec y, PetscScalar *result)
{
  VecCheckReady(x, "VecDotEnd");
  if (!result) SETERRQ(PETSC_ERR_ARG_NULL, "VecDotEnd: null result");
  SplitOp       *op;
  PetscErrorCode ierr = split_end(x, y, 0, &op); CHKERRQ(ierr);
  *result = op->res[0];
  return 0;
}
extern "C" PetscErrorCode VecTDotBegin(Vec x, Vec y, PetscScalar *result) { return VecDotBegin(x, y, result); }
extern "C" PetscErrorCode VecTDotEnd(Vec x, Vec y, PetscScalar *result) { return VecDotEnd(x, y, result); }
extern "C" PetscErrorCode VecNormBegin(Vec x, NormType ntype, PetscReal *result)
{
  (void)result;
  VecCheckReady(x, "VecNormBegin");
  int kind;
  switch (ntype) {
  case NORM_2: case NORM_FROBENIUS: kind = 1; break;
  case NORM_1: kind = 2; break;
  case NORM_INFINITY: kind = 3; break;
  case NORM_1_AND_2: kind = 4; break;
  default: SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "VecNormBegin: unknown NormType %d", (int)ntype);
  }
  return split_begin(x, NULL, kind, ntype);
}
extern "C" PetscErrorCode VecNormEnd(Vec x, NormType ntype, PetscReal *result)
{
  VecCheckReady(x, "VecNormEnd");
  if (!result) SETERRQ(PETSC_ERR_ARG_NULL, "VecNormEnd: null result");
  SplitOp       *op;
  PetscErrorCode ierr = split_end(x, NULL, 1, &op); CHKERRQ(ierr);
  if (op->nt != ntype) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Called VecNormEnd() with a different norm type than VecNormBegin()");
  if (op->kind == 1) result[0] = sqrt(op->res[0]);
  else if (op->kind == 4) { result[0] = op->res[0]; result[1] = sqrt(op->res[1]); }
  else result[0] = op->res[0];
  return 0;
}
/* comb.c PetscCommSplitReductionBegin: start the (non-blocking) reduction now.  The local kernels are in flight since their Begin; the
   ranks meet at the first End. */
extern "C" PetscErrorCode PetscCommSplitReductionBegin(PetscComm comm) { if (!comm) SETERRQ(PETSC_ERR_ARG_NULL, "PetscCommSplitReductionBegin: null communicator"); return 0; }

extern "C" PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *val)
{
  VecCheckReady(x, "VecNorm");
  if (!val) SETERRQ(PETSC_ERR_ARG_NULL, "VecNorm: null result");
  if ((type == NORM_2 || type == NORM_FROBENIUS) && bp_try_norm2(x, val)) return 0;
  return (*x->ops.norm)(x, type, val);
}
/* VecNormalize rvector.c:328-347: skip scaling when the norm is 0 or 1; scale by tmp = 1.0/norm */
extern "C" PetscErrorCode VecNormalize(Vec x, PetscReal *val)
{
  PetscReal      norm;
  PetscErrorCode ierr = VecNorm(x, NORM_2, &norm); CHKERRQ(ierr);
  if (norm != 0.0 && norm != 1.0) {
    const PetscScalar tmp = 1.0 / norm;
    ierr = VecScale(x, tmp); CHKERRQ(ierr);
  }
  if (val) *val = norm;
  return 0;
}

/* ================================================================================================ */
/* Host CSR under assembly (Mat_SeqAIJ: i, j, a, imax/ilen, diag, compressedrow; aij.h:10-38,106-124)  */
struct HostAIJ {
  PetscInt m = 0, n = 0;
  std::vector<int64_t>  start;     /* first slot of each row in j/a */
  std::vector<PetscInt> len, cap;  /* ilen / imax */
  std::vector<PetscInt> j;
  std::vector<double>   a;
  bool     relocated = false;      /* some row was moved to the tail of j/a */
  bool     assembled = false;
  uint64_t nonzerostate = 0, dev_nonzerostate = ~0ull;
  bool     values_dirty = false;
  /* assembly products (aij.c:968-1024) */
  std::vector<PetscInt> ai, diag, cprow_i, cprow_rindex;
  PetscInt nz = 0, nonzerorowcnt = 0, rmax = 0;
  bool     cprow_use = false;
  phip_mat dev = NULL;
  bool     t_ready = false;        /* the device matrix carries its transposed copy (phip_mat_transpose_setup) */
  bool     dev_numeric = false;    /* dev lives on a GPU (false on a planning-only context: value operations then stay on the host) */
  bool     host_stale = false;     /* MatScale / MatShift / MatDiagonalScale ran on the device: a[] is behind until sync_host() */
  /* bring a[] up to date before the host reads or changes values (the role of MatCUSPCopyFromGPU-style coherence flags, petscimpl.h:749) */
  PetscErrorCode sync_host()
  {
    if (host_stale && dev) CHKPHIP(phip_mat_get_values(dev, a.data()));
    host_stale = false;
    return 0;
  }

  void prealloc(PetscInt m_, PetscInt n_, PetscInt nzrow, const PetscInt *nnz)
  {
    m = m_; n = n_;
    start.assign((size_t)m + 1, 0); len.assign((size_t)m, 0); cap.assign((size_t)m, 0);
    int64_t tot = 0;
    for (PetscInt r = 0; r < m; r++) { cap[(size_t)r] = nnz ? nnz[r] : nzrow; start[(size_t)r] = tot; tot += cap[(size_t)r]; }
    start[(size_t)m] = tot;
    j.assign((size_t)tot, 0); a.assign((size_t)tot, 0.0);
    relocated = false; assembled = false; nonzerostate++;
  }
  /* sorted insert of one entry: MatSetValues_SeqAIJ aij.c:465-494 */
  void set(PetscInt row, PetscInt col, double v, bool add)
  {
    PetscInt *rp = j.data() + start[(size_t)row];
    PetscInt  nrow = len[(size_t)row];
    PetscInt *pos = std::lower_bound(rp, rp + nrow, col);
    size_t    k = (size_t)(pos - rp);
    if (k < (size_t)nrow && *pos == col) {
      double &t = a[(size_t)start[(size_t)row] + k];
      if (add) t += v; else t = v;
      values_dirty = true;
      return;
    }
    if (nrow >= cap[(size_t)row]) { /* no room: move the row to the tail with more slack (role of MatSeqXAIJReallocateAIJ, aij.h:148-179) */
      const PetscInt newcap = cap[(size_t)row] * 2 + 15;
      const size_t   old = (size_t)start[(size_t)row], tail = j.size();
      j.resize(tail + (size_t)newcap); a.resize(tail + (size_t)newcap);
      std::copy(j.begin() + (long)old, j.begin() + (long)old + nrow, j.begin() + (long)tail);
      std::copy(a.begin() + (long)old, a.begin() + (long)old + nrow, a.begin() + (long)tail);
      start[(size_t)row] = (int64_t)tail; cap[(size_t)row] = newcap; relocated = true;
    }
    const size_t s = (size_t)start[(size_t)row];
    for (size_t q = (size_t)nrow; q > k; q--) { j[s + q] = j[s + q - 1]; a[s + q] = a[s + q - 1]; }
    j[s + k] = col; a[s + k] = v;
    len[(size_t)row] = nrow + 1;
    nonzerostate++; values_dirty = true;
  }
  /* MatAssemblyEnd_SeqAIJ aij.c:968-1024 */
  void assembly_end()
  {
    int64_t tot = 0;
    for (PetscInt r = 0; r < m; r++) tot += len[(size_t)r];
    bool slack = relocated;
    for (PetscInt r = 0; r < m && !slack; r++) slack = cap[(size_t)r] != len[(size_t)r];
    if (slack) {
      if (!relocated) { /* rows are in order: squeeze forward in place (aij.c:981-995) */
        int64_t w = 0;
        for (PetscInt r = 0; r < m; r++) {
          const int64_t s = start[(size_t)r];
          const PetscInt L = len[(size_t)r];
          if (w != s) for (PetscInt q = 0; q < L; q++) { j[(size_t)(w + q)] = j[(size_t)(s + q)]; a[(size_t)(w + q)] = a[(size_t)(s + q)]; }
          start[(size_t)r] = w; w += L;
        }
        j.resize((size_t)tot); a.resize((size_t)tot);
      } else {
        std::vector<PetscInt> nj((size_t)tot); std::vector<double> na((size_t)tot);
        int64_t w = 0;
        for (PetscInt r = 0; r < m; r++) {
          const int64_t s = start[(size_t)r];
          const PetscInt L = len[(size_t)r];
          std::copy(j.begin() + (long)s, j.begin() + (long)s + L, nj.begin() + (long)w);
          std::copy(a.begin() + (long)s, a.begin() + (long)s + L, na.begin() + (long)w);
          start[(size_t)r] = w; w += L;
        }
        j.swap(nj); a.swap(na);
      }
      start[(size_t)m] = tot;
      for (PetscInt r = 0; r < m; r++) cap[(size_t)r] = len[(size_t)r];
      relocated = false;
    }
    finish_products();
  }
  void finish_products()
  {
    ai.assign((size_t)m + 1, 0);
    nonzerorowcnt = 0; rmax = 0;
    for (PetscInt r = 0; r < m; r++) {
      ai[(size_t)r + 1] = ai[(size_t)r] + len[(size_t)r];
      nonzerorowcnt += len[(size_t)r] > 0;
      if (len[(size_t)r] > rmax) rmax = len[(size_t)r];
    }
    nz = ai[(size_t)m];
    /* MatMarkDiagonal_SeqAIJ aij.c:1506-1527 */
    diag.assign((size_t)m, 0);
    for (PetscInt r = 0; r < m; r++) {
      diag[(size_t)r] = ai[(size_t)r + 1];
      for (PetscInt k = ai[(size_t)r]; k < ai[(size_t)r + 1]; k++) if (j[(size_t)k] == r) { diag[(size_t)r] = k; break; }
    }
    /* MatCheckCompressedRow(ratio 0.6) compressedrow.c:45-72 */
    const PetscInt zrows = m - nonzerorowcnt;
    cprow_i.clear(); cprow_rindex.clear();
    if (zrows < 0.6 * m) cprow_use = false;
    else {
      cprow_use = true;
      cprow_i.push_back(0);
      for (PetscInt r = 0; r < m; r++) {
        if (ai[(size_t)r + 1] - ai[(size_t)r] == 0) continue;
        cprow_i.push_back(ai[(size_t)r + 1]);
        cprow_rindex.push_back(r);
      }
    }
    assembled = true;
  }
  /* adopt caller CSR (MatCreateSeqAIJWithArrays) */
  void adopt(PetscInt m_, PetscInt n_, const PetscInt *ii, const PetscInt *jj, const double *aa)
  {
    m = m_; n = n_;
    const int64_t tot = ii[m];
    start.assign((size_t)m + 1, 0); len.assign((size_t)m, 0); cap.assign((size_t)m, 0);
    for (PetscInt r = 0; r < m; r++) { start[(size_t)r] = ii[r]; len[(size_t)r] = cap[(size_t)r] = ii[r + 1] - ii[r]; }
    start[(size_t)m] = tot;
    j.assign(jj, jj + tot); a.assign(aa, aa + tot);
    relocated = false; nonzerostate++; values_dirty = true;
  }
  PetscErrorCode upload(phip_ctx ctx)
  {
    if (host_stale && !values_dirty && dev && dev_nonzerostate == nonzerostate) return 0; /* the device holds the newer values */
    /* both copies changed since they last agreed: every host-side writer syncs first (mat_sync_host), so this is a bug in this file, and
       pulling the device values here would silently drop the host's edits */
    if (host_stale && values_dirty) SETERRQ(PETSC_ERR_PLIB, "host values were edited while the device copy held newer ones (missing mat_sync_host)");
    { PetscErrorCode ierr = sync_host(); CHKERRQ(ierr); }
    if (dev && dev_nonzerostate == nonzerostate) {
      if (values_dirty) CHKPHIP(phip_mat_update_values(dev, a.data()));
    } else {
      if (dev) { CHKPHIP(phip_mat_destroy(dev)); dev = NULL; }
      t_ready = false;
      CHKPHIP(phip_mat_create(ctx, m, n, ai.data(), j.data(), a.data(), cprow_use ? 1 : 0, (PetscInt)cprow_rindex.size(),
                              cprow_use ? cprow_i.data() : NULL, cprow_use ? cprow_rindex.data() : NULL, &dev));
      dev_nonzerostate = nonzerostate;
      int devno = -1;
      CHKPHIP(phip_device_info(ctx, &devno, NULL, NULL, NULL, 0));
      dev_numeric = devno >= 0; /* a planning-only context keeps the values on the host */
    }
    values_dirty = false;
    return 0;
  }
  /* built at the first transposed product; phip_mat_update_values keeps it current while the nonzero pattern stays */
  PetscErrorCode need_transpose()
  {
    if (t_ready) return 0;
    { PetscErrorCode ierr = sync_host(); CHKERRQ(ierr); }
    CHKPHIP(phip_mat_transpose_setup(dev, ai.data(), j.data(), a.data()));
    t_ready = true;
    return 0;
  }
  void destroy() { if (dev) phip_mat_destroy(dev); dev = NULL; t_ready = false; }
};

/* ================================================================================================ */
/* Mat                                                                                              */
struct _MatOps { /* the _MatOps slots (include/petsc/private/matimpl.h:28-206) this slice fills */
  PetscErrorCode (*setvalues)(Mat, PetscInt, const PetscInt[], PetscInt, const PetscInt[], const PetscScalar[], InsertMode);
  PetscErrorCode (*mult)(Mat, Vec, Vec);
  PetscErrorCode (*multadd)(Mat, Vec, Vec, Vec);
  PetscErrorCode (*multtranspose)(Mat, Vec, Vec);
  PetscErrorCode (*multtransposeadd)(Mat, Vec, Vec, Vec);
  PetscErrorCode (*getdiagonal)(Mat, Vec);
  PetscErrorCode (*scale)(Mat, PetscScalar);
  PetscErrorCode (*diagonalscale)(Mat, Vec, Vec);
  PetscErrorCode (*shift)(Mat, PetscScalar);
  PetscErrorCode (*assemblyend)(Mat, MatAssemblyType);
  PetscErrorCode (*zeroentries)(Mat);
  PetscErrorCode (*getvecs)(Mat, Vec *, Vec *);
  PetscErrorCode (*destroy)(Mat);
};

struct StashEntry { PetscInt row, col; double v; };

struct _p_Mat {
  PetscComm   comm;
  std::string type;
  PetscInt    m, n, M, N;
  std::vector<PetscInt> rrange, crange;
  bool        type_set, preallocated, assembled, was_assembled;
  uint64_t    state;
  _MatOps     ops;
  /* Mat_SeqAIJ / Mat_MPIAIJ (mpiaij.h:36-79): A = diagonal block (or the whole Seq matrix), B = off-diagonal block */
  HostAIJ     A, B;
  std::vector<PetscInt> garray;
  std::vector<PetscInt> to_procs, to_starts, to_indices, from_procs, from_starts;
  double     *lvec;
  phip_halo   halo;
  std::vector<StashEntry> stash; int stash_mode;
  bool        mpi;
};

/* the host CSR values of both blocks, current (after device-side MatScale / MatShift / MatDiagonalScale) */
static PetscErrorCode mat_sync_host(Mat A)
{
  PetscErrorCode ierr = A->A.sync_host(); CHKERRQ(ierr);
  return A->B.sync_host();
}

static std::map<std::string, PetscErrorCode (*)(Mat)> &MatList()
{
  static std::map<std::string, PetscErrorCode (*)(Mat)> l;
  return l;
}
extern "C" PetscErrorCode MatRegister(const char *sname, PetscErrorCode (*ctor)(Mat))
{
  if (!sname || !ctor) SETERRQ(PETSC_ERR_ARG_NULL, "MatRegister: null argument");
  MatList()[sname] = ctor;
  return 0;
}

#define MCTX(A) ((A)->comm->ctx)
#define MatCheck(A, name) do { if (!(A)) SETERRQ(PETSC_ERR_ARG_NULL, name ": null matrix"); if (!(A)->type_set) SETERRQ(PETSC_ERR_ARG_TYPENOTSET, name ": matrix type not set"); } while (0)
#define MatCheckAssembled(A, name) do { MatCheck(A, name); if (!(A)->assembled) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Not for unassembled matrix"); } while (0)

extern "C" PetscErrorCode MatCreate(PetscComm comm, Mat *A)
{
  if (!comm || !A) SETERRQ(PETSC_ERR_ARG_NULL, "MatCreate: null argument");
  Mat B = new _p_Mat();
  B->comm = comm; B->m = B->n = B->M = B->N = PETSC_DECIDE;
  B->type_set = B->preallocated = B->assembled = B->was_assembled = false; B->state = 0;
  memset(&B->ops, 0, sizeof(B->ops));
  B->lvec = NULL; B->halo = NULL; B->stash_mode = 0; B->mpi = false;
  *A = B;
  return 0;
}
extern "C" PetscErrorCode MatSetSizes(Mat A, PetscInt m, PetscInt n, PetscInt M, PetscInt N)
{
  if (!A) SETERRQ(PETSC_ERR_ARG_NULL, "MatSetSizes: null matrix");
  if (M > 0 && m > M) SETERRQ(PETSC_ERR_ARG_INCOMP, "Local column size %d cannot be larger than global column size %d", m, M); /* gcreate.c:152 */
  if (N > 0 && n > N) SETERRQ(PETSC_ERR_ARG_INCOMP, "Local row size %d cannot be larger than global row size %d", n, N);
  if (A->type_set && A->preallocated) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Cannot change/reset matrix sizes after preallocation");
  A->m = m; A->n = n; A->M = M; A->N = N;
  return 0;
}
static PetscErrorCode mat_layouts(Mat A)
{
  if (!A->rrange.empty()) return 0;
  PetscErrorCode ierr = layout_setup(A->comm, &A->m, &A->M, A->rrange); CHKERRQ(ierr);
  ierr = layout_setup(A->comm, &A->n, &A->N, A->crange); CHKERRQ(ierr);
  return 0;
}

/* ---- SeqAIJHIP ---- */
static PetscErrorCode MatSetValues_SeqAIJHIP(Mat A, PetscInt m, const PetscInt im[], PetscInt n, const PetscInt in[], const PetscScalar v[], InsertMode mode)
{
  for (PetscInt k = 0; k < m; k++) {
    const PetscInt row = im[k];
    if (row < 0) continue;
    if (row >= A->m) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Row too large: row %d max %d", row, A->m - 1);
    for (PetscInt l = 0; l < n; l++) {
      if (in[l] < 0) continue;
      if (in[l] >= A->n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Column too large: col %d max %d", in[l], A->n - 1);
      A->A.set(row, in[l], v[l + k * n], mode == ADD_VALUES);
    }
  }
  return 0;
}
static PetscErrorCode MatAssemblyEnd_SeqAIJHIP(Mat A, MatAssemblyType type)
{
  if (type == MAT_FLUSH_ASSEMBLY) return 0;
  A->A.assembly_end();
  return A->A.upload(MCTX(A)); /* precedent MatAssemblyEnd_SeqAIJCUSP aijcusp.cu:451-462: CPU assemble, then copy to the GPU */
}
static PetscErrorCode MatMult_SeqAIJHIP(Mat A, Vec x, Vec y)
{
  if (A->m == A->n) { /* by-product sum x_i y_i: KSPSolve_CG asks for VecTDot(P,W) right after MatMult(A,P,W), cg.c:206-207 */
    if (bp_want(y, 2)) {
      phip_ticket t;
      CHKPHIP(phip_mat_mult_dot(A->A.dev, x->d, y->d, x->d, 1, &t));
      bp_record(y, 2, true, &t, x, NULL);
      return 0;
    }
    bp_record(y, 2, false, NULL, x, NULL);
  }
  CHKPHIP(phip_mat_mult(A->A.dev, x->d, y->d));
  return 0;
}
static PetscErrorCode MatMultAdd_SeqAIJHIP(Mat A, Vec x, Vec y, Vec z) { CHKPHIP(phip_mat_multadd(A->A.dev, x->d, y->d, z->d)); return 0; }
/* MatMultTranspose_SeqAIJ aij.c:1258-1266 / MatMultTransposeAdd_SeqAIJ aij.c:1207-1254 */
static PetscErrorCode MatMultTranspose_SeqAIJHIP(Mat A, Vec x, Vec y)
{
  PetscErrorCode ierr = A->A.need_transpose(); CHKERRQ(ierr);
  CHKPHIP(phip_mat_mult_transpose(A->A.dev, x->d, y->d));
  return 0;
}
static PetscErrorCode MatMultTransposeAdd_SeqAIJHIP(Mat A, Vec x, Vec y, Vec z)
{
  PetscErrorCode ierr = A->A.need_transpose(); CHKERRQ(ierr);
  CHKPHIP(phip_mat_mult_transpose_add(A->A.dev, x->d, y->d, z->d));
  return 0;
}
static PetscErrorCode MatGetDiagonal_SeqAIJHIP(Mat A, Vec v) { CHKPHIP(phip_mat_getdiagonal(A->A.dev, v->d)); return 0; }
static PetscErrorCode MatZeroEntries_AIJHIP(Mat A) /* MatZeroEntries_SeqAIJ aij.c:1058-1067, on both copies: no value travels */
{
  HostAIJ *blk[2] = {&A->A, &A->B};
  for (int w = 0; w < (A->mpi ? 2 : 1); w++) {
    std::fill(blk[w]->a.begin(), blk[w]->a.end(), 0.0);
    blk[w]->host_stale = false;
    if (blk[w]->dev && blk[w]->dev_numeric && blk[w]->assembled && blk[w]->dev_nonzerostate == blk[w]->nonzerostate && !blk[w]->values_dirty) CHKPHIP(phip_mat_zero_entries(blk[w]->dev));
    else blk[w]->values_dirty = true;
  }
  return 0;
}
/* ---- value updates that keep the nonzero pattern: the host CSR is the master copy (as under the reference's GPU types, whose
   interface routines mark the device copy stale: matrix.c:4850-4853, 4902-4905, axpy.c:166-169); only the VALUES travel to the device
   afterwards (HostAIJ::upload -> phip_mat_update_values), also into the SELL-dict and transposed copies. ---- */
static void aij_scale(HostAIJ &B, double alpha) /* MatScale_SeqAIJ aij.c:2475-2488: BLASscal over a->a */
{
  for (size_t k = 0; k < (size_t)B.nz; k++) B.a[k] *= alpha;
  B.values_dirty = true;
}
static void aij_diagonal_scale(HostAIJ &B, const double *l, const double *r) /* MatDiagonalScale_SeqAIJ aij.c:2200-2235: rows by l, then entries by r[col] */
{
  if (l) for (PetscInt i = 0; i < B.m; i++) for (PetscInt k = B.ai[(size_t)i]; k < B.ai[(size_t)i + 1]; k++) B.a[(size_t)k] *= l[i];
  if (r) for (size_t k = 0; k < (size_t)B.nz; k++) B.a[k] *= r[B.j[k]];
  B.values_dirty = true;
}
static PetscErrorCode mat_upload_values(Mat A)
{
  PetscErrorCode ierr = A->A.upload(MCTX(A)); CHKERRQ(ierr);
  if (A->mpi) { ierr = A->B.upload(MCTX(A)); CHKERRQ(ierr); }
  return 0;
}
/* is the device copy of the block current, so that a value update can run there? */
static bool blk_on_device(const HostAIJ &B) { return B.dev && B.dev_numeric && B.assembled && B.dev_nonzerostate == B.nonzerostate && !B.values_dirty; }

static PetscErrorCode MatScale_AIJHIP(Mat A, PetscScalar a) /* aij.c:2475 / MatScale_MPIAIJ mpiaij.c:1139-1148: both blocks, on the device */
{
  HostAIJ *blk[2] = {&A->A, &A->B};
  for (int w = 0; w < (A->mpi ? 2 : 1); w++) {
    if (blk_on_device(*blk[w])) { CHKPHIP(phip_mat_scale(blk[w]->dev, a)); blk[w]->host_stale = true; }
    else { PetscErrorCode ierr = blk[w]->sync_host(); CHKERRQ(ierr); aij_scale(*blk[w], a); }
  }
  return mat_upload_values(A);
}
static PetscErrorCode MatDiagonalScale_SeqAIJHIP(Mat A, Vec ll, Vec rr)
{
  if (ll && ll->n != A->m) SETERRQ(PETSC_ERR_ARG_SIZ, "Left scaling vector wrong length");
  if (rr && rr->n != A->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Right scaling vector wrong length");
  if (blk_on_device(A->A)) { /* MatDiagonalScale_SeqAIJ aij.c:2200-2235 on the device: rows by l, then entries by r[col] */
    CHKPHIP(phip_mat_diagonal_scale(A->A.dev, ll ? ll->d : NULL, rr ? rr->d : NULL));
    A->A.host_stale = true;
    return 0;
  }
  const PetscScalar *l = NULL, *r = NULL;
  PetscErrorCode     ierr = A->A.sync_host(); CHKERRQ(ierr);
  if (ll) { ierr = VecGetArrayRead(ll, &l); CHKERRQ(ierr); }
  if (rr && rr != ll) { ierr = VecGetArrayRead(rr, &r); CHKERRQ(ierr); } else if (rr) r = l;
  aij_diagonal_scale(A->A, l, r);
  if (rr && rr != ll) { ierr = VecRestoreArrayRead(rr, &r); CHKERRQ(ierr); }
  if (ll) { ierr = VecRestoreArrayRead(ll, &l); CHKERRQ(ierr); }
  return mat_upload_values(A);
}
static PetscErrorCode MatShift_AIJHIP(Mat A, PetscScalar a); /* below MatSetValues / MatAssemblyEnd */

static PetscErrorCode MatCreateVecs_AIJHIP(Mat A, Vec *right, Vec *left) /* MatCreateVecs_SeqAIJCUSP aijcusp.cu:287-309 */
{
  PetscErrorCode ierr;
  if (right) {
    ierr = VecCreate(A->comm, right); CHKERRQ(ierr);
    ierr = VecSetSizes(*right, A->n, A->N); CHKERRQ(ierr);
    ierr = VecSetType(*right, VECHIP); CHKERRQ(ierr);
  }
  if (left) {
    ierr = VecCreate(A->comm, left); CHKERRQ(ierr);
    ierr = VecSetSizes(*left, A->m, A->M); CHKERRQ(ierr);
    ierr = VecSetType(*left, VECHIP); CHKERRQ(ierr);
  }
  return 0;
}
static PetscErrorCode MatDestroy_AIJHIP(Mat A)
{
  A->A.destroy(); A->B.destroy();
  if (A->halo) phip_halo_destroy(A->halo);
  if (A->lvec) phip_free(MCTX(A), A->lvec);
  A->halo = NULL; A->lvec = NULL;
  return 0;
}
/* MatCreate_SeqAIJCUSP aijcusp.cu:565-598 is the recipe: host AIJ + overridden mult/multadd/assemblyend/destroy/getvecs */
static PetscErrorCode MatCreate_SeqAIJHIP(Mat A)
{
  if (A->comm->size > 1) SETERRQ(PETSC_ERR_ARG_WRONG, "Comm must be of size 1"); /* aij.c:3937 */
  PetscErrorCode ierr = mat_layouts(A); CHKERRQ(ierr);
  A->ops.setvalues = MatSetValues_SeqAIJHIP; A->ops.mult = MatMult_SeqAIJHIP; A->ops.multadd = MatMultAdd_SeqAIJHIP;
  A->ops.multtranspose = MatMultTranspose_SeqAIJHIP; A->ops.multtransposeadd = MatMultTransposeAdd_SeqAIJHIP;
  A->ops.scale = MatScale_AIJHIP; A->ops.diagonalscale = MatDiagonalScale_SeqAIJHIP; A->ops.shift = MatShift_AIJHIP;
  A->ops.getdiagonal = MatGetDiagonal_SeqAIJHIP; A->ops.assemblyend = MatAssemblyEnd_SeqAIJHIP; A->ops.zeroentries = MatZeroEntries_AIJHIP;
  A->ops.getvecs = MatCreateVecs_AIJHIP; A->ops.destroy = MatDestroy_AIJHIP;
  A->type = MATSEQAIJHIP; A->type_set = true; A->mpi = false;
  return 0;
}

/* ---- MPIAIJHIP ---- */
/* MatSetValues_MPIAIJ mpiaij.c:528-632: local rows: column in [cstart,cend) -> A with local column; otherwise -> B with the GLOBAL
   column before the first assembly (:614) and the compact column afterwards (colmap lookup :588-613); other rows -> stash (:619-629) */
static PetscErrorCode MatSetValues_MPIAIJHIP(Mat A, PetscInt m, const PetscInt im[], PetscInt n, const PetscInt in[], const PetscScalar v[], InsertMode mode)
{
  const PetscInt rstart = A->rrange[(size_t)A->comm->rank], rend = A->rrange[(size_t)A->comm->rank + 1];
  const PetscInt cstart = A->crange[(size_t)A->comm->rank], cend = A->crange[(size_t)A->comm->rank + 1];
  for (PetscInt i = 0; i < m; i++) {
    if (im[i] < 0) continue;
    if (im[i] >= A->M) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Row too large: row %d max %d", im[i], A->M - 1);
    if (im[i] >= rstart && im[i] < rend) {
      const PetscInt row = im[i] - rstart;
      for (PetscInt jx = 0; jx < n; jx++) {
        const double value = v[i * n + jx];
        if (in[jx] >= cstart && in[jx] < cend) A->A.set(row, in[jx] - cstart, value, mode == ADD_VALUES);
        else if (in[jx] < 0) continue;
        else if (in[jx] >= A->N) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Column too large: col %d max %d", in[jx], A->N - 1);
        else {
          PetscInt col = in[jx];
          if (A->was_assembled) {
            auto it = std::lower_bound(A->garray.begin(), A->garray.end(), in[jx]);
            if (it == A->garray.end() || *it != in[jx])
              SETERRQ(PETSC_ERR_SUP, "Inserting a new off-process nonzero at global row/column (%d, %d) after the first assembly is not supported (the reference disassembles the matrix, mpiaij.c:598-599)", im[i], in[jx]);
            col = (PetscInt)(it - A->garray.begin());
          }
          A->B.set(row, col, value, mode == ADD_VALUES);
        }
      }
    } else {
      if (A->stash_mode && A->stash_mode != (int)mode) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Cannot mix ADD_VALUES and INSERT_VALUES for off-process entries");
      A->stash_mode = (int)mode;
      for (PetscInt jx = 0; jx < n; jx++) {
        if (in[jx] < 0) continue;
        if (in[jx] >= A->N) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Column too large: col %d max %d", in[jx], A->N - 1);
        A->stash.push_back({im[i], in[jx], v[i * n + jx]});
      }
    }
  }
  return 0;
}

/* drain the stash (MatAssemblyBegin/End_MPIAIJ mpiaij.c:680-737): entries travel to the owner of their row */
static PetscErrorCode mat_stash_exchange(Mat A)
{
  const int size = A->comm->size, rank = A->comm->rank;
  int32_t   mine[2] = {(int32_t)A->stash.size(), A->stash_mode};
  std::vector<int32_t> all((size_t)2 * size);
  CHKPHIP(phip_comm_allgather_host(A->comm->comm, mine, all.data(), sizeof(mine)));
  bool any = false; int mode = 0;
  for (int r = 0; r < size; r++) { if (all[(size_t)2 * r]) any = true; if (all[(size_t)2 * r + 1]) { if (mode && mode != all[(size_t)2 * r + 1]) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Some processors inserted others added"); mode = all[(size_t)2 * r + 1]; } }
  if (!any) return 0;
  std::vector<int32_t> scount((size_t)size, 0), rcount((size_t)size, 0);
  std::vector<int> owner(A->stash.size());
  for (size_t k = 0; k < A->stash.size(); k++) {
    const int o = (int)(std::upper_bound(A->rrange.begin(), A->rrange.end(), A->stash[k].row) - A->rrange.begin()) - 1;
    owner[k] = o; scount[(size_t)o] += 4; /* row, col, value as two int32 words */
  }
  std::vector<int32_t> cm((size_t)size * size);
  CHKPHIP(phip_comm_allgather_host(A->comm->comm, scount.data(), cm.data(), sizeof(int32_t) * (size_t)size));
  for (int p = 0; p < size; p++) rcount[(size_t)p] = cm[(size_t)p * size + rank];
  std::vector<size_t> off((size_t)size + 1, 0);
  for (int p = 0; p < size; p++) off[(size_t)p + 1] = off[(size_t)p] + (size_t)scount[(size_t)p];
  std::vector<int32_t> sbuf(off[(size_t)size] + 1), fill((size_t)size, 0);
  for (size_t k = 0; k < A->stash.size(); k++) {
    int32_t *w = sbuf.data() + off[(size_t)owner[k]] + fill[(size_t)owner[k]];
    w[0] = A->stash[k].row; w[1] = A->stash[k].col; memcpy(w + 2, &A->stash[k].v, sizeof(double));
    fill[(size_t)owner[k]] += 4;
  }
  size_t rtot = 0;
  for (int p = 0; p < size; p++) rtot += (size_t)rcount[(size_t)p];
  std::vector<int32_t> rbuf(rtot + 1);
  CHKPHIP(phip_comm_alltoallv_i32(A->comm->comm, sbuf.data(), scount.data(), rbuf.data(), rcount.data()));
  A->stash.clear(); A->stash_mode = 0;
  /* the entries land in the host CSR through MatSetValues_MPIAIJHIP directly (not through the public MatSetValues, which syncs): after a
     device-side MatScale / MatShift / MatDiagonalScale the host values are behind, and a rank that only RECEIVES entries has not synced
     them itself -- they must catch up first, or the additions go into stale values and the upload would have to choose between the two */
  if (rtot) { PetscErrorCode serr = mat_sync_host(A); CHKERRQ(serr); }
  for (size_t k = 0; k + 3 < rtot + 1 && k < rtot; k += 4) {
    PetscInt row = rbuf[k], col = rbuf[k + 1]; double val; memcpy(&val, &rbuf[k + 2], sizeof(double));
    PetscErrorCode ierr = MatSetValues_MPIAIJHIP(A, 1, &row, 1, &col, &val, (InsertMode)mode); CHKERRQ(ierr);
  }
  return 0;
}

/* MatSetUpMultiply_MPIAIJ mmaij.c:69-129 + VecScatterCreate_PtoS vpscat.c:2297-2495 */
static PetscErrorCode MatSetUpMultiply_MPIAIJHIP(Mat A)
{
  const int size = A->comm->size, rank = A->comm->rank;
  HostAIJ  &B = A->B;
  /* mark the global columns B uses; garray = those columns ascending; rewrite B's columns to the compact ids (mmaij.c:71-96) */
  std::vector<PetscInt> indices((size_t)A->N + 1, 0);
  for (PetscInt r = 0; r < B.m; r++)
    for (PetscInt k = 0; k < B.len[(size_t)r]; k++) indices[(size_t)B.j[(size_t)B.start[(size_t)r] + (size_t)k]] = 1;
  A->garray.clear();
  for (PetscInt c = 0; c < A->N; c++) if (indices[(size_t)c]) A->garray.push_back(c);
  const PetscInt ec = (PetscInt)A->garray.size();
  for (PetscInt c = 0; c < ec; c++) indices[(size_t)A->garray[(size_t)c]] = c;
  for (PetscInt r = 0; r < B.m; r++)
    for (PetscInt k = 0; k < B.len[(size_t)r]; k++) { PetscInt &c = B.j[(size_t)B.start[(size_t)r] + (size_t)k]; c = indices[(size_t)c]; }
  B.n = ec; /* aij->B->cmap->n = ec, mmaij.c:97 */
  /* receive side (vpscat.c:2327-2340, 2423-2447): owner of every ghost, neighbours ascending, contiguous landing in lvec */
  std::vector<int32_t> need((size_t)size, 0);
  {
    int jr = 0;
    for (PetscInt k = 0; k < ec; k++) {
      const PetscInt idx = A->garray[(size_t)k];
      if (idx < A->crange[(size_t)jr]) jr = 0;
      while (idx >= A->crange[(size_t)jr + 1]) jr++;
      if (jr == rank) SETERRQ(PETSC_ERR_LIB, "MatSetUpMultiply: ghost column %d is locally owned", idx);
      need[(size_t)jr]++;
    }
  }
  A->from_procs.clear(); A->from_starts.assign(1, 0);
  for (int p = 0; p < size; p++) if (need[(size_t)p]) { A->from_procs.push_back(p); A->from_starts.push_back(A->from_starts.back() + need[(size_t)p]); }
  /* handshake (PetscGatherNumberOfMessages/MessageLengths + Isend/Irecv of the index lists, vpscat.c:2346-2392) */
  std::vector<int32_t> cm((size_t)size * size), want((size_t)size, 0);
  CHKPHIP(phip_comm_allgather_host(A->comm->comm, need.data(), cm.data(), sizeof(int32_t) * (size_t)size));
  size_t wtot = 0;
  for (int p = 0; p < size; p++) { want[(size_t)p] = cm[(size_t)p * size + rank]; wtot += (size_t)want[(size_t)p]; }
  std::vector<int32_t> rbuf(wtot + 1);
  CHKPHIP(phip_comm_alltoallv_i32(A->comm->comm, A->garray.data(), need.data(), rbuf.data(), want.data()));
  /* send side: neighbours ascending (PetscSortMPIIntWithArray :2348), indices = global - owners[rank] (:2416) */
  A->to_procs.clear(); A->to_starts.assign(1, 0); A->to_indices.clear();
  const PetscInt base = A->crange[(size_t)rank];
  size_t pos = 0;
  for (int p = 0; p < size; p++) {
    if (!want[(size_t)p]) continue;
    A->to_procs.push_back(p);
    for (int32_t k = 0; k < want[(size_t)p]; k++) {
      const PetscInt loc = rbuf[pos + (size_t)k] - base;
      if (loc < 0 || loc >= A->n) SETERRQ(PETSC_ERR_LIB, "MatSetUpMultiply: rank %d asked for entry %d that rank %d does not own", p, rbuf[pos + (size_t)k], rank);
      A->to_indices.push_back(loc);
    }
    pos += (size_t)want[(size_t)p];
    A->to_starts.push_back((PetscInt)A->to_indices.size());
  }
  /* lvec (mmaij.c:104) on the device, and the device exchange plan */
  if (A->lvec) { CHKPHIP(phip_free(MCTX(A), A->lvec)); A->lvec = NULL; }
  CHKPHIP(phip_malloc(MCTX(A), sizeof(double) * (size_t)ec, (void **)&A->lvec));
  if (A->halo) { CHKPHIP(phip_halo_destroy(A->halo)); A->halo = NULL; }
  CHKPHIP(phip_halo_create(A->comm->comm, (PetscInt)A->to_procs.size(), A->to_procs.data(), A->to_starts.data(), A->to_indices.data(),
                           (PetscInt)A->from_procs.size(), A->from_procs.data(), A->from_starts.data(), &A->halo));
  return 0;
}

/* MatAssemblyEnd_MPIAIJ mpiaij.c:703-773 */
static PetscErrorCode MatAssemblyEnd_MPIAIJHIP(Mat A, MatAssemblyType type)
{
  PetscErrorCode ierr = mat_stash_exchange(A); CHKERRQ(ierr);
  if (type == MAT_FLUSH_ASSEMBLY) return 0;
  A->A.assembly_end();
  if (!A->was_assembled) { ierr = MatSetUpMultiply_MPIAIJHIP(A); CHKERRQ(ierr); } /* :753-755 */
  A->B.assembly_end();
  ierr = A->A.upload(MCTX(A)); CHKERRQ(ierr);
  ierr = A->B.upload(MCTX(A)); CHKERRQ(ierr);
  return 0;
}
/* MatMult_MPIAIJ mpiaij.c:984-998: ScatterBegin; A*x; ScatterEnd; y += B*lvec -- the exchange overlaps the diagonal block */
static PetscErrorCode MatMult_MPIAIJHIP(Mat A, Vec x, Vec y)
{
  const bool square = A->m == A->n;
  const bool want = square && bp_want(y, 2);
  phip_ticket t;
  CHKPHIP(phip_halo_begin(A->halo, x->d, A->lvec));
  if (want) CHKPHIP(phip_mat_mult_dot(A->A.dev, x->d, y->d, x->d, 0, &t)); /* local sum x_i (A_d x)_i ... */
  else CHKPHIP(phip_mat_mult(A->A.dev, x->d, y->d));
  CHKPHIP(phip_halo_end(A->halo));
  if (want) CHKPHIP(phip_mat_multadd_dot(A->B.dev, A->lvec, y->d, x->d, BP(A->comm)->host_reduce ? 1 : 0, &t)); /* ... + sum x_i (B lvec)_i */
  else CHKPHIP(phip_mat_multadd(A->B.dev, A->lvec, y->d, y->d));
  if (square) bp_record(y, 2, want, want ? &t : NULL, x, NULL);
  return 0;
}
/* MatMultAdd_MPIAIJ mpiaij.c:1014-1025 */
static PetscErrorCode MatMultAdd_MPIAIJHIP(Mat A, Vec x, Vec y, Vec z)
{
  CHKPHIP(phip_halo_begin(A->halo, x->d, A->lvec));
  CHKPHIP(phip_mat_multadd(A->A.dev, x->d, y->d, z->d));
  CHKPHIP(phip_halo_end(A->halo));
  CHKPHIP(phip_mat_multadd(A->B.dev, A->lvec, z->d, z->d));
  return 0;
}
/* MatMultTranspose_MPIAIJ mpiaij.c:1029-1056 (the !merged branch): lvec = B^T x; reverse scatter begins; y = A^T x; the scatter's end
   adds the ghost contributions into y.  The exchange overlaps the diagonal block's transposed product. */
static PetscErrorCode MatMultTranspose_MPIAIJHIP(Mat A, Vec x, Vec y)
{
  PetscErrorCode ierr = A->A.need_transpose(); CHKERRQ(ierr);
  ierr = A->B.need_transpose(); CHKERRQ(ierr);
  CHKPHIP(phip_mat_mult_transpose(A->B.dev, x->d, A->lvec));
  CHKPHIP(phip_halo_reverse_begin(A->halo, A->lvec));
  CHKPHIP(phip_mat_mult_transpose(A->A.dev, x->d, y->d));
  CHKPHIP(phip_halo_reverse_end(A->halo, y->d));
  return 0;
}
/* MatMultTransposeAdd_MPIAIJ mpiaij.c:1102-1117 */
static PetscErrorCode MatMultTransposeAdd_MPIAIJHIP(Mat A, Vec x, Vec y, Vec z)
{
  PetscErrorCode ierr = A->A.need_transpose(); CHKERRQ(ierr);
  ierr = A->B.need_transpose(); CHKERRQ(ierr);
  CHKPHIP(phip_mat_mult_transpose(A->B.dev, x->d, A->lvec));
  CHKPHIP(phip_halo_reverse_begin(A->halo, A->lvec));
  CHKPHIP(phip_mat_mult_transpose_add(A->A.dev, x->d, y->d, z->d));
  CHKPHIP(phip_halo_reverse_end(A->halo, z->d));
  return 0;
}
/* MatDiagonalScale_MPIAIJ mpiaij.c:2040-2069: the right vector's ghost values come through the same scatter as MatMult's */
static PetscErrorCode MatDiagonalScale_MPIAIJHIP(Mat A, Vec ll, Vec rr)
{
  if (rr && rr->n != A->n) SETERRQ(PETSC_ERR_ARG_SIZ, "right vector non-conforming local size");
  if (ll && ll->n != A->m) SETERRQ(PETSC_ERR_ARG_SIZ, "left vector non-conforming local size");
  PetscErrorCode ierr;
  if (rr) { /* collective: every rank exchanges, whether or not its blocks can take the update on the device */
    CHKPHIP(phip_halo_begin(A->halo, rr->d, A->lvec));
    CHKPHIP(phip_halo_end(A->halo));
  }
  if (blk_on_device(A->A) && blk_on_device(A->B)) { /* mpiaij.c:2040-2069 on the device; the off-diagonal block's columns are the ghost entries of lvec */
    CHKPHIP(phip_mat_diagonal_scale(A->B.dev, ll ? ll->d : NULL, rr ? A->lvec : NULL));
    CHKPHIP(phip_mat_diagonal_scale(A->A.dev, ll ? ll->d : NULL, rr ? rr->d : NULL));
    A->A.host_stale = A->B.host_stale = true;
    return 0;
  }
  ierr = mat_sync_host(A); CHKERRQ(ierr);
  std::vector<double> ghost(A->garray.size() + 1);
  if (rr && !A->garray.empty()) CHKPHIP(phip_memcpy_d2h(MCTX(A), ghost.data(), A->lvec, sizeof(double) * A->garray.size()));
  const PetscScalar *l = NULL, *r = NULL;
  if (ll) { ierr = VecGetArrayRead(ll, &l); CHKERRQ(ierr); }
  if (rr && rr != ll) { ierr = VecGetArrayRead(rr, &r); CHKERRQ(ierr); } else if (rr) r = l;
  if (ll) aij_diagonal_scale(A->B, l, NULL);
  aij_diagonal_scale(A->A, l, r);
  if (rr) aij_diagonal_scale(A->B, NULL, ghost.data());
  if (rr && rr != ll) { ierr = VecRestoreArrayRead(rr, &r); CHKERRQ(ierr); }
  if (ll) { ierr = VecRestoreArrayRead(ll, &l); CHKERRQ(ierr); }
  return mat_upload_values(A);
}
/* MatGetDiagonal_MPIAIJ mpiaij.c:1125-1135 */
static PetscErrorCode MatGetDiagonal_MPIAIJHIP(Mat A, Vec v)
{
  if (A->M != A->N) SETERRQ(PETSC_ERR_SUP, "Supports only square matrix where A->A is diag block");
  if (A->rrange[(size_t)A->comm->rank] != A->crange[(size_t)A->comm->rank] || A->m != A->n) SETERRQ(PETSC_ERR_ARG_SIZ, "row partition must equal col partition");
  CHKPHIP(phip_mat_getdiagonal(A->A.dev, v->d));
  return 0;
}
/* MatCreate_MPIAIJCUSP mpiaijcusp.cu:196-226 is the recipe: MPIAIJ whose A and B blocks are the GPU Seq type */
static PetscErrorCode MatCreate_MPIAIJHIP(Mat A)
{
  PetscErrorCode ierr = mat_layouts(A); CHKERRQ(ierr);
  A->ops.setvalues = MatSetValues_MPIAIJHIP; A->ops.mult = MatMult_MPIAIJHIP; A->ops.multadd = MatMultAdd_MPIAIJHIP;
  A->ops.multtranspose = MatMultTranspose_MPIAIJHIP; A->ops.multtransposeadd = MatMultTransposeAdd_MPIAIJHIP;
  A->ops.scale = MatScale_AIJHIP; A->ops.diagonalscale = MatDiagonalScale_MPIAIJHIP; A->ops.shift = MatShift_AIJHIP;
  A->ops.getdiagonal = MatGetDiagonal_MPIAIJHIP; A->ops.assemblyend = MatAssemblyEnd_MPIAIJHIP; A->ops.zeroentries = MatZeroEntries_AIJHIP;
  A->ops.getvecs = MatCreateVecs_AIJHIP; A->ops.destroy = MatDestroy_AIJHIP;
  A->type = MATMPIAIJHIP; A->type_set = true; A->mpi = true;
  return 0;
}
static PetscErrorCode MatCreate_AIJHIP(Mat A) { return (A->comm->size == 1) ? MatCreate_SeqAIJHIP(A) : MatCreate_MPIAIJHIP(A); } /* matreg.c:47-57 */

static void MatRegisterAll()
{
  static bool done = false;
  if (done) return;
  done = true;
  MatRegister(MATSEQAIJHIP, MatCreate_SeqAIJHIP);
  MatRegister(MATMPIAIJHIP, MatCreate_MPIAIJHIP);
  MatRegister(MATAIJHIP, MatCreate_AIJHIP);
}

extern "C" PetscErrorCode MatSetType(Mat A, const char *type)
{
  if (!A || !type) SETERRQ(PETSC_ERR_ARG_NULL, "MatSetType: null argument");
  MatRegisterAll();
  auto it = MatList().find(type);
  if (it == MatList().end()) SETERRQ(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown Mat type given: %s", type); /* matreg.c:63 */
  if (A->type_set) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "MatSetType: type already set to %s", A->type.c_str());
  if ((A->m == PETSC_DECIDE && A->M == PETSC_DECIDE) || (A->n == PETSC_DECIDE && A->N == PETSC_DECIDE)) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Must call MatSetSizes() first");
  return (*it->second)(A);
}
extern "C" PetscErrorCode MatGetType(Mat A, const char **type)
{
  if (!A || !type) SETERRQ(PETSC_ERR_ARG_NULL, "MatGetType: null argument");
  *type = A->type.c_str();
  return 0;
}
extern "C" PetscErrorCode MatSeqAIJSetPreallocation(Mat A, PetscInt nz, const PetscInt nnz[])
{
  MatCheck(A, "MatSeqAIJSetPreallocation");
  if (A->mpi) return 0; /* PetscTryMethod: silently ignored for other types (aij.c:3538) */
  if (nz == PETSC_DEFAULT || nz == PETSC_DECIDE) nz = 5; /* aij.c:3570 */
  if (nz < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "nz cannot be less than 0: value %d", nz);
  if (nnz) for (PetscInt r = 0; r < A->m; r++) {
    if (nnz[r] < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "nnz cannot be less than 0: local row %d value %d", r, nnz[r]);
    if (nnz[r] > A->n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "nnz cannot be greater than row length: local row %d value %d rowlength %d", r, nnz[r], A->n);
  }
  A->A.prealloc(A->m, A->n, nz, nnz);
  A->preallocated = true; A->assembled = false;
  return 0;
}
/* MatMPIAIJSetPreallocation_MPIAIJ mpiaij.c:2793-2821: A is m x n(local), B is m x N until the first assembly */
extern "C" PetscErrorCode MatMPIAIJSetPreallocation(Mat A, PetscInt d_nz, const PetscInt d_nnz[], PetscInt o_nz, const PetscInt o_nnz[])
{
  MatCheck(A, "MatMPIAIJSetPreallocation");
  if (!A->mpi) return 0;
  if (d_nz == PETSC_DEFAULT || d_nz == PETSC_DECIDE) d_nz = 5;
  if (o_nz == PETSC_DEFAULT || o_nz == PETSC_DECIDE) o_nz = 2;
  if (d_nz < 0 || o_nz < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "d_nz/o_nz cannot be less than 0: values %d %d", d_nz, o_nz);
  A->A.prealloc(A->m, A->n, d_nz, d_nnz);
  A->B.prealloc(A->m, A->N, o_nz, o_nnz);
  A->garray.clear(); A->was_assembled = false;
  A->preallocated = true; A->assembled = false;
  return 0;
}
extern "C" PetscErrorCode MatSetUp(Mat A)
{
  MatCheck(A, "MatSetUp");
  if (A->preallocated) return 0;
  return A->mpi ? MatMPIAIJSetPreallocation(A, PETSC_DEFAULT, NULL, PETSC_DEFAULT, NULL) : MatSeqAIJSetPreallocation(A, PETSC_DEFAULT, NULL);
}
extern "C" PetscErrorCode MatSetValues(Mat A, PetscInt m, const PetscInt idxm[], PetscInt n, const PetscInt idxn[], const PetscScalar v[], InsertMode addv)
{
  if (A && (A->A.host_stale || A->B.host_stale)) { PetscErrorCode serr = mat_sync_host(A); CHKERRQ(serr); }
  MatCheck(A, "MatSetValues");
  if (!m || !n) return 0;
  if (!A->preallocated) { PetscErrorCode ierr = MatSetUp(A); CHKERRQ(ierr); }
  if (addv != INSERT_VALUES && addv != ADD_VALUES) SETERRQ(PETSC_ERR_ARG_WRONG, "MatSetValues: InsertMode must be INSERT_VALUES or ADD_VALUES");
  A->assembled = false; /* matrix.c:1183 */
  return (*A->ops.setvalues)(A, m, idxm, n, idxn, v, addv);
}
extern "C" PetscErrorCode MatAssemblyBegin(Mat A, MatAssemblyType type) { MatCheck(A, "MatAssemblyBegin"); (void)type; return 0; }
extern "C" PetscErrorCode MatAssemblyEnd(Mat A, MatAssemblyType type)
{
  MatCheck(A, "MatAssemblyEnd");
  if (!A->preallocated) { PetscErrorCode ierr = MatSetUp(A); CHKERRQ(ierr); }
  PetscErrorCode ierr = (*A->ops.assemblyend)(A, type); CHKERRQ(ierr);
  if (type == MAT_FINAL_ASSEMBLY) { A->assembled = true; A->was_assembled = true; A->state++; } /* matrix.c:5111 */
  return 0;
}
extern "C" PetscErrorCode MatZeroEntries(Mat A)
{
  MatCheck(A, "MatZeroEntries");
  PetscErrorCode ierr = (*A->ops.zeroentries)(A); CHKERRQ(ierr);
  if (A->assembled) { ierr = A->A.upload(MCTX(A)); CHKERRQ(ierr); if (A->mpi) { ierr = A->B.upload(MCTX(A)); CHKERRQ(ierr); } }
  A->state++;
  return 0;
}
extern "C" PetscErrorCode MatZeroRows(Mat A, PetscInt numRows, const PetscInt rows[], PetscScalar diagv, Vec x, Vec b)
{
  MatCheckAssembled(A, "MatZeroRows");
  if (x || b) SETERRQ(PETSC_ERR_SUP, "MatZeroRows: fixing the right-hand side (x, b non-NULL) is not supported");
  { PetscErrorCode serr = mat_sync_host(A); CHKERRQ(serr); }
  const PetscInt rstart = A->rrange[(size_t)A->comm->rank], rend = A->rrange[(size_t)A->comm->rank + 1];
  for (PetscInt k = 0; k < numRows; k++) {
    if (rows[k] < 0 || rows[k] >= A->M) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Row %d out of range", rows[k]);
    if (rows[k] < rstart || rows[k] >= rend) SETERRQ(PETSC_ERR_SUP, "MatZeroRows: row %d is owned by another process", rows[k]);
    const PetscInt r = rows[k] - rstart;
    HostAIJ *blocks[2] = {&A->A, &A->B};
    for (int w = 0; w < (A->mpi ? 2 : 1); w++)
      for (PetscInt q = 0; q < blocks[w]->len[(size_t)r]; q++) blocks[w]->a[(size_t)blocks[w]->start[(size_t)r] + (size_t)q] = 0.0;
    if (diagv != 0.0) {
      if (A->A.diag[(size_t)r] >= A->A.ai[(size_t)r + 1]) SETERRQ(PETSC_ERR_SUP, "MatZeroRows: row %d has no diagonal entry to set", rows[k]);
      A->A.a[(size_t)A->A.diag[(size_t)r]] = diagv;
    }
  }
  A->A.values_dirty = true; A->B.values_dirty = true;
  PetscErrorCode ierr = A->A.upload(MCTX(A)); CHKERRQ(ierr);
  if (A->mpi) { ierr = A->B.upload(MCTX(A)); CHKERRQ(ierr); }
  A->state++;
  return 0;
}

extern "C" PetscErrorCode MatCreateSeqAIJWithArrays(PetscComm comm, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *A)
{
  if (!i || !A) SETERRQ(PETSC_ERR_ARG_NULL, "MatCreateSeqAIJWithArrays: null argument");
  if (i[0]) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "i (row indices) must start with 0"); /* aij.c:4040 */
  PetscErrorCode ierr = MatCreate(comm, A); CHKERRQ(ierr);
  ierr = MatSetSizes(*A, m, n, m, n); CHKERRQ(ierr);
  ierr = MatSetType(*A, MATSEQAIJHIP); CHKERRQ(ierr);
  for (PetscInt r = 0; r < m; r++) {
    if (i[r + 1] < i[r]) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Negative row length in i (row indices) row = %d length = %d", r, i[r + 1] - i[r]);
    for (PetscInt k = i[r]; k < i[r + 1]; k++) {
      if (j[k] < 0 || j[k] >= n) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Column %d out of range in row %d", j[k], r);
      if (k > i[r] && j[k] <= j[k - 1]) SETERRQ(PETSC_ERR_ARG_WRONG, "Column indices of row %d are not sorted and unique", r);
    }
  }
  (*A)->A.adopt(m, n, i, j, a);
  (*A)->preallocated = true;
  ierr = MatAssemblyBegin(*A, MAT_FINAL_ASSEMBLY); CHKERRQ(ierr);
  return MatAssemblyEnd(*A, MAT_FINAL_ASSEMBLY);
}

/* MatCreateMPIAIJWithArrays mpiaij.c:3647-3667 -> MatMPIAIJSetPreallocationCSR_MPIAIJ :3326-3384: local rows, GLOBAL sorted columns */
extern "C" PetscErrorCode MatCreateMPIAIJWithArrays(PetscComm comm, PetscInt m, PetscInt n, PetscInt M, PetscInt N, const PetscInt i[], const PetscInt j[], const PetscScalar a[], Mat *Aout)
{
  if (!i || !Aout) SETERRQ(PETSC_ERR_ARG_NULL, "MatCreateMPIAIJWithArrays: null argument");
  if (i[0]) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "i (row indices) must start with 0");
  if (m < 0) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "local number of rows (m) cannot be PETSC_DECIDE, or negative");
  PetscErrorCode ierr = MatCreate(comm, Aout); CHKERRQ(ierr);
  Mat A = *Aout;
  ierr = MatSetSizes(A, m, n, M, N); CHKERRQ(ierr);
  ierr = MatSetType(A, MATMPIAIJHIP); CHKERRQ(ierr);
  const PetscInt cstart = A->crange[(size_t)comm->rank], cend = A->crange[(size_t)comm->rank + 1];
  std::vector<PetscInt> dn((size_t)m), on((size_t)m);
  for (PetscInt r = 0; r < m; r++) {
    PetscInt d = 0;
    for (PetscInt k = i[r]; k < i[r + 1]; k++) {
      if (j[k] < 0 || j[k] >= A->N) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "Column %d out of range in local row %d", j[k], r);
      if (k > i[r] && j[k] <= j[k - 1]) SETERRQ(PETSC_ERR_ARG_WRONG, "Column indices of local row %d are not sorted and unique", r);
      d += (j[k] >= cstart && j[k] < cend);
    }
    dn[(size_t)r] = d; on[(size_t)r] = i[r + 1] - i[r] - d;
  }
  ierr = MatMPIAIJSetPreallocation(A, 0, dn.data(), 0, on.data()); CHKERRQ(ierr);
  /* sorted rows: append directly (same result as row-by-row MatSetValues_MPIAIJ, mpiaij.c:3371-3378) */
  for (PetscInt r = 0; r < m; r++) {
    size_t pa = (size_t)A->A.start[(size_t)r], pb = (size_t)A->B.start[(size_t)r];
    for (PetscInt k = i[r]; k < i[r + 1]; k++) {
      if (j[k] >= cstart && j[k] < cend) { A->A.j[pa] = j[k] - cstart; A->A.a[pa++] = a[k]; }
      else { A->B.j[pb] = j[k]; A->B.a[pb++] = a[k]; }
    }
    A->A.len[(size_t)r] = dn[(size_t)r]; A->B.len[(size_t)r] = on[(size_t)r];
  }
  A->A.nonzerostate++; A->B.nonzerostate++;
  ierr = MatAssemblyBegin(A, MAT_FINAL_ASSEMBLY); CHKERRQ(ierr);
  return MatAssemblyEnd(A, MAT_FINAL_ASSEMBLY);
}

extern "C" PetscErrorCode MatGetSize(Mat A, PetscInt *M, PetscInt *N) { MatCheck(A, "MatGetSize"); if (M) *M = A->M; if (N) *N = A->N; return 0; }
extern "C" PetscErrorCode MatGetLocalSize(Mat A, PetscInt *m, PetscInt *n) { MatCheck(A, "MatGetLocalSize"); if (m) *m = A->m; if (n) *n = A->n; return 0; }
extern "C" PetscErrorCode MatGetOwnershipRange(Mat A, PetscInt *rstart, PetscInt *rend)
{
  MatCheck(A, "MatGetOwnershipRange");
  if (rstart) *rstart = A->rrange[(size_t)A->comm->rank];
  if (rend) *rend = A->rrange[(size_t)A->comm->rank + 1];
  return 0;
}
extern "C" PetscErrorCode MatCreateVecs(Mat A, Vec *right, Vec *left) { MatCheck(A, "MatCreateVecs"); return (*A->ops.getvecs)(A, right, left); }

/* MatMult matrix.c:2201-2230: checks, dispatch, and the output state bump the device op must do itself */
extern "C" PetscErrorCode MatMult(Mat A, Vec x, Vec y)
{
  MatCheckAssembled(A, "MatMult"); VecCheckReady(x, "MatMult"); VecCheckReady(y, "MatMult");
  if (x == y) SETERRQ(PETSC_ERR_ARG_IDN, "x and y must be different vectors");
  if (A->N != x->N) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec x: global dim %d %d", A->N, x->N);
  if (A->M != y->N) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec y: global dim %d %d", A->M, y->N);
  if (A->m != y->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec y: local dim %d %d", A->m, y->n);
  if (A->n != x->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec x: local dim %d %d", A->n, x->n);
  PetscErrorCode ierr = (*A->ops.mult)(A, x, y); CHKERRQ(ierr);
  y->state++;
  return 0;
}
extern "C" PetscErrorCode MatMultAdd(Mat A, Vec v1, Vec v2, Vec v3) /* matrix.c:2346-2375 */
{
  MatCheckAssembled(A, "MatMultAdd"); VecCheckReady(v1, "MatMultAdd"); VecCheckReady(v2, "MatMultAdd"); VecCheckReady(v3, "MatMultAdd");
  if (A->N != v1->N) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec v1: global dim %d %d", A->N, v1->N);
  if (A->m != v3->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec v3: local dim %d %d", A->m, v3->n);
  if (A->m != v2->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec v2: local dim %d %d", A->m, v2->n);
  if (v1 == v3) SETERRQ(PETSC_ERR_ARG_IDN, "v1 and v3 must be different vectors");
  PetscErrorCode ierr = (*A->ops.multadd)(A, v1, v2, v3); CHKERRQ(ierr);
  v3->state++;
  return 0;
}
/* MatShift_SeqAIJ aij.c:3064-3075 / MatShift_MPIAIJ mpiaij.c:2589-2601 -> MatShift_Basic gcreate.c:6-24: ADD a to every owned (i,i), assemble */
static PetscErrorCode MatShift_AIJHIP(Mat A, PetscScalar a)
{
  const PetscInt rstart = A->rrange[(size_t)A->comm->rank], rend = A->rrange[(size_t)A->comm->rank + 1];
  PetscErrorCode ierr;
  /* every owned row stores its diagonal entry (in the diagonal block): the pattern stays, the addition runs on the device.  Collective
     decision: a rank that must insert entries re-assembles, and assembly is collective (stash exchange). */
  int mine = blk_on_device(A->A) && (!A->mpi || rstart == A->crange[(size_t)A->comm->rank]) && A->m <= A->n;
  for (PetscInt r = 0; mine && r < A->m; r++) mine = A->A.diag[(size_t)r] < A->A.ai[(size_t)r + 1];
  double all = mine ? 0.0 : 1.0;
  if (A->mpi) CHKPHIP(phip_comm_allreduce_host(A->comm->comm, &all, 1, 1));
  if (all == 0.0) {
    CHKPHIP(phip_mat_shift(A->A.dev, a));
    A->A.host_stale = true;
    A->state++;
    return 0;
  }
  for (PetscInt i = rstart; i < rend; i++) { ierr = MatSetValues(A, 1, &i, 1, &i, &a, ADD_VALUES); CHKERRQ(ierr); }
  ierr = MatAssemblyBegin(A, MAT_FINAL_ASSEMBLY); CHKERRQ(ierr);
  return MatAssemblyEnd(A, MAT_FINAL_ASSEMBLY);
}
extern "C" PetscErrorCode MatScale(Mat A, PetscScalar a) /* matrix.c:4883-4913 */
{
  MatCheckAssembled(A, "MatScale");
  if (a != 1.0) {
    if (!A->ops.scale) SETERRQ(PETSC_ERR_SUP, "Mat type %s", A->type.c_str());
    PetscErrorCode ierr = (*A->ops.scale)(A, a); CHKERRQ(ierr);
    A->state++;
  }
  return 0;
}
extern "C" PetscErrorCode MatDiagonalScale(Mat A, Vec l, Vec r) /* matrix.c:4832-4861 */
{
  MatCheck(A, "MatDiagonalScale");
  if (!A->ops.diagonalscale) SETERRQ(PETSC_ERR_SUP, "Mat type %s", A->type.c_str());
  if (!A->assembled) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Not for unassembled matrix");
  if (l) VecCheckReady(l, "MatDiagonalScale");
  if (r) VecCheckReady(r, "MatDiagonalScale");
  PetscErrorCode ierr = (*A->ops.diagonalscale)(A, l, r); CHKERRQ(ierr);
  A->state++;
  return 0;
}
extern "C" PetscErrorCode MatShift(Mat A, PetscScalar a) /* axpy.c:154-177 */
{
  MatCheckAssembled(A, "MatShift");
  if (!A->ops.shift) SETERRQ(PETSC_ERR_SUP, "Mat type %s", A->type.c_str());
  return (*A->ops.shift)(A, a);
}
extern "C" PetscErrorCode MatMultTranspose(Mat A, Vec x, Vec y) /* matrix.c:2259-2288 */
{
  MatCheckAssembled(A, "MatMultTranspose"); VecCheckReady(x, "MatMultTranspose"); VecCheckReady(y, "MatMultTranspose");
  if (x == y) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "x and y must be different vectors");
  if (A->M != x->N) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec x: global dim %d %d", A->M, x->N);
  if (A->N != y->N) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec y: global dim %d %d", A->N, y->N);
  if (A->m != x->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec x: local dim %d %d", A->m, x->n);
  if (A->n != y->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec y: local dim %d %d", A->n, y->n);
  if (!A->ops.multtranspose) SETERRQ(PETSC_ERR_SUP, "This matrix type does not have a multiply tranpose defined");
  PetscErrorCode ierr = (*A->ops.multtranspose)(A, x, y); CHKERRQ(ierr);
  y->state++;
  return 0;
}
extern "C" PetscErrorCode MatMultTransposeAdd(Mat A, Vec v1, Vec v2, Vec v3) /* matrix.c:2435-2462 */
{
  MatCheckAssembled(A, "MatMultTransposeAdd"); VecCheckReady(v1, "MatMultTransposeAdd"); VecCheckReady(v2, "MatMultTransposeAdd"); VecCheckReady(v3, "MatMultTransposeAdd");
  if (!A->ops.multtransposeadd) SETERRQ(PETSC_ERR_SUP, "Mat type %s", A->type.c_str());
  if (v1 == v3) SETERRQ(PETSC_ERR_ARG_IDN, "v1 and v3 must be different vectors");
  if (A->M != v1->N) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec v1: global dim %d %d", A->M, v1->N);
  if (A->N != v2->N) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec v2: global dim %d %d", A->N, v2->N);
  if (A->N != v3->N) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec v3: global dim %d %d", A->N, v3->N);
  if (A->m != v1->n || A->n != v2->n || A->n != v3->n) SETERRQ(PETSC_ERR_ARG_SIZ, "Mat mat,Vec v1,v2,v3: local dims %d %d, %d %d %d", A->m, A->n, v1->n, v2->n, v3->n);
  PetscErrorCode ierr = (*A->ops.multtransposeadd)(A, v1, v2, v3); CHKERRQ(ierr);
  v3->state++;
  return 0;
}
extern "C" PetscErrorCode MatGetDiagonal(Mat A, Vec v)
{
  MatCheckAssembled(A, "MatGetDiagonal"); VecCheckReady(v, "MatGetDiagonal");
  if (v->n != A->m) SETERRQ(PETSC_ERR_ARG_SIZ, "Nonconforming matrix and vector"); /* aij.c:1178 */
  PetscErrorCode ierr = (*A->ops.getdiagonal)(A, v); CHKERRQ(ierr);
  v->state++;
  return 0;
}
/* MatGetRowMaxAbs_SeqAIJ aij.c:2755-2790 / _MPIAIJ mpiaij.c:2418-2460: set-up only, done on the host CSR */
extern "C" PetscErrorCode MatGetRowMaxAbs(Mat A, Vec v, PetscInt idx[])
{
  MatCheckAssembled(A, "MatGetRowMaxAbs"); VecCheckReady(v, "MatGetRowMaxAbs");
  if (idx) SETERRQ(PETSC_ERR_SUP, "MatGetRowMaxAbs: returning the column indices is not supported");
  { PetscErrorCode serr = mat_sync_host(A); CHKERRQ(serr); }
  if (v->n != A->m) SETERRQ(PETSC_ERR_ARG_SIZ, "Nonconforming matrix and vector");
  PetscScalar   *x;
  PetscErrorCode ierr = VecGetArray(v, &x); CHKERRQ(ierr);
  for (PetscInt r = 0; r < A->m; r++) {
    double mx = 0.0;
    for (PetscInt k = A->A.ai[(size_t)r]; k < A->A.ai[(size_t)r + 1]; k++) mx = std::max(mx, fabs(A->A.a[(size_t)k]));
    if (A->mpi) for (PetscInt k = A->B.ai[(size_t)r]; k < A->B.ai[(size_t)r + 1]; k++) mx = std::max(mx, fabs(A->B.a[(size_t)k]));
    x[r] = mx;
  }
  return VecRestoreArray(v, &x);
}
/* MatGetRowSum matrix.c:4445-4470: v = A * ones */
extern "C" PetscErrorCode MatGetRowSum(Mat A, Vec v)
{
  MatCheckAssembled(A, "MatGetRowSum");
  Vec            ones;
  PetscErrorCode ierr = MatCreateVecs(A, &ones, NULL); CHKERRQ(ierr);
  ierr = VecSet(ones, 1.0); CHKERRQ(ierr);
  ierr = MatMult(A, ones, v); CHKERRQ(ierr);
  return VecDestroy(&ones);
}
extern "C" PetscErrorCode MatDestroy(Mat *A)
{
  if (!A || !*A) return 0;
  if ((*A)->ops.destroy) { PetscErrorCode ierr = (*(*A)->ops.destroy)(*A); CHKERRQ(ierr); }
  delete *A;
  *A = NULL;
  return 0;
}

extern "C" PetscErrorCode MatHipGetBlockInfo(Mat A, int which, PetscInt *m, PetscInt *n, PetscInt *nz, PetscInt *nonzerorowcnt, PetscInt *rmax, PetscBool *cprow_use, PetscInt *cprow_nrows)
{
  MatCheckAssembled(A, "MatHipGetBlockInfo");
  if (which == 1 && !A->mpi) SETERRQ(PETSC_ERR_ARG_WRONG, "MatHipGetBlockInfo: a Seq matrix has no off-diagonal block");
  const HostAIJ &H = which ? A->B : A->A;
  if (m) *m = H.m; if (n) *n = H.n; if (nz) *nz = H.nz; if (nonzerorowcnt) *nonzerorowcnt = H.nonzerorowcnt; if (rmax) *rmax = H.rmax;
  if (cprow_use) *cprow_use = H.cprow_use ? PETSC_TRUE : PETSC_FALSE;
  if (cprow_nrows) *cprow_nrows = (PetscInt)H.cprow_rindex.size();
  return 0;
}
extern "C" PetscErrorCode MatHipGetBlockArrays(Mat A, int which, PetscInt *i, PetscInt *j, PetscScalar *a, PetscInt *diag, PetscInt *cprow_i, PetscInt *cprow_rindex)
{
  if (A && (A->A.host_stale || A->B.host_stale)) { PetscErrorCode serr = mat_sync_host(A); CHKERRQ(serr); }
  MatCheckAssembled(A, "MatHipGetBlockArrays");
  if (which == 1 && !A->mpi) SETERRQ(PETSC_ERR_ARG_WRONG, "MatHipGetBlockArrays: a Seq matrix has no off-diagonal block");
  const HostAIJ &H = which ? A->B : A->A;
  if (i) memcpy(i, H.ai.data(), sizeof(PetscInt) * ((size_t)H.m + 1));
  if (j) memcpy(j, H.j.data(), sizeof(PetscInt) * (size_t)H.nz);
  if (a) memcpy(a, H.a.data(), sizeof(double) * (size_t)H.nz);
  if (diag) memcpy(diag, H.diag.data(), sizeof(PetscInt) * (size_t)H.m);
  if (H.cprow_use && cprow_i) memcpy(cprow_i, H.cprow_i.data(), sizeof(PetscInt) * H.cprow_i.size());
  if (H.cprow_use && cprow_rindex) memcpy(cprow_rindex, H.cprow_rindex.data(), sizeof(PetscInt) * H.cprow_rindex.size());
  return 0;
}
extern "C" PetscErrorCode MatHipGetDeviceMat(Mat A, int which, phip_mat *dev)
{
  MatCheckAssembled(A, "MatHipGetDeviceMat");
  if (!dev || which < 0 || which > 1 || (which == 1 && !A->mpi)) SETERRQ(PETSC_ERR_ARG_OUTOFRANGE, "MatHipGetDeviceMat: block %d", which);
  *dev = which ? A->B.dev : A->A.dev;
  return 0;
}
extern "C" PetscErrorCode MatHipGetGhostInfo(Mat A, PetscInt *ec, PetscInt *to_n, PetscInt *to_len, PetscInt *from_n)
{
  MatCheckAssembled(A, "MatHipGetGhostInfo");
  if (!A->mpi) SETERRQ(PETSC_ERR_ARG_WRONG, "MatHipGetGhostInfo: not an MPIAIJHIP matrix");
  if (ec) *ec = (PetscInt)A->garray.size();
  if (to_n) *to_n = (PetscInt)A->to_procs.size();
  if (to_len) *to_len = (PetscInt)A->to_indices.size();
  if (from_n) *from_n = (PetscInt)A->from_procs.size();
  return 0;
}
extern "C" PetscErrorCode MatHipGetGhostArrays(Mat A, PetscInt *garray, PetscInt *to_procs, PetscInt *to_starts, PetscInt *to_indices, PetscInt *from_procs, PetscInt *from_starts)
{
  MatCheckAssembled(A, "MatHipGetGhostArrays");
  if (!A->mpi) SETERRQ(PETSC_ERR_ARG_WRONG, "MatHipGetGhostArrays: not an MPIAIJHIP matrix");
  if (garray) memcpy(garray, A->garray.data(), sizeof(PetscInt) * A->garray.size());
  if (to_procs) memcpy(to_procs, A->to_procs.data(), sizeof(PetscInt) * A->to_procs.size());
  if (to_starts) memcpy(to_starts, A->to_starts.data(), sizeof(PetscInt) * A->to_starts.size());
  if (to_indices) memcpy(to_indices, A->to_indices.data(), sizeof(PetscInt) * A->to_indices.size());
  if (from_procs) memcpy(from_procs, A->from_procs.data(), sizeof(PetscInt) * A->from_procs.size());
  if (from_starts) memcpy(from_starts, A->from_starts.data(), sizeof(PetscInt) * A->from_starts.size());
  return 0;
}

/* ================================================================================================ */
/* PetscBinary IO (big-endian; src/sys/fileio/sysio.c:217-300 byte-swaps on little-endian hosts)       */
static inline uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
static inline uint64_t bswap64(uint64_t v) { return __builtin_bswap64(v); }
static PetscErrorCode read_i32(FILE *f, PetscInt *out, size_t n)
{
  if (n && fread(out, sizeof(PetscInt), n, f) != n) SETERRQ(PETSC_ERR_FILE_READ, "Read past end of file");
  for (size_t k = 0; k < n; k++) out[k] = (PetscInt)bswap32((uint32_t)out[k]);
  return 0;
}
static PetscErrorCode read_f64(FILE *f, double *out, size_t n)
{
  if (n && fread(out, sizeof(double), n, f) != n) SETERRQ(PETSC_ERR_FILE_READ, "Read past end of file");
  for (size_t k = 0; k < n; k++) { uint64_t u; memcpy(&u, &out[k], 8); u = bswap64(u); memcpy(&out[k], &u, 8); }
  return 0;
}
static PetscErrorCode write_i32(FILE *f, const PetscInt *in, size_t n)
{
  std::vector<uint32_t> b(n);
  for (size_t k = 0; k < n; k++) b[k] = bswap32((uint32_t)in[k]);
  if (n && fwrite(b.data(), 4, n, f) != n) SETERRQ(PETSC_ERR_FILE_WRITE, "Error writing to file");
  return 0;
}
static PetscErrorCode write_f64(FILE *f, const double *in, size_t n)
{
  std::vector<uint64_t> b(n);
  for (size_t k = 0; k < n; k++) { uint64_t u; memcpy(&u, &in[k], 8); b[k] = bswap64(u); }
  if (n && fwrite(b.data(), 8, n, f) != n) SETERRQ(PETSC_ERR_FILE_WRITE, "Error writing to file");
  return 0;
}

/* MatLoad_SeqAIJ aij.c:4109-4177 / MatLoad_MPIAIJ mpiaij.c:2890-3089.  Same resulting matrix (row split m = M/size + ((M%size)>rank),
   mpiaij.c:2929); every rank reads its own slab of the file instead of rank 0 shipping it. */
extern "C" PetscErrorCode MatLoad(Mat A, const char *filename, int64_t *offset)
{
  if (!A || !filename) SETERRQ(PETSC_ERR_ARG_NULL, "MatLoad: null argument");
  FILE *f = fopen(filename, "rb");
  if (!f) SETERRQ(PETSC_ERR_FILE_OPEN, "Cannot open file %s for reading", filename);
  const int64_t off0 = offset ? *offset : 0;
  PetscInt      header[4];
  PetscErrorCode ierr;
  if (fseeko(f, (off_t)off0, SEEK_SET)) { fclose(f); SETERRQ(PETSC_ERR_FILE_READ, "Cannot seek in %s", filename); }
  ierr = read_i32(f, header, 4); if (ierr) { fclose(f); return ierr; }
  if (header[0] != MAT_FILE_CLASSID) { fclose(f); SETERRQ(PETSC_ERR_FILE_UNEXPECTED, "not matrix object"); }           /* aij.c:4134 */
  const PetscInt M = header[1], N = header[2], nz = header[3];
  if (nz < 0) { fclose(f); SETERRQ(PETSC_ERR_FILE_UNEXPECTED, "Matrix stored in special format on disk,cannot load as SeqAIJ"); } /* aij.c:4137 */
  if (!A->type_set) {
    if (A->M == PETSC_DECIDE && A->m == PETSC_DECIDE) { ierr = MatSetSizes(A, PETSC_DECIDE, PETSC_DECIDE, M, N); if (ierr) { fclose(f); return ierr; } }
    ierr = MatSetType(A, MATAIJHIP); if (ierr) { fclose(f); return ierr; }
  }
  if (A->M != M || A->N != N) { fclose(f); SETERRQ(PETSC_ERR_FILE_UNEXPECTED, "Inconsistent # of rows/cols:Matrix in file is %d x %d, matrix has %d x %d", M, N, A->M, A->N); }
  const PetscInt rstart = A->rrange[(size_t)A->comm->rank], rend = A->rrange[(size_t)A->comm->rank + 1], m = rend - rstart;
  std::vector<PetscInt> rowlens((size_t)rend + 1);
  ierr = read_i32(f, rowlens.data(), (size_t)rend); if (ierr) { fclose(f); return ierr; }
  int64_t before = 0;
  for (PetscInt r = 0; r < rstart; r++) before += rowlens[(size_t)r];
  std::vector<PetscInt> li((size_t)m + 1, 0);
  for (PetscInt r = 0; r < m; r++) li[(size_t)r + 1] = li[(size_t)r] + rowlens[(size_t)(rstart + r)];
  const size_t lnz = (size_t)li[(size_t)m];
  std::vector<PetscInt> lj(lnz + 1);
  std::vector<double>   la(lnz + 1);
  const int64_t cols0 = off0 + 16 + 4 * (int64_t)M, vals0 = cols0 + 4 * (int64_t)nz;
  fseeko(f, (off_t)(cols0 + 4 * before), SEEK_SET);
  ierr = read_i32(f, lj.data(), lnz); if (ierr) { fclose(f); return ierr; }
  fseeko(f, (off_t)(vals0 + 8 * before), SEEK_SET);
  ierr = read_f64(f, la.data(), lnz); if (ierr) { fclose(f); return ierr; }
  fclose(f);
  if (offset) *offset = vals0 + 8 * (int64_t)nz;
  /* preallocate exactly (mpiaij.c:3010-3040 counts the diagonal / off-diagonal entries of every row the same way) */
  std::vector<PetscInt> dn((size_t)m), on((size_t)m);
  const PetscInt cstart = A->crange[(size_t)A->comm->rank], cend = A->crange[(size_t)A->comm->rank + 1];
  bool sorted = true;
  for (PetscInt r = 0; r < m; r++) {
    PetscInt d = 0;
    for (PetscInt k = li[(size_t)r]; k < li[(size_t)r + 1]; k++) {
      if (lj[(size_t)k] < 0 || lj[(size_t)k] >= N) SETERRQ(PETSC_ERR_FILE_UNEXPECTED, "Column %d out of range in row %d of the file", lj[(size_t)k], rstart + r);
      d += (lj[(size_t)k] >= cstart && lj[(size_t)k] < cend);
      if (k > li[(size_t)r] && lj[(size_t)k] <= lj[(size_t)k - 1]) sorted = false;
    }
    dn[(size_t)r] = A->mpi ? d : li[(size_t)r + 1] - li[(size_t)r]; on[(size_t)r] = li[(size_t)r + 1] - li[(size_t)r] - d;
  }
  if (A->mpi) { ierr = MatMPIAIJSetPreallocation(A, 0, dn.data(), 0, on.data()); CHKERRQ(ierr); }
  else { ierr = MatSeqAIJSetPreallocation(A, 0, dn.data()); CHKERRQ(ierr); }
  if (sorted) {
    /* rows as MatView writes them (sorted, unique columns): every rank lays its slab straight into the diagonal / off-diagonal blocks --
       the result of the reference's row-by-row MatSetValues (aij.c:4163-4171, mpiaij.c:3051) without its per-entry searches
        */
    for (PetscInt r = 0; r < m; r++) {
      size_t pa = (size_t)A->A.start[(size_t)r], pb = A->mpi ? (size_t)A->B.start[(size_t)r] : 0;
      for (PetscInt k = li[(size_t)r]; k < li[(size_t)r + 1]; k++) {
        const PetscInt c = lj[(size_t)k];
        if (!A->mpi) { A->A.j[pa] = c; A->A.a[pa++] = la[(size_t)k]; }
        else if (c >= cstart && c < cend) { A->A.j[pa] = c - cstart; A->A.a[pa++] = la[(size_t)k]; }
        else { A->B.j[pb] = c; A->B.a[pb++] = la[(size_t)k]; }
      }
      A->A.len[(size_t)r] = dn[(size_t)r];
      if (A->mpi) A->B.len[(size_t)r] = on[(size_t)r];
    }
    A->A.nonzerostate++; A->A.values_dirty = true;
    if (A->mpi) { A->B.nonzerostate++; A->B.values_dirty = true; }
  } else {
    /* unsorted rows (the format allows them): sorted insertion, row by row */
    for (PetscInt r = 0; r < m; r++) {
      const PetscInt grow = rstart + r;
      ierr = MatSetValues(A, 1, &grow, li[(size_t)r + 1] - li[(size_t)r], lj.data() + li[(size_t)r], la.data() + li[(size_t)r], INSERT_VALUES); CHKERRQ(ierr);
    }
  }
  ierr = MatAssemblyBegin(A, MAT_FINAL_ASSEMBLY); CHKERRQ(ierr);
  return MatAssemblyEnd(A, MAT_FINAL_ASSEMBLY);
}

/* VecLoad_Binary src/vec/vec/utils/vecio.c:77-148 */
extern "C" PetscErrorCode VecLoad(Vec v, const char *filename, int64_t *offset)
{
  if (!v || !filename) SETERRQ(PETSC_ERR_ARG_NULL, "VecLoad: null argument");
  FILE *f = fopen(filename, "rb");
  if (!f) SETERRQ(PETSC_ERR_FILE_OPEN, "Cannot open file %s for reading", filename);
  const int64_t off0 = offset ? *offset : 0;
  PetscInt      header[2];
  fseeko(f, (off_t)off0, SEEK_SET);
  PetscErrorCode ierr = read_i32(f, header, 2); if (ierr) { fclose(f); return ierr; }
  if (header[0] != VEC_FILE_CLASSID) { fclose(f); SETERRQ(PETSC_ERR_ARG_WRONG, "Not a vector next in file"); }
  const PetscInt N = header[1];
  if (!v->setup) {
    if (v->N == PETSC_DECIDE && v->n == PETSC_DECIDE) v->N = N;
    ierr = VecSetType(v, VECHIP); if (ierr) { fclose(f); return ierr; }
  }
  if (v->N != N) { fclose(f); SETERRQ(PETSC_ERR_FILE_UNEXPECTED, "Vector in file different length (%d) then input vector (%d)", N, v->N); }
  const PetscInt lo = v->range[(size_t)v->comm->rank];
  PetscScalar   *a;
  ierr = VecGetArray(v, &a); if (ierr) { fclose(f); return ierr; }
  fseeko(f, (off_t)(off0 + 8 + 8 * (int64_t)lo), SEEK_SET);
  ierr = read_f64(f, a, (size_t)v->n); fclose(f); CHKERRQ(ierr);
  if (offset) *offset = off0 + 8 + 8 * (int64_t)N;
  return VecRestoreArray(v, &a);
}

/* MatView_SeqAIJ_Binary aij.c:547-582 / MatView_MPIAIJ_Binary mpiaij.c:1180-1330: ranks append their parts in rank order */
extern "C" PetscErrorCode MatView(Mat A, const char *filename, int append)
{
  if (A && (A->A.host_stale || A->B.host_stale)) { PetscErrorCode serr = mat_sync_host(A); CHKERRQ(serr); }
  MatCheckAssembled(A, "MatView");
  const int size = A->comm->size, rank = A->comm->rank;
  PetscInt  lnz = A->A.nz + (A->mpi ? A->B.nz : 0);
  std::vector<PetscInt> all((size_t)size);
  CHKPHIP(phip_comm_allgather_host(A->comm->comm, &lnz, all.data(), sizeof(PetscInt)));
  int64_t nz = 0;
  for (int r = 0; r < size; r++) nz += all[(size_t)r];
  if (nz > 2147483647) SETERRQ(PETSC_ERR_SUP, "MatView: more than 2^31-1 nonzeros cannot be stored in the 32-bit PetscBinary header");
  /* merged rows with global, ascending columns: B's columns left of the diagonal block, A, then the rest (MatGetRow_MPIAIJ order) */
  const PetscInt cstart = A->crange[(size_t)rank];
  std::vector<PetscInt> rl((size_t)A->m), cj((size_t)lnz + 1);
  std::vector<double>   ca((size_t)lnz + 1);
  size_t w = 0;
  for (PetscInt r = 0; r < A->m; r++) {
    const size_t w0 = w;
    PetscInt kb = A->mpi ? A->B.ai[(size_t)r] : 0, kbe = A->mpi ? A->B.ai[(size_t)r + 1] : 0;
    for (; kb < kbe && A->garray[(size_t)A->B.j[(size_t)kb]] < cstart; kb++) { cj[w] = A->garray[(size_t)A->B.j[(size_t)kb]]; ca[w++] = A->B.a[(size_t)kb]; }
    for (PetscInt k = A->A.ai[(size_t)r]; k < A->A.ai[(size_t)r + 1]; k++) { cj[w] = A->A.j[(size_t)k] + cstart; ca[w++] = A->A.a[(size_t)k]; }
    for (; kb < kbe; kb++) { cj[w] = A->garray[(size_t)A->B.j[(size_t)kb]]; ca[w++] = A->B.a[(size_t)kb]; }
    rl[(size_t)r] = (PetscInt)(w - w0);
  }
  for (int phase = 0; phase < 3; phase++) {
    for (int r = 0; r < size; r++) {
      if (r == rank) {
        FILE *f = fopen(filename, (phase == 0 && r == 0 && !append) ? "wb" : "ab");
        if (!f) SETERRQ(PETSC_ERR_FILE_OPEN, "Cannot open file %s for writing", filename);
        PetscErrorCode ierr = 0;
        if (phase == 0) {
          if (r == 0) { PetscInt header[4] = {MAT_FILE_CLASSID, A->M, A->N, (PetscInt)nz}; ierr = write_i32(f, header, 4); }
          if (!ierr) ierr = write_i32(f, rl.data(), (size_t)A->m);
        } else if (phase == 1) ierr = write_i32(f, cj.data(), (size_t)lnz);
        else ierr = write_f64(f, ca.data(), (size_t)lnz);
        fclose(f);
        CHKERRQ(ierr);
      }
      CHKPHIP(phip_comm_barrier(A->comm->comm));
    }
  }
  return 0;
}
extern "C" PetscErrorCode VecView(Vec v, const char *filename, int append)
{
  VecCheckReady(v, "VecView");
  const PetscScalar *a;
  PetscErrorCode     ierr = VecGetArrayRead(v, &a); CHKERRQ(ierr);
  for (int r = 0; r < v->comm->size; r++) {
    if (r == v->comm->rank) {
      FILE *f = fopen(filename, (r == 0 && !append) ? "wb" : "ab");
      if (!f) SETERRQ(PETSC_ERR_FILE_OPEN, "Cannot open file %s for writing", filename);
      if (r == 0) { PetscInt header[2] = {VEC_FILE_CLASSID, v->N}; ierr = write_i32(f, header, 2); }
      if (!ierr) ierr = write_f64(f, a, (size_t)v->n);
      fclose(f);
      CHKERRQ(ierr);
    }
    CHKPHIP(phip_comm_barrier(v->comm->comm));
  }
  return VecRestoreArrayRead(v, &a);
}

/* ================================================================================================ */
/* PC: PCNONE and PCJACOBI (src/ksp/pc/impls/jacobi/jacobi.c)                                        */
struct _p_PC {
  PetscComm   comm;
  std::string type;
  Mat         mat, pmat;
  Vec         diag, diagsqrt;
  PCJacobiType jtype;
  bool        useabs;
  bool        setupcalled;
  uint64_t    matstate;
};
extern "C" PetscErrorCode PCCreate(PetscComm comm, PC *pc)
{
  if (!comm || !pc) SETERRQ(PETSC_ERR_ARG_NULL, "PCCreate: null argument");
  PC p = new _p_PC();
  p->comm = comm; p->type = ""; p->mat = p->pmat = NULL; p->diag = p->diagsqrt = NULL; p->jtype = PC_JACOBI_DIAGONAL; p->useabs = false;
  p->setupcalled = false; p->matstate = 0;
  *pc = p;
  return 0;
}
static PetscErrorCode PCReset_Jacobi(PC pc) /* jacobi.c:322-331 */
{
  PetscErrorCode ierr = VecDestroy(&pc->diag); CHKERRQ(ierr);
  return VecDestroy(&pc->diagsqrt);
}
extern "C" PetscErrorCode PCSetType(PC pc, const char *type)
{
  if (!pc || !type) SETERRQ(PETSC_ERR_ARG_NULL, "PCSetType: null argument");
  if (strcmp(type, PCNONE) && strcmp(type, PCJACOBI)) SETERRQ(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unable to find requested PC type %s", type);
  if (pc->type != type) { PetscErrorCode ierr = PCReset_Jacobi(pc); CHKERRQ(ierr); pc->setupcalled = false; }
  pc->type = type;
  return 0;
}
extern "C" PetscErrorCode PCSetOperators(PC pc, Mat Amat, Mat Pmat)
{
  if (!pc) SETERRQ(PETSC_ERR_ARG_NULL, "PCSetOperators: null PC");
  if (pc->pmat != Pmat || pc->mat != Amat) { PetscErrorCode ierr = PCReset_Jacobi(pc); CHKERRQ(ierr); }
  pc->mat = Amat; pc->pmat = Pmat; pc->setupcalled = false;
  return 0;
}
extern "C" PetscErrorCode PCJacobiSetType(PC pc, PCJacobiType t) { if (!pc) SETERRQ(PETSC_ERR_ARG_NULL, "PCJacobiSetType: null PC"); pc->jtype = t; pc->setupcalled = false; return 0; }
extern "C" PetscErrorCode PCJacobiSetUseAbs(PC pc, PetscBool flg) { if (!pc) SETERRQ(PETSC_ERR_ARG_NULL, "PCJacobiSetUseAbs: null PC"); pc->useabs = flg != 0; pc->setupcalled = false; return 0; }

/* PCSetUp_Jacobi jacobi.c:140-219: fill whichever of diag / diagsqrt exists */
static PetscErrorCode PCSetUp_Jacobi(PC pc)
{
  PetscErrorCode ierr;
  Vec            targets[2] = {pc->diag, pc->diagsqrt};
  for (int w = 0; w < 2; w++) {
    Vec d = targets[w];
    if (!d) continue;
    if (pc->jtype == PC_JACOBI_ROWMAX) { ierr = MatGetRowMaxAbs(pc->pmat, d, NULL); CHKERRQ(ierr); }
    else if (pc->jtype == PC_JACOBI_ROWSUM) { ierr = MatGetRowSum(pc->pmat, d); CHKERRQ(ierr); }
    else { ierr = MatGetDiagonal(pc->pmat, d); CHKERRQ(ierr); }
    /* VecReciprocal + the host loop :182-195 (or :204-212) as one device kernel */
    CHKPHIP(phip_jacobi_setup(pc->comm->ctx, d->n, d->d, pc->useabs ? 1 : 0, w, NULL));
    d->state++;
  }
  return 0;
}
extern "C" PetscErrorCode PCSetUp(PC pc)
{
  if (!pc) SETERRQ(PETSC_ERR_ARG_NULL, "PCSetUp: null PC");
  if (pc->type.empty()) pc->type = PCJACOBI;
  if (!pc->pmat) SETERRQ(PETSC_ERR_ARG_WRONGSTATE, "Matrix must be set first"); /* precon.c:889 */
  if (pc->setupcalled && pc->matstate == pc->pmat->state) return 0;
  if (pc->type == PCJACOBI) { PetscErrorCode ierr = PCSetUp_Jacobi(pc); CHKERRQ(ierr); }
  pc->setupcalled = true; pc->matstate = pc->pmat->state;
  return 0;
}
extern "C" PetscErrorCode PCApply(PC pc, Vec x, Vec y)
{
  if (!pc) SETERRQ(PETSC_ERR_ARG_NULL, "PCApply: null PC");
  if (x == y) SETERRQ(PETSC_ERR_ARG_IDN, "x and y must be different vectors"); /* precon.c:440 */
  PetscErrorCode ierr = PCSetUp(pc); CHKERRQ(ierr);
  if (pc->type == PCNONE) return VecCopy(x, y);
  if (!pc->diag) { /* PCSetUp_Jacobi_NonSymmetric jacobi.c:253-263: lazily on first apply */
    ierr = MatCreateVecs(pc->pmat, &pc->diag, NULL); CHKERRQ(ierr);
    ierr = PCSetUp_Jacobi(pc); CHKERRQ(ierr);
  }
  return VecPointwiseMult(y, x, pc->diag); /* jacobi.c:288 */
}
static PetscErrorCode PCApplySymmetricLeftOrRight(PC pc, Vec x, Vec y)
{
  PetscErrorCode ierr = PCSetUp(pc); CHKERRQ(ierr);
  if (pc->type == PCNONE) return VecCopy(x, y);
  if (!pc->diagsqrt) { /* jacobi.c:231-241 */
    ierr = MatCreateVecs(pc->pmat, &pc->diagsqrt, NULL); CHKERRQ(ierr);
    ierr = PCSetUp_Jacobi(pc); CHKERRQ(ierr);
  }
  return VecPointwiseMult(y, x, pc->diagsqrt); /* jacobi.c:316 */
}
extern "C" PetscErrorCode PCApplySymmetricLeft(PC pc, Vec x, Vec y) { if (!pc) SETERRQ(PETSC_ERR_ARG_NULL, "null PC"); return PCApplySymmetricLeftOrRight(pc, x, y); }
extern "C" PetscErrorCode PCApplySymmetricRight(PC pc, Vec x, Vec y) { if (!pc) SETERRQ(PETSC_ERR_ARG_NULL, "null PC"); return PCApplySymmetricLeftOrRight(pc, x, y); }
extern "C" PetscErrorCode PCDestroy(PC *pc)
{
  if (!pc || !*pc) return 0;
  PetscErrorCode ierr = PCReset_Jacobi(*pc); CHKERRQ(ierr);
  delete *pc;
  *pc = NULL;
  return 0;
}

/* PCGetType (src/ksp/pc/interface/pcset.c): the type name, or NULL before PCSetType */
extern "C" PetscErrorCode PCGetType(PC pc, const char **type)
{
  if (!pc || !type) SETERRQ(PETSC_ERR_ARG_NULL, "PCGetType: null argument");
  *type = pc->type.empty() ? NULL : pc->type.c_str();
  return 0;
}
/* callers layered on this library (the restated KSP drivers of harness/) report their errors through the same thread-local text that
   PetscHipLastError returns; returns `code` so that `return PetscHipSetError(code, ...)` reads like SETERRQ */
extern "C" PetscErrorCode PetscHipSetError(PetscErrorCode code, const char *msg)
{
  snprintf(g_herr, sizeof(g_herr), "%s", msg ? msg : "");
  return code;
}
