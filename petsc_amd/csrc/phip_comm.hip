// Communicator and ghost exchange of libpetschip (see include/petschip.h).
//
// backend "rccl"    one process per GPU; librccl is dlopen'ed on first use (so single-GPU users never need it).
//                   Reductions are ncclAllReduce on the compute stream; the ghost exchange is one grouped
//                   ncclSend/ncclRecv per neighbour on the communication stream, received straight into lvec
//                   (the receive side of MatSetUpMultiply's scatter is contiguous, vpscat.c:2524-2532).
//                   xGMI is point to point, so neighbour exchanges use one link per peer and never a ring.
//                   Scalars that the host needs (dot products, norms) do NOT go through RCCL: every rank's kernel publishes its local
//                   result to pinned host memory, and the ranks of a node combine them through a shared-memory ring
//                   (phip_comm_allreduce_host below): a few hundred nanoseconds per rank instead of a collective kernel launch.
// backend "shm"     the host side of "rccl" alone: ranks are processes of one node joined by the shared-memory ring (host all-reduce,
//                   all-gather, all-to-all, barrier).  Ghost exchanges are staged through the host and that ring: functional, so that
//                   several processes can share ONE GPU (RCCL refuses that) and the whole multi-process path -- bench.py under
//                   torch.distributed.run included -- can be rehearsed on a one-GPU box; not a performance path.
// backend "self"    size 1.
// backend "threads" the ranks are host threads of one process that share one GPU; collectives rendezvous through a
//                   process-wide table.  It exists so that the distributed code path (plan handshake, pack kernel, lvec
//                   offsets, overlap ordering) can be exercised on a single-GPU box; it is not a performance path.
#include "phip_internal.h"
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <string.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <vector>

/* ---- RCCL entry points (resolved at run time) --------------------------------------------------- */
struct RcclApi {
  void *handle;
  decltype(&ncclGetUniqueId)   GetUniqueId;
  decltype(&ncclCommInitRank)  CommInitRank;
  decltype(&ncclCommDestroy)   CommDestroy;
  decltype(&ncclCommAbort)     CommAbort; /* optional */
  decltype(&ncclAllReduce)     AllReduce;
  decltype(&ncclAllGather)     AllGather;
  decltype(&ncclSend)          Send;
  decltype(&ncclRecv)          Recv;
  decltype(&ncclGroupStart)    GroupStart;
  decltype(&ncclGroupEnd)      GroupEnd;
  decltype(&ncclGetErrorString) GetErrorString;
};
static RcclApi    g_rccl;
static std::mutex g_rccl_mutex;

static int rccl_load(void)
{
  std::lock_guard<std::mutex> lk(g_rccl_mutex);
  if (g_rccl.handle) return 0;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void       *h = NULL;
  for (size_t i = 0; i < sizeof(names) / sizeof(names[0]) && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  PHIP_REQUIRE(h, PHIP_ERR_LIB, "cannot load librccl: %s", dlerror());
#define RESOLVE(field, sym)                                                                   \
  g_rccl.field = (decltype(g_rccl.field))dlsym(h, sym);                                       \
  PHIP_REQUIRE(g_rccl.field, PHIP_ERR_LIB, "librccl lacks symbol %s", sym)
  RESOLVE(GetUniqueId, "ncclGetUniqueId");
  RESOLVE(CommInitRank, "ncclCommInitRank");
  RESOLVE(CommDestroy, "ncclCommDestroy");
  g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(h, "ncclCommAbort");
  RESOLVE(AllReduce, "ncclAllReduce");
  RESOLVE(AllGather, "ncclAllGather");
  RESOLVE(Send, "ncclSend");
  RESOLVE(Recv, "ncclRecv");
  RESOLVE(GroupStart, "ncclGroupStart");
  RESOLVE(GroupEnd, "ncclGroupEnd");
  RESOLVE(GetErrorString, "ncclGetErrorString");
#undef RESOLVE
  g_rccl.handle = h;
  return 0;
}

#define PHIP_CHECK_NCCL(call)                                                                          \
  do {                                                                                                 \
    ncclResult_t _r = (call);                                                                          \
    if (_r != ncclSuccess) {                                                                           \
      phip_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(_r));    \
      return PHIP_ERR_LIB;                                                                             \
    }                                                                                                  \
  } while (0)

/* ---- "threads" backend rendezvous ---------------------------------------------------------------- */
struct ThreadGroup {
  int                     n;
  std::mutex              m;
  std::condition_variable cv;
  int                     waiting = 0;
  long                    generation = 0;
  std::vector<const void *> slot_a, slot_b;
  int                     refs = 0;
  void                   *hr_mem = NULL; /* host reduction ring shared by the threads of the group */
  void barrier()
  {
    std::unique_lock<std::mutex> lk(m);
    const long gen = generation;
    if (++waiting == n) { waiting = 0; generation++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != generation; });
  }
};
static std::map<uint64_t, ThreadGroup *> g_groups;
static std::mutex                        g_groups_mutex;

enum { BACKEND_SELF = 0, BACKEND_RCCL = 1, BACKEND_THREADS = 2, BACKEND_CALLBACKS = 3, BACKEND_SHM = 4 };

/* ---- host reduction ring ---------------------------------------------------------------------------------------------
   HR_ROUNDS x size slots in memory every rank of the node can address (POSIX shared memory between processes, the heap between the
   threads of the "threads" backend).  Round s: rank r fills slot[s % HR_ROUNDS][r] and release-stores s into its sequence word, then
   reads the slots of all ranks as their sequence words reach s, always in rank order -- every rank computes the same bits.
   A slot is reused HR_ROUNDS rounds later; a rank can be at most one round ahead of the slowest reader (it cannot finish round s+1
   before every rank has written round s+1, i.e. finished reading round s), so two rounds of slots would already be enough. */
#define HR_ROUNDS 4
#define HR_PAYLOAD (sizeof(double) * PHIP_MAX_MVEC)
struct alignas(64) HostRingSlot {
  uint64_t seq;
  uint64_t pad[7];
  unsigned char payload[HR_PAYLOAD];
};
struct HostRing {
  HostRingSlot *slots;
  size_t        bytes;
  uint64_t      round;
  int           is_shm;
  char          name[48];
};
static size_t hr_bytes(int size) { return sizeof(HostRingSlot) * HR_ROUNDS * (size_t)size; }

struct phip_comm_s {
  phip_allgather_fn     cb_allgather;
  phip_alltoallv_i32_fn cb_alltoallv;
  void                 *cb_user;
  phip_ctx     ctx;
  int          backend, rank, size;
  ncclComm_t   nccl;
  ThreadGroup *tg;
  uint64_t     tg_key;
  HostRing     hr;       /* hr.slots == NULL: no ring (ranks on several nodes) -> phip_comm_allreduce_host stages through the device */
  double      *d_stage;  /* device scratch of that fallback */
};

struct phip_halo_s {
  phip_comm comm;
  int32_t   to_n, from_n, nsend, nrecv;
  std::vector<int32_t> to_procs, to_starts, from_procs, from_starts;
  int32_t  *d_to_indices;
  double   *d_sendbuf;
  hipEvent_t ev_packed, ev_done;
  const double *cur_lvec;
  double   *lvec;
  /* reverse exchange with ADD_VALUES: the distinct local targets of to_indices, and for each the positions in the receive buffer that add
     into it (ascending = ascending neighbour rank) */
  int32_t   ntgt;
  int32_t  *d_tgt, *d_tptr, *d_tpos;
  const double *rev_lvec;
  std::vector<int32_t> cnt; /* shm backend: cnt[p*size+q] = entries rank p sends to rank q in the forward exchange */
  /* peer-write exchange (PHIP_HALO_IPC=1, ranks of one node): the pack kernel stores straight into the neighbours' receive buffers */
  int                  ipc_on;
  unsigned long long   ipc_seq;      /* number of the exchange under way, 1, 2, ... */
  double              *ipc_recv;     /* [nrecv] what the neighbours store; copied to lvec by the wait kernel */
  unsigned long long  *ipc_sync;     /* [from_n] flags (a sender stores seq once its segment is complete), [to_n] acknowledgements (a receiver
                                        stores seq once it has copied the segment out), [2] arrival counters of the two kernels */
  int32_t             *d_send_seg;   /* [nsend] neighbour (index into to_procs) of every send entry */
  int32_t             *d_to_starts;  /* [to_n + 1] */
  double             **d_peer_recv;  /* [to_n] where neighbour i wants my segment (its ipc_recv + its offset for me) */
  unsigned long long **d_peer_flag;  /* [to_n] my flag word in neighbour i's ipc_sync */
  unsigned long long **d_peer_ack;   /* [from_n] my acknowledgement word in neighbour j's ipc_sync */
  std::vector<void *>  ipc_opened;   /* mappings to close */
  const char          *ipc_why;      /* why the path is not in use although it was asked for (NULL: it is, or it was not asked for) */
};

/* ---- host reduction ring: attach / detach / one round -------------------------------------------------------------------- */
#define HR_MAGIC 0x70686970686f7374ull /* "phiphost" */
#define HR_HEADER 64
static double hr_now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double hr_timeout(void)
{
  static const double t = getenv("PHIP_HOST_RING_TIMEOUT") ? atof(getenv("PHIP_HOST_RING_TIMEOUT")) : 120.0;
  return t;
}
/* wait until *p == want (acquire); fails after hr_timeout() seconds: a peer died or never joined */
static int hr_wait(const uint64_t *p, uint64_t want, const char *what, int peer)
{
  double t0 = 0.0;
  for (unsigned long spin = 1;; spin++) {
    if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == want) return 0;
    __builtin_ia32_pause();
    if ((spin & 0xFFFF) == 0) {
      const double now = hr_now();
      if (t0 == 0.0) t0 = now;
      else if (now - t0 > hr_timeout()) {
        phip_set_error("%s: rank %d did not arrive within %.0f s", what, peer, hr_timeout());
        return PHIP_ERR_LIB;
      }
      if ((spin & 0xFFFFFF) == 0) sched_yield();
    }
  }
}
static uint64_t hr_hash(const void *id128, int nranks)
{
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < 128; i++) { h ^= ((const unsigned char *)id128)[i]; h *= 1099511628211ull; }
  h ^= (uint64_t)nranks; h *= 1099511628211ull;
  return h;
}
/* processes of one node: rank 0 creates the segment, the others open it once it exists; the name dies after the first round.
   Header words: [0] HR_MAGIC once the segment is ready, [1] the creator's pid.  A run that died during set-up leaves its segment
   behind under the same name: rank 0 unlinks and recreates it, and the other ranks take a segment only while its creator is alive,
   so one that was opened before rank 0 replaced it is dropped and the name looked up again. */
static int hr_attach_shm(phip_comm c, const void *id128)
{
  HostRing *r = &c->hr;
  r->bytes = HR_HEADER + hr_bytes(c->size);
  snprintf(r->name, sizeof(r->name), "/phip_%016llx", (unsigned long long)hr_hash(id128, c->size));
  void *mem = MAP_FAILED;
  if (c->rank == 0) {
    shm_unlink(r->name); /* a leftover of a run that died under the same key */
    int fd = shm_open(r->name, O_CREAT | O_EXCL | O_RDWR, 0600);
    PHIP_REQUIRE(fd >= 0, PHIP_ERR_LIB, "phip_comm_create: shm_open(%s) failed: %s", r->name, strerror(errno));
    if (ftruncate(fd, (off_t)r->bytes)) { close(fd); shm_unlink(r->name); PHIP_REQUIRE(0, PHIP_ERR_LIB, "phip_comm_create: ftruncate(%s): %s", r->name, strerror(errno)); }
    mem = mmap(NULL, r->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) { shm_unlink(r->name); PHIP_REQUIRE(0, PHIP_ERR_LIB, "phip_comm_create: mmap(%s): %s", r->name, strerror(errno)); }
    ((uint64_t *)mem)[1] = (uint64_t)getpid();
    __atomic_store_n((uint64_t *)mem, HR_MAGIC, __ATOMIC_RELEASE); /* a fresh segment is zero-filled */
  } else {
    const double t0 = hr_now();
    for (;;) {
      int fd = shm_open(r->name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (!fstat(fd, &st) && (size_t)st.st_size == r->bytes) {
          mem = mmap(NULL, r->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
          if (mem != MAP_FAILED) {
            if (__atomic_load_n((const uint64_t *)mem, __ATOMIC_ACQUIRE) == HR_MAGIC) {
              const pid_t creator = (pid_t)((const uint64_t *)mem)[1];
              if (creator > 0 && (kill(creator, 0) == 0 || errno == EPERM)) { close(fd); break; }
            }
            munmap(mem, r->bytes); mem = MAP_FAILED; /* not ready yet, or the leftover of a dead run */
          }
        }
        close(fd);
      }
      PHIP_REQUIRE(hr_now() - t0 < hr_timeout(), PHIP_ERR_LIB, "phip_comm_create: rank 0 never created %s (ranks on different nodes?)", r->name);
      usleep(200);
    }
  }
  r->slots = (HostRingSlot *)((char *)mem + HR_HEADER);
  r->is_shm = 1; r->round = 0;
  return 0;
}
static void hr_detach(phip_comm c)
{
  if (c->hr.slots && c->hr.is_shm) munmap((char *)c->hr.slots - HR_HEADER, c->hr.bytes);
  c->hr.slots = NULL;
}
/* one round: every rank contributes `bytes` (<= HR_PAYLOAD); on return peers[p] points at rank p's contribution, valid until this
   rank's next round but one */
static int hr_round(phip_comm c, const void *mine, size_t bytes, const unsigned char **peers)
{
  HostRing      *r = &c->hr;
  const uint64_t s = ++r->round;
  HostRingSlot  *row = r->slots + (size_t)(s % HR_ROUNDS) * (size_t)c->size;
  memcpy(row[c->rank].payload, mine, bytes);
  __atomic_store_n(&row[c->rank].seq, s, __ATOMIC_RELEASE);
  for (int p = 0; p < c->size; p++) {
    if (p != c->rank) { int rc = hr_wait(&row[p].seq, s, "host ring", p); if (rc) return rc; }
    peers[p] = row[p].payload;
  }
  return 0;
}

extern "C" int phip_comm_unique_id(void *id128)
{
  PHIP_REQUIRE(id128, PHIP_ERR_ARG_NULL, "phip_comm_unique_id: null buffer");
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  PHIP_CHECK_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, 128);
  return 0;
}

/* everything a communicator holds; failed: the peers may be gone, so the RCCL communicator is aborted rather than destroyed */
static void comm_release(phip_comm c, bool failed)
{
  if (c->nccl) {
    hipSetDevice(c->ctx->device);
    if (!failed) {
      hipStreamSynchronize(c->ctx->stream);
      hipStreamSynchronize(c->ctx->comm_stream);
      g_rccl.CommDestroy(c->nccl);
    } else if (g_rccl.CommAbort) g_rccl.CommAbort(c->nccl);
  }
  if (c->d_stage) hipFree(c->d_stage);
  hr_detach(c);
  if (c->tg) {
    std::lock_guard<std::mutex> lk(g_groups_mutex);
    if (--c->tg->refs == 0) { g_groups.erase(c->tg_key); free(c->tg->hr_mem); delete c->tg; }
  }
  delete c;
}

static int comm_create_fill(phip_comm c, phip_ctx ctx, const char *backend, const void *id128, int nranks, int rank)
{
  if (!strcmp(backend, "self")) {
    PHIP_REQUIRE(nranks == 1, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_create: backend self needs nranks 1");
    c->backend = BACKEND_SELF;
  } else if (!strcmp(backend, "rccl")) {
    PHIP_REQUIRE(id128, PHIP_ERR_ARG_NULL, "phip_comm_create: rccl backend needs the unique id");
    PHIP_NEED_DEVICE(ctx, "phip_comm_create(rccl)");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    PHIP_CHECK_HIP(hipSetDevice(ctx->device));
    PHIP_CHECK_NCCL(g_rccl.CommInitRank(&c->nccl, nranks, id, rank));
    c->backend = BACKEND_RCCL;
    PHIP_CHECK_HIP(hipMalloc((void **)&c->d_stage, sizeof(double) * PHIP_MAX_MVEC));
    if (nranks > 1) {
      /* The host ring needs every rank on one node (compare host names; PHIP_HOST_RING=0 forces the device route) and a shared
         segment every rank can map.  The ranks agree through two all-gathers; if anything is amiss all of them stay on the device route. */
      PHIP_REQUIRE(nranks <= 1024, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_create: %d ranks", nranks);
      char     host[256] = {0};
      gethostname(host, sizeof(host) - 1);
      uint64_t rec[2] = {1469598103934665603ull, 1};
      for (size_t i = 0; i < sizeof(host); i++) { rec[0] ^= (unsigned char)host[i]; rec[0] *= 1099511628211ull; }
      if (getenv("PHIP_HOST_RING") && !atoi(getenv("PHIP_HOST_RING"))) rec[0] ^= (uint64_t)rank + 1; /* pretend the ranks are apart */
      if (rank == 0) rec[1] = hr_attach_shm(c, id128) == 0; /* rank 0 creates the segment before anyone looks for it */
      std::vector<uint64_t> all((size_t)2 * nranks);
      if ((rc = phip_comm_allgather_host(c, rec, all.data(), sizeof(rec)))) return rc;
      bool go = all[1] != 0;
      for (int p = 0; p < nranks; p++) go = go && all[(size_t)2 * p] == all[0];
      uint64_t ok = 1;
      if (go && rank != 0) ok = hr_attach_shm(c, id128) == 0;
      std::vector<uint64_t> oks((size_t)nranks);
      if ((rc = phip_comm_allgather_host(c, &ok, oks.data(), sizeof(ok)))) return rc;
      for (int p = 0; p < nranks; p++) go = go && oks[(size_t)p] != 0;
      if (go) {
        const unsigned char *peers[1024];
        rc = hr_round(c, &ok, sizeof(ok), peers); /* everyone has mapped the segment: its name can go */
      } else hr_detach(c);
      if (rank == 0 && all[1]) shm_unlink(c->hr.name);
      if (rc) return rc;
    }
  } else if (!strcmp(backend, "shm")) {
    PHIP_REQUIRE(id128, PHIP_ERR_ARG_NULL, "phip_comm_create: shm backend needs a 128-byte key shared by the ranks");
    PHIP_REQUIRE(nranks <= 1024, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_create: %d ranks on one node", nranks);
    c->backend = BACKEND_SHM;
    if (nranks > 1) {
      int rc = hr_attach_shm(c, id128);
      if (rc) return rc;
      const unsigned char *peers[1024];
      uint64_t             one = 1;
      if ((rc = hr_round(c, &one, sizeof(one), peers))) { if (rank == 0) shm_unlink(c->hr.name); return rc; }
      if (rank == 0) shm_unlink(c->hr.name);
    }
  } else if (!strcmp(backend, "threads")) {
    PHIP_REQUIRE(id128, PHIP_ERR_ARG_NULL, "phip_comm_create: threads backend needs a group key");
    memcpy(&c->tg_key, id128, sizeof(uint64_t));
    {
      std::lock_guard<std::mutex> lk(g_groups_mutex);
      ThreadGroup *&g = g_groups[c->tg_key];
      if (!g) {
        g = new ThreadGroup(); g->n = nranks; g->slot_a.resize((size_t)nranks); g->slot_b.resize((size_t)nranks);
        g->hr_mem = aligned_alloc(64, hr_bytes(nranks));
        memset(g->hr_mem, 0, hr_bytes(nranks));
      }
      g->refs++;
      c->tg = g; /* from here comm_release lets go of the group */
      PHIP_REQUIRE(g->n == nranks, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_create: thread group size mismatch");
      c->hr.slots = (HostRingSlot *)g->hr_mem; c->hr.bytes = hr_bytes(nranks); c->hr.round = 0; c->hr.is_shm = 0;
    }
    c->backend = BACKEND_THREADS;
  } else PHIP_REQUIRE(0, PHIP_ERR_SUP, "phip_comm_create: unknown backend '%s'", backend);
  return 0;
}

extern "C" int phip_comm_create(phip_ctx ctx, const char *backend, const void *id128, int nranks, int rank, phip_comm *out)
{
  PHIP_REQUIRE(ctx && backend && out, PHIP_ERR_ARG_NULL, "phip_comm_create: null argument");
  PHIP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_create: rank %d of %d", rank, nranks);
  phip_comm c = new phip_comm_s();
  c->ctx = ctx; c->rank = rank; c->size = nranks; c->nccl = NULL; c->tg = NULL; c->tg_key = 0; c->backend = BACKEND_SELF;
  memset(&c->hr, 0, sizeof(c->hr)); c->d_stage = NULL;
  const int rc = comm_create_fill(c, ctx, backend, id128, nranks, rank);
  if (rc) { comm_release(c, true); return rc; }
  *out = c;
  return 0;
}

extern "C" int phip_comm_create_callbacks(phip_ctx ctx, int nranks, int rank, phip_allgather_fn allgather, phip_alltoallv_i32_fn alltoallv,
                                          void *user, phip_comm *out)
{
  PHIP_REQUIRE(ctx && out && allgather && alltoallv, PHIP_ERR_ARG_NULL, "phip_comm_create_callbacks: null argument");
  PHIP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_create_callbacks: rank %d of %d", rank, nranks);
  phip_comm c = new phip_comm_s();
  c->ctx = ctx; c->rank = rank; c->size = nranks; c->nccl = NULL; c->tg = NULL; c->tg_key = 0;
  memset(&c->hr, 0, sizeof(c->hr)); c->d_stage = NULL;
  c->backend = BACKEND_CALLBACKS; c->cb_allgather = allgather; c->cb_alltoallv = alltoallv; c->cb_user = user;
  *out = c;
  return 0;
}

extern "C" int phip_comm_destroy(phip_comm c)
{
  if (!c) return 0;
  comm_release(c, false);
  return 0;
}

extern "C" int phip_comm_rank_size(phip_comm c, int *rank, int *size)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_comm_rank_size: null communicator");
  if (rank) *rank = c->rank;
  if (size) *size = c->size;
  return 0;
}

extern "C" int phip_comm_barrier(phip_comm c)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_comm_barrier: null communicator");
  if (c->backend == BACKEND_THREADS) c->tg->barrier();
  else if (c->backend == BACKEND_RCCL || c->backend == BACKEND_CALLBACKS || c->backend == BACKEND_SHM) {
    int32_t one = 1;
    std::vector<int32_t> all((size_t)c->size);
    return phip_comm_allgather_host(c, &one, all.data(), sizeof(int32_t));
  }
  return 0;
}

extern "C" int phip_comm_allreduce(phip_comm c, double *d_buf, int32_t n, int op)
{
  PHIP_REQUIRE(c && (d_buf || !n), PHIP_ERR_ARG_NULL, "phip_comm_allreduce: null argument");
  PHIP_REQUIRE(n >= 0 && n <= PHIP_MAX_MVEC, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_allreduce: n=%d", n);
  PHIP_REQUIRE(op == 0 || op == 1, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_allreduce: op %d", op);
  if (c->size == 1 || n == 0) return 0;
  PHIP_NEED_DEVICE(c->ctx, "phip_comm_allreduce");
  PHIP_FLUSH(c->ctx);
  PHIP_REQUIRE(c->backend != BACKEND_CALLBACKS && c->backend != BACKEND_SHM, PHIP_ERR_SUP, "phip_comm_allreduce: this backend carries no device traffic");
  if (c->backend == BACKEND_RCCL) {
    PHIP_CHECK_HIP(hipSetDevice(c->ctx->device)); /* the caller's thread may have another device current (torch, a second context) */
    PHIP_CHECK_NCCL(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->nccl, c->ctx->stream));
    return 0;
  }
  /* threads: gather every rank's values on the host, combine in rank order, write back */
  ThreadGroup *g = c->tg;
  PHIP_CHECK_HIP(hipStreamSynchronize(c->ctx->stream));
  g->slot_a[(size_t)c->rank] = d_buf;
  g->barrier();
  std::vector<double> acc((size_t)n), tmp((size_t)n);
  for (int p = 0; p < c->size; p++) {
    PHIP_CHECK_HIP(hipMemcpy(tmp.data(), g->slot_a[(size_t)p], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
      if (p == 0) acc[(size_t)i] = tmp[(size_t)i];
      else if (op == 0) acc[(size_t)i] += tmp[(size_t)i];
      else acc[(size_t)i] = (tmp[(size_t)i] != tmp[(size_t)i] || tmp[(size_t)i] > acc[(size_t)i]) ? tmp[(size_t)i] : acc[(size_t)i];
    }
  }
  g->barrier(); /* everyone has read before anyone overwrites */
  PHIP_CHECK_HIP(hipMemcpy(d_buf, acc.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  return 0;
}

/* All-reduce of n host doubles (op 0 SUM, 1 MAX with NaN propagation): the MPI_Allreduce of VecDot_MPI / VecNorm_MPI / VecMDot_MPI
   (pbvec.c:9-19, pvec2.c:10-85).  The local kernels publish their results to pinned host memory anyway (the unmodified KSP consumes
   every reduction on the host), so the cross-rank step stays on the host: one round of the shared-memory ring, combined in rank
   order on every rank (bit-identical results everywhere).  Ranks on several nodes: staged through the device and ncclAllReduce. */
extern "C" int phip_comm_allreduce_host(phip_comm c, double *h_vals, int32_t n, int op)
{
  PHIP_REQUIRE(c && (h_vals || !n), PHIP_ERR_ARG_NULL, "phip_comm_allreduce_host: null argument");
  PHIP_REQUIRE(n >= 0 && n <= PHIP_MAX_MVEC, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_allreduce_host: n=%d", n);
  PHIP_REQUIRE(op == 0 || op == 1, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_allreduce_host: op %d", op);
  if (c->size == 1 || n == 0) return 0;
  std::vector<double>  gathered;
  const unsigned char *peers[1024];
  if (c->hr.slots) {
    int rc = hr_round(c, h_vals, sizeof(double) * (size_t)n, peers);
    if (rc) return rc;
  } else if (c->backend == BACKEND_CALLBACKS) {
    gathered.resize((size_t)n * (size_t)c->size);
    int rc = c->cb_allgather(c->cb_user, h_vals, gathered.data(), sizeof(double) * (size_t)n);
    PHIP_REQUIRE(rc == 0, PHIP_ERR_LIB, "phip_comm_allreduce_host: the application's all-gather callback failed with %d", rc);
    PHIP_REQUIRE(c->size <= 1024, PHIP_ERR_ARG_OUTOFRANGE, "phip_comm_allreduce_host: %d ranks", c->size);
    for (int p = 0; p < c->size; p++) peers[p] = (const unsigned char *)(gathered.data() + (size_t)p * (size_t)n);
  } else {
    PHIP_REQUIRE(c->backend == BACKEND_RCCL, PHIP_ERR_SUP, "phip_comm_allreduce_host: this communicator has no host ring");
    PHIP_FLUSH(c->ctx);
    PHIP_CHECK_HIP(hipMemcpyAsync(c->d_stage, h_vals, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->ctx->stream));
    PHIP_CHECK_NCCL(g_rccl.AllReduce(c->d_stage, c->d_stage, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->nccl, c->ctx->stream));
    PHIP_CHECK_HIP(hipMemcpyAsync(h_vals, c->d_stage, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->ctx->stream));
    PHIP_CHECK_HIP(hipStreamSynchronize(c->ctx->stream));
    return 0;
  }
  double acc[PHIP_MAX_MVEC];
  for (int p = 0; p < c->size; p++) {
    double tmp[PHIP_MAX_MVEC];
    memcpy(tmp, peers[p], sizeof(double) * (size_t)n);
    for (int i = 0; i < n; i++) {
      if (p == 0) acc[i] = tmp[i];
      else if (op == 0) acc[i] += tmp[i];
      else acc[i] = (tmp[i] != tmp[i] || tmp[i] > acc[i]) ? tmp[i] : acc[i];
    }
  }
  memcpy(h_vals, acc, sizeof(double) * (size_t)n);
  return 0;
}

extern "C" int phip_comm_allgather_host(phip_comm c, const void *h_send, void *h_recv, size_t bytes)
{
  PHIP_REQUIRE(c && h_send && h_recv, PHIP_ERR_ARG_NULL, "phip_comm_allgather_host: null argument");
  if (c->backend == BACKEND_SELF || c->size == 1) { memcpy(h_recv, h_send, bytes); return 0; }
  if (c->backend == BACKEND_SHM) { /* through the ring, HR_PAYLOAD bytes per rank and round */
    const unsigned char *peers[1024];
    for (size_t off = 0; off < bytes; off += HR_PAYLOAD) {
      const size_t chunk = bytes - off < HR_PAYLOAD ? bytes - off : HR_PAYLOAD;
      int          rc = hr_round(c, (const char *)h_send + off, chunk, peers);
      if (rc) return rc;
      for (int p = 0; p < c->size; p++) memcpy((char *)h_recv + (size_t)p * bytes + off, peers[p], chunk);
    }
    return 0;
  }
  if (c->backend == BACKEND_CALLBACKS) {
    int rc = c->cb_allgather(c->cb_user, h_send, h_recv, bytes);
    PHIP_REQUIRE(rc == 0, PHIP_ERR_LIB, "phip_comm_allgather_host: the application's all-gather callback failed with %d", rc);
    return 0;
  }
  if (c->backend == BACKEND_THREADS) {
    ThreadGroup *g = c->tg;
    g->slot_a[(size_t)c->rank] = h_send;
    g->barrier();
    for (int p = 0; p < c->size; p++) memcpy((char *)h_recv + (size_t)p * bytes, g->slot_a[(size_t)p], bytes);
    g->barrier();
    return 0;
  }
  void *d_s = NULL, *d_r = NULL;
  PHIP_CHECK_HIP(hipSetDevice(c->ctx->device));
  PHIP_CHECK_HIP(hipMalloc(&d_s, bytes + 16));
  PHIP_CHECK_HIP(hipMalloc(&d_r, bytes * (size_t)c->size + 16));
  PHIP_CHECK_HIP(hipMemcpyAsync(d_s, h_send, bytes, hipMemcpyHostToDevice, c->ctx->stream));
  PHIP_CHECK_NCCL(g_rccl.AllGather(d_s, d_r, bytes, ncclInt8, c->nccl, c->ctx->stream));
  PHIP_CHECK_HIP(hipMemcpyAsync(h_recv, d_r, bytes * (size_t)c->size, hipMemcpyDeviceToHost, c->ctx->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(c->ctx->stream));
  PHIP_CHECK_HIP(hipFree(d_s));
  PHIP_CHECK_HIP(hipFree(d_r));
  return 0;
}

/* shm backend: all-gather of buffers of different lengths (lengths first, then the data padded to the longest) */
static int shm_allgatherv(phip_comm c, const void *mine, size_t my_bytes, std::vector<std::vector<unsigned char>> &all)
{
  std::vector<uint64_t> len((size_t)c->size);
  uint64_t              me = my_bytes;
  int                   rc = phip_comm_allgather_host(c, &me, len.data(), sizeof(uint64_t));
  if (rc) return rc;
  size_t mx = 8;
  for (int p = 0; p < c->size; p++) if (len[(size_t)p] > mx) mx = (size_t)len[(size_t)p];
  std::vector<unsigned char> send(mx, 0), recv(mx * (size_t)c->size);
  if (my_bytes) memcpy(send.data(), mine, my_bytes);
  if ((rc = phip_comm_allgather_host(c, send.data(), recv.data(), mx))) return rc;
  all.assign((size_t)c->size, std::vector<unsigned char>());
  for (int p = 0; p < c->size; p++) all[(size_t)p].assign(recv.begin() + (long)(mx * (size_t)p), recv.begin() + (long)(mx * (size_t)p + (size_t)len[(size_t)p]));
  return 0;
}

extern "C" int phip_comm_alltoallv_i32(phip_comm c, const int32_t *h_send, const int32_t *send_counts, int32_t *h_recv, const int32_t *recv_counts)
{
  PHIP_REQUIRE(c && send_counts && recv_counts, PHIP_ERR_ARG_NULL, "phip_comm_alltoallv_i32: null argument");
  std::vector<size_t> so((size_t)c->size + 1, 0), ro((size_t)c->size + 1, 0);
  for (int p = 0; p < c->size; p++) { so[(size_t)p + 1] = so[(size_t)p] + (size_t)send_counts[p]; ro[(size_t)p + 1] = ro[(size_t)p] + (size_t)recv_counts[p]; }
  if (c->backend == BACKEND_SELF) {
    PHIP_REQUIRE(send_counts[0] == recv_counts[0], PHIP_ERR_ARG_SIZ, "phip_comm_alltoallv_i32: self counts differ");
    if (send_counts[0]) memcpy(h_recv, h_send, sizeof(int32_t) * (size_t)send_counts[0]);
    return 0;
  }
  if (c->backend == BACKEND_SHM) { /* everyone gathers every rank's counts and lists, and keeps its own share */
    std::vector<int32_t> sc((size_t)c->size * (size_t)c->size);
    int rc = phip_comm_allgather_host(c, send_counts, sc.data(), sizeof(int32_t) * (size_t)c->size);
    if (rc) return rc;
    std::vector<std::vector<unsigned char>> all;
    if ((rc = shm_allgatherv(c, h_send, sizeof(int32_t) * so[(size_t)c->size], all))) return rc;
    for (int p = 0; p < c->size; p++) {
      size_t off = 0;
      for (int q = 0; q < c->rank; q++) off += (size_t)sc[(size_t)p * (size_t)c->size + (size_t)q];
      const int32_t cnt = sc[(size_t)p * (size_t)c->size + (size_t)c->rank];
      PHIP_REQUIRE(cnt == recv_counts[p], PHIP_ERR_ARG_SIZ, "phip_comm_alltoallv_i32: rank %d sends %d to %d, which expects %d", p, cnt, c->rank, recv_counts[p]);
      if (cnt) memcpy(h_recv + ro[(size_t)p], all[(size_t)p].data() + sizeof(int32_t) * off, sizeof(int32_t) * (size_t)cnt);
    }
    return 0;
  }
  if (c->backend == BACKEND_CALLBACKS) {
    int rc = c->cb_alltoallv(c->cb_user, h_send, send_counts, h_recv, recv_counts);
    PHIP_REQUIRE(rc == 0, PHIP_ERR_LIB, "phip_comm_alltoallv_i32: the application's all-to-all callback failed with %d", rc);
    return 0;
  }
  if (c->backend == BACKEND_THREADS) {
    ThreadGroup *g = c->tg;
    g->slot_a[(size_t)c->rank] = h_send;
    g->slot_b[(size_t)c->rank] = send_counts;
    g->barrier();
    for (int p = 0; p < c->size; p++) {
      const int32_t *pc = (const int32_t *)g->slot_b[(size_t)p];
      size_t         off = 0;
      for (int q = 0; q < c->rank; q++) off += (size_t)pc[q];
      PHIP_REQUIRE(pc[c->rank] == recv_counts[p], PHIP_ERR_ARG_SIZ, "phip_comm_alltoallv_i32: rank %d sends %d to %d, which expects %d", p, pc[c->rank], c->rank, recv_counts[p]);
      if (recv_counts[p]) memcpy(h_recv + ro[(size_t)p], (const int32_t *)g->slot_a[(size_t)p] + off, sizeof(int32_t) * (size_t)recv_counts[p]);
    }
    g->barrier();
    return 0;
  }
  int32_t *d_s = NULL, *d_r = NULL;
  PHIP_CHECK_HIP(hipSetDevice(c->ctx->device));
  PHIP_CHECK_HIP(hipMalloc((void **)&d_s, sizeof(int32_t) * (so[(size_t)c->size] + 4)));
  PHIP_CHECK_HIP(hipMalloc((void **)&d_r, sizeof(int32_t) * (ro[(size_t)c->size] + 4)));
  if (so[(size_t)c->size]) PHIP_CHECK_HIP(hipMemcpyAsync(d_s, h_send, sizeof(int32_t) * so[(size_t)c->size], hipMemcpyHostToDevice, c->ctx->stream));
  PHIP_CHECK_NCCL(g_rccl.GroupStart());
  for (int p = 0; p < c->size; p++) {
    if (send_counts[p]) PHIP_CHECK_NCCL(g_rccl.Send(d_s + so[(size_t)p], (size_t)send_counts[p], ncclInt32, p, c->nccl, c->ctx->stream));
    if (recv_counts[p]) PHIP_CHECK_NCCL(g_rccl.Recv(d_r + ro[(size_t)p], (size_t)recv_counts[p], ncclInt32, p, c->nccl, c->ctx->stream));
  }
  PHIP_CHECK_NCCL(g_rccl.GroupEnd());
  if (ro[(size_t)c->size]) PHIP_CHECK_HIP(hipMemcpyAsync(h_recv, d_r, sizeof(int32_t) * ro[(size_t)c->size], hipMemcpyDeviceToHost, c->ctx->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(c->ctx->stream));
  PHIP_CHECK_HIP(hipFree(d_s));
  PHIP_CHECK_HIP(hipFree(d_r));
  return 0;
}

/* ---- ghost exchange ------------------------------------------------------------------------------ */
/* Pack_1, vpscat.c:494-498: y[i] = x[idx[i]] -- all neighbours' segments in one launch */
__global__ __launch_bounds__(PHIP_THREADS) void halo_pack_kernel(int32_t n, const int32_t *__restrict__ idx, const double *__restrict__ x, double *__restrict__ y)
{
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n; i += stride) y[i] = x[idx[i]];
}

/* ---- peer-write ghost exchange (PHIP_HALO_IPC=1) ------------------------------------------------------------------------------
   Ranks of one node map each other's receive buffers (hipIpcGetMemHandle / hipIpcOpenMemHandle, exchanged once at phip_halo_create)
   and the pack kernel stores x[to_indices] straight into them: no send/receive kernels, no second stream, no staging -- one launch in
   phip_halo_begin, one in phip_halo_end.  Hand-off per exchange number seq (MI355X_MICROARCH.md, hand-offs):
     sender    every workgroup: stores, __threadfence_system, arrival counter; the last one stores seq into its flag word at each
               neighbour (system scope).  Before storing, the kernel waits until each neighbour acknowledged exchange seq - 1.
     receiver  (phip_halo_end) waits for the flags of all its neighbours, copies its receive buffer to lvec with loads that bypass
               the caches, and the last workgroup stores seq into its acknowledgement word at each sender.
   Every wait is bounded: a kernel that gives up leaves the context's error word set (the next reduction reports PHIP_ERR_LIB) and ends.
   The reverse exchange (MatMultTranspose) keeps the backend's own path.  Works between processes that share one GPU (the rehearsal
   of tests/test_host_gpu.py) and, by construction, between the GPUs of a node; the latter has not run yet. */
#define IPC_SPIN_MAX (1 << 24)
static std::atomic<unsigned long long> g_ipc_exchanges(0);
extern "C" int phip_halo_ipc_count(unsigned long long *n)
{
  PHIP_REQUIRE(n, PHIP_ERR_ARG_NULL, "phip_halo_ipc_count: null argument");
  *n = g_ipc_exchanges.load();
  return 0;
}
__device__ __forceinline__ unsigned long long ipc_load_sys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ bool ipc_wait_ge(const unsigned long long *p, unsigned long long want)
{
  for (int spin = 0; spin < IPC_SPIN_MAX; spin++) {
    if (ipc_load_sys(p) >= want) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return true; }
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}

__global__ __launch_bounds__(PHIP_THREADS) void halo_pack_ipc_kernel(int32_t nsend, int32_t to_n, const int32_t *__restrict__ to_starts, const int32_t *__restrict__ seg,
                                                                   const int32_t *__restrict__ idx, const double *__restrict__ x, double *const *__restrict__ peer_recv,
                                                                   unsigned long long *const *__restrict__ peer_flag, const unsigned long long *acks,
                                                                   unsigned long long seq, unsigned int *counter, unsigned int *err)
{
  __shared__ int last, bad;
  const int      t = threadIdx.x;
  if (t == 0) bad = 0;
  __syncthreads();
  if (t < to_n && !ipc_wait_ge(&acks[t], seq - 1)) { bad = 1; if (err) __hip_atomic_store(err, 0x1bc00001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  __syncthreads();
  /* a neighbour has not acknowledged the previous exchange: its buffer may still be read, so nothing is stored and no flag is raised --
     the neighbour's own wait gives up in turn and fills its ghost values with NaN; the error word fails the next host-visible call */
  if (bad) return;
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t k = (int64_t)blockIdx.x * PHIP_THREADS + t; k < nsend; k += stride) {
    const int i = seg[k];
    peer_recv[i][k - to_starts[i]] = x[idx[k]];
  }
  __threadfence_system(); /* this workgroup's stores are visible to the neighbours before it counts as arrived */
  __syncthreads();
  if (t == 0) last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  if (t < to_n) __hip_atomic_store(peer_flag[t], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (t == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(PHIP_THREADS) void halo_wait_ipc_kernel(int32_t nrecv, int32_t from_n, const unsigned long long *flags, unsigned long long seq,
                                                                   const double *recv, double *__restrict__ lvec, unsigned long long *const *__restrict__ peer_ack,
                                                                   unsigned int *counter, unsigned int *err)
{
  __shared__ int last, bad;
  const int      t = threadIdx.x;
  if (t == 0) bad = 0;
  __syncthreads();
  if (t < from_n && !ipc_wait_ge(&flags[t], seq)) { bad = 1; if (err) __hip_atomic_store(err, 0x1bc00002u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  if (bad) { /* a segment never arrived: the ghost values become NaN (a stale buffer must not pass for this exchange's) and nothing is acknowledged */
    for (int64_t k = (int64_t)blockIdx.x * PHIP_THREADS + t; k < nrecv; k += stride) lvec[k] = __longlong_as_double(0x7ff8000000000000ll);
    return;
  }
  for (int64_t k = (int64_t)blockIdx.x * PHIP_THREADS + t; k < nrecv; k += stride) /* written by another agent: not through this one's caches */
    lvec[k] = __longlong_as_double((long long)ipc_load_sys(reinterpret_cast<const unsigned long long *>(recv) + k));
  __syncthreads();
  if (t == 0) last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  if (t < from_n) __hip_atomic_store(peer_ack[t], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (t == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static int shm_allgatherv(phip_comm c, const void *mine, size_t my_bytes, std::vector<std::vector<unsigned char>> &all);

/* every rank publishes {handles of its receive buffer and sync block, its receive and send plans}; each rank maps what its neighbours
   published and works out where its segments and signals go.  Collective; fails on every rank alike or on none. */
static int halo_ipc_setup(phip_halo h)
{
  phip_comm c = h->comm;
  phip_ctx  ctx = c->ctx;
  h->ipc_why = NULL;
  { /* more neighbours than a workgroup has threads, anywhere: not this way (decided together) */
    int32_t              mine = h->to_n > PHIP_THREADS || h->from_n > PHIP_THREADS;
    std::vector<int32_t> every((size_t)c->size);
    int                  rc0 = phip_comm_allgather_host(c, &mine, every.data(), sizeof(int32_t));
    if (rc0) return rc0;
    for (int p = 0; p < c->size; p++) if (every[(size_t)p]) { h->ipc_why = "a rank has more than 256 neighbours"; return 0; }
  }
  const size_t nsync = (size_t)h->from_n + (size_t)h->to_n + 2;
  /* another GPU stores into these two while a kernel of this one polls and reads them: fine-grained device memory (coherent between
     agents inside a kernel; ordinary hipMalloc memory is only guaranteed coherent at kernel boundaries).  If the runtime refuses the
     flag, the exchange is not used (ipc_why) rather than run on memory that may serve stale lines. */
  {
    hipError_t e1 = hipExtMallocWithFlags((void **)&h->ipc_recv, sizeof(double) * ((size_t)h->nrecv + 2), hipDeviceMallocFinegrained);
    hipError_t e2 = e1 == hipSuccess ? hipExtMallocWithFlags((void **)&h->ipc_sync, sizeof(unsigned long long) * nsync, hipDeviceMallocFinegrained) : e1;
    if (e1 != hipSuccess || e2 != hipSuccess) {
      (void)hipGetLastError();
      if (h->ipc_recv) { hipFree(h->ipc_recv); h->ipc_recv = NULL; }
      h->ipc_sync = NULL;
    }
  }
  const int have_fine = h->ipc_recv && h->ipc_sync;
  if (have_fine) PHIP_CHECK_HIP(hipMemset(h->ipc_sync, 0, sizeof(unsigned long long) * nsync));
  PHIP_CHECK_HIP(hipDeviceSynchronize()); /* hipMemset of device memory does not wait: the words must be zero before a neighbour can store into them */
  struct Head { hipIpcMemHandle_t recv, sync; int32_t from_n, to_n, ok, pad; char busid[32]; };
  std::vector<unsigned char> rec(sizeof(Head) + sizeof(int32_t) * ((size_t)2 * h->from_n + 1 + (size_t)h->to_n));
  Head                       hd;
  memset(&hd, 0, sizeof(hd));
  hd.from_n = h->from_n; hd.to_n = h->to_n;
  if (hipDeviceGetPCIBusId(hd.busid, (int)sizeof(hd.busid), ctx->device) != hipSuccess) hd.busid[0] = 0; /* which GPU the buffers live on */
  hd.busid[sizeof(hd.busid) - 1] = 0;
  hd.ok = have_fine && hipIpcGetMemHandle(&hd.recv, h->ipc_recv) == hipSuccess && hipIpcGetMemHandle(&hd.sync, h->ipc_sync) == hipSuccess;
  (void)hipGetLastError();
  memcpy(rec.data(), &hd, sizeof(hd));
  int32_t *tail = reinterpret_cast<int32_t *>(rec.data() + sizeof(Head));
  for (int j = 0; j < h->from_n; j++) tail[j] = h->from_procs[(size_t)j];
  for (int j = 0; j <= h->from_n; j++) tail[h->from_n + j] = h->from_starts[(size_t)j];
  for (int i = 0; i < h->to_n; i++) tail[2 * h->from_n + 1 + i] = h->to_procs[(size_t)i];
  std::vector<std::vector<unsigned char>> all;
  int rc = shm_allgatherv(c, rec.data(), rec.size(), all);
  if (rc) return rc;
  int allok = 1;
  for (int p = 0; p < c->size; p++) { Head q; memcpy(&q, all[(size_t)p].data(), sizeof(q)); allok = allok && q.ok; }
  if (!allok) { h->ipc_why = "fine-grained device memory or its hipIpcGetMemHandle was refused on some rank"; return 0; }
  std::vector<double *>             peer_recv((size_t)h->to_n + 1, NULL);
  std::vector<unsigned long long *> peer_flag((size_t)h->to_n + 1, NULL), peer_ack((size_t)h->from_n + 1, NULL);
  std::vector<void *>               recv_of((size_t)c->size, NULL), sync_of((size_t)c->size, NULL);
  auto map_peer = [&](int p) -> int {
    if (sync_of[(size_t)p]) return 0;
    Head q;
    memcpy(&q, all[(size_t)p].data(), sizeof(q));
    /* a neighbour on another GPU of this process's view: its memory must be reachable from here BEFORE a kernel stores into it (an
       unreachable address is a memory fault, not an error code) -- ask, and switch peer access on rather than rely on the lazy flag */
    int pdev = -1;
    if (q.busid[0] && hipDeviceGetByPCIBusId(&pdev, q.busid) == hipSuccess && pdev != ctx->device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, ctx->device, pdev) != hipSuccess || !can) { (void)hipGetLastError(); return 3; }
      const hipError_t e = hipDeviceEnablePeerAccess(pdev, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return 3; }
    }
    (void)hipGetLastError();
    if (hipIpcOpenMemHandle(&recv_of[(size_t)p], q.recv, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return 1; }
    h->ipc_opened.push_back(recv_of[(size_t)p]);
    if (hipIpcOpenMemHandle(&sync_of[(size_t)p], q.sync, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); sync_of[(size_t)p] = NULL; return 1; }
    h->ipc_opened.push_back(sync_of[(size_t)p]);
    return 0;
  };
  int bad = 0;
  for (int i = 0; i < h->to_n && !bad; i++) { /* neighbour p receives from me: its plan says where */
    const int p = h->to_procs[(size_t)i];
    if ((bad = map_peer(p))) break;
    Head q;
    memcpy(&q, all[(size_t)p].data(), sizeof(q));
    const int32_t *pt = reinterpret_cast<const int32_t *>(all[(size_t)p].data() + sizeof(Head));
    int            j = -1;
    for (int k = 0; k < q.from_n; k++) if (pt[k] == c->rank) j = k;
    if (j < 0 || pt[q.from_n + j + 1] - pt[q.from_n + j] != h->to_starts[(size_t)i + 1] - h->to_starts[(size_t)i]) { bad = 2; break; }
    peer_recv[(size_t)i] = (double *)recv_of[(size_t)p] + pt[q.from_n + j];
    peer_flag[(size_t)i] = (unsigned long long *)sync_of[(size_t)p] + j;
  }
  for (int j = 0; j < h->from_n && !bad; j++) { /* neighbour p sends to me: my acknowledgement goes behind its flags */
    const int p = h->from_procs[(size_t)j];
    if ((bad = map_peer(p))) break;
    Head q;
    memcpy(&q, all[(size_t)p].data(), sizeof(q));
    const int32_t *pt = reinterpret_cast<const int32_t *>(all[(size_t)p].data() + sizeof(Head));
    int            i = -1;
    for (int k = 0; k < q.to_n; k++) if (pt[2 * q.from_n + 1 + k] == c->rank) i = k;
    if (i < 0) { bad = 2; break; }
    peer_ack[(size_t)j] = (unsigned long long *)sync_of[(size_t)p] + q.from_n + i;
  }
  /* agree before anyone relies on it */
  std::vector<int32_t> bads((size_t)c->size);
  int32_t              mybad = bad;
  if ((rc = phip_comm_allgather_host(c, &mybad, bads.data(), sizeof(int32_t)))) return rc;
  for (int p = 0; p < c->size; p++)
    if (bads[(size_t)p]) { h->ipc_why = bads[(size_t)p] == 1 ? "hipIpcOpenMemHandle failed on some rank" : bads[(size_t)p] == 3 ? "two neighbours' GPUs have no peer access" : "the plans of two neighbours do not match"; return 0; }
  std::vector<int32_t> seg((size_t)h->nsend + 1, 0);
  for (int i = 0; i < h->to_n; i++) for (int32_t k = h->to_starts[(size_t)i]; k < h->to_starts[(size_t)i + 1]; k++) seg[(size_t)k] = i;
  PHIP_CHECK_HIP(hipMalloc((void **)&h->d_send_seg, sizeof(int32_t) * seg.size()));
  PHIP_CHECK_HIP(hipMalloc((void **)&h->d_to_starts, sizeof(int32_t) * ((size_t)h->to_n + 1)));
  PHIP_CHECK_HIP(hipMalloc((void **)&h->d_peer_recv, sizeof(void *) * peer_recv.size()));
  PHIP_CHECK_HIP(hipMalloc((void **)&h->d_peer_flag, sizeof(void *) * peer_flag.size()));
  PHIP_CHECK_HIP(hipMalloc((void **)&h->d_peer_ack, sizeof(void *) * peer_ack.size()));
  PHIP_CHECK_HIP(hipMemcpy(h->d_send_seg, seg.data(), sizeof(int32_t) * seg.size(), hipMemcpyHostToDevice));
  PHIP_CHECK_HIP(hipMemcpy(h->d_to_starts, h->to_starts.data(), sizeof(int32_t) * ((size_t)h->to_n + 1), hipMemcpyHostToDevice));
  PHIP_CHECK_HIP(hipMemcpy(h->d_peer_recv, peer_recv.data(), sizeof(void *) * peer_recv.size(), hipMemcpyHostToDevice));
  PHIP_CHECK_HIP(hipMemcpy(h->d_peer_flag, peer_flag.data(), sizeof(void *) * peer_flag.size(), hipMemcpyHostToDevice));
  PHIP_CHECK_HIP(hipMemcpy(h->d_peer_ack, peer_ack.data(), sizeof(void *) * peer_ack.size(), hipMemcpyHostToDevice));
  h->ipc_seq = 0;
  if ((rc = phip_comm_barrier(c))) return rc; /* nobody stores into a buffer before its owner has zeroed the sync words and everyone has mapped */
  /* Three trial exchanges before anything relies on the path: every rank sends a constant of its own per round (1000 rank + round), so
     a receiver knows what each segment must hold without knowing its neighbour's indices.  Exercises mapping, flags, acknowledgements
     and the reuse of the buffers; the ranks then agree on the verdict. */
  int32_t maxidx = 0;
  {
    std::vector<int32_t> idx((size_t)h->nsend + 1);
    if (h->nsend) PHIP_CHECK_HIP(hipMemcpy(idx.data(), h->d_to_indices, sizeof(int32_t) * (size_t)h->nsend, hipMemcpyDeviceToHost));
    for (int32_t k = 0; k < h->nsend; k++) maxidx = std::max(maxidx, idx[(size_t)k]);
  }
  double *tx = NULL, *tl = NULL;
  PHIP_CHECK_HIP(hipMalloc((void **)&tx, sizeof(double) * ((size_t)maxidx + 2)));
  PHIP_CHECK_HIP(hipMalloc((void **)&tl, sizeof(double) * ((size_t)h->nrecv + 2)));
  std::vector<double> hx((size_t)maxidx + 1), hl((size_t)h->nrecv + 1);
  int32_t             good = 1;
  h->ipc_on = 1;
  for (int round = 1; round <= 3; round++) {
    std::fill(hx.begin(), hx.end(), 1000.0 * c->rank + round);
    PHIP_CHECK_HIP(hipMemcpy(tx, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice));
    PHIP_CHECK_HIP(hipMemsetAsync(tl, 0xff, sizeof(double) * ((size_t)h->nrecv + 1), ctx->stream)); /* on the stream of the exchange: it does not wait for the null stream */
    int r1 = phip_halo_begin(h, tx, tl);
    int r2 = r1 ? r1 : phip_halo_end(h);
    if (r2 || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); good = 0; }
    else if (h->nrecv) {
      PHIP_CHECK_HIP(hipMemcpy(hl.data(), tl, sizeof(double) * (size_t)h->nrecv, hipMemcpyDeviceToHost));
      for (int j = 0; j < h->from_n && good; j++)
        for (int32_t k = h->from_starts[(size_t)j]; k < h->from_starts[(size_t)j + 1]; k++)
          if (hl[(size_t)k] != 1000.0 * h->from_procs[(size_t)j] + round) { good = 0; break; }
    }
    if (ctx->h_err && __atomic_load_n(ctx->h_err, __ATOMIC_ACQUIRE)) { __atomic_store_n(ctx->h_err, 0u, __ATOMIC_RELAXED); good = 0; }
  }
  h->lvec = NULL;
  hipFree(tx); hipFree(tl);
  std::vector<int32_t> goods((size_t)c->size);
  if ((rc = phip_comm_allgather_host(c, &good, goods.data(), sizeof(int32_t)))) return rc;
  for (int p = 0; p < c->size; p++) if (!goods[(size_t)p]) { h->ipc_on = 0; h->ipc_why = "the trial exchanges did not deliver what was sent"; }
  return 0;
}

extern "C" int phip_halo_create(phip_comm comm, int32_t to_n, const int32_t *to_procs, const int32_t *to_starts, const int32_t *to_indices,
                                int32_t from_n, const int32_t *from_procs, const int32_t *from_starts, phip_halo *out)
{
  PHIP_REQUIRE(comm && out, PHIP_ERR_ARG_NULL, "phip_halo_create: null argument");
  PHIP_REQUIRE(to_n >= 0 && from_n >= 0 && to_n < comm->size + 1 && from_n < comm->size + 1, PHIP_ERR_ARG_OUTOFRANGE, "phip_halo_create: %d sends / %d receives on %d ranks", to_n, from_n, comm->size);
  PHIP_REQUIRE((!to_n || (to_procs && to_starts && to_indices)) && (!from_n || (from_procs && from_starts)), PHIP_ERR_ARG_NULL, "phip_halo_create: null plan array");
  phip_halo h = new phip_halo_s();
  h->comm = comm; h->to_n = to_n; h->from_n = from_n;
  h->to_procs.assign(to_procs, to_procs + to_n);
  h->to_starts.assign(1, 0); if (to_n) h->to_starts.assign(to_starts, to_starts + to_n + 1);
  h->from_procs.assign(from_procs, from_procs + from_n);
  h->from_starts.assign(1, 0); if (from_n) h->from_starts.assign(from_starts, from_starts + from_n + 1);
  h->nsend = h->to_starts[(size_t)to_n]; h->nrecv = h->from_starts[(size_t)from_n];
  for (int i = 0; i < to_n; i++) PHIP_REQUIRE(to_procs[i] >= 0 && to_procs[i] < comm->size && to_procs[i] != comm->rank && to_starts[i + 1] >= to_starts[i], PHIP_ERR_ARG_OUTOFRANGE, "phip_halo_create: bad send entry %d", i);
  for (int i = 0; i < from_n; i++) PHIP_REQUIRE(from_procs[i] >= 0 && from_procs[i] < comm->size && from_procs[i] != comm->rank && from_starts[i + 1] >= from_starts[i], PHIP_ERR_ARG_OUTOFRANGE, "phip_halo_create: bad receive entry %d", i);
  for (int i = 0; i < h->nsend; i++) PHIP_REQUIRE(to_indices[i] >= 0, PHIP_ERR_ARG_OUTOFRANGE, "phip_halo_create: negative send index");
  phip_ctx ctx = comm->ctx;
  h->d_to_indices = NULL; h->d_sendbuf = NULL; h->lvec = NULL;
  h->ntgt = 0; h->d_tgt = h->d_tptr = h->d_tpos = NULL; h->rev_lvec = NULL;
  h->ipc_why = NULL; h->ipc_on = 0; h->ipc_seq = 0; h->ipc_recv = NULL; h->ipc_sync = NULL; h->d_send_seg = NULL; h->d_to_starts = NULL; h->d_peer_recv = NULL; h->d_peer_flag = NULL; h->d_peer_ack = NULL;
  if (ctx->device < 0) { *out = h; return 0; } /* planning-only: the plan stays on the host */
  PHIP_CHECK_HIP(hipSetDevice(ctx->device));
  PHIP_CHECK_HIP(hipMalloc((void **)&h->d_to_indices, sizeof(int32_t) * ((size_t)h->nsend + 4)));
  PHIP_CHECK_HIP(hipMalloc((void **)&h->d_sendbuf, sizeof(double) * ((size_t)h->nsend + 2)));
  if (h->nsend) PHIP_CHECK_HIP(hipMemcpyAsync(h->d_to_indices, to_indices, sizeof(int32_t) * (size_t)h->nsend, hipMemcpyHostToDevice, ctx->stream));
  {
    std::vector<std::pair<int32_t, int32_t>> byt((size_t)h->nsend);
    for (int32_t k = 0; k < h->nsend; k++) byt[(size_t)k] = std::make_pair(to_indices[k], k);
    std::sort(byt.begin(), byt.end());
    std::vector<int32_t> tgt, tptr(1, 0), tpos((size_t)h->nsend + 1);
    for (int32_t k = 0; k < h->nsend; k++) {
      if (!k || byt[(size_t)k].first != byt[(size_t)k - 1].first) { if (k) tptr.push_back(k); tgt.push_back(byt[(size_t)k].first); }
      tpos[(size_t)k] = byt[(size_t)k].second;
    }
    if (h->nsend) tptr.push_back(h->nsend);
    h->ntgt = (int32_t)tgt.size();
    PHIP_CHECK_HIP(hipMalloc((void **)&h->d_tgt, sizeof(int32_t) * ((size_t)h->ntgt + 4)));
    PHIP_CHECK_HIP(hipMalloc((void **)&h->d_tptr, sizeof(int32_t) * ((size_t)h->ntgt + 4)));
    PHIP_CHECK_HIP(hipMalloc((void **)&h->d_tpos, sizeof(int32_t) * ((size_t)h->nsend + 4)));
    if (h->nsend) {
      PHIP_CHECK_HIP(hipMemcpyAsync(h->d_tgt, tgt.data(), sizeof(int32_t) * (size_t)h->ntgt, hipMemcpyHostToDevice, ctx->stream));
      PHIP_CHECK_HIP(hipMemcpyAsync(h->d_tptr, tptr.data(), sizeof(int32_t) * ((size_t)h->ntgt + 1), hipMemcpyHostToDevice, ctx->stream));
      PHIP_CHECK_HIP(hipMemcpyAsync(h->d_tpos, tpos.data(), sizeof(int32_t) * (size_t)h->nsend, hipMemcpyHostToDevice, ctx->stream));
    }
    PHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream)); /* the staging vectors go out of scope here */
  }
  PHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (comm->backend == BACKEND_SHM && comm->size > 1) { /* who sends how much to whom: the staged exchange finds its segments with it */
    std::vector<int32_t> row((size_t)comm->size, 0);
    for (int i = 0; i < to_n; i++) row[(size_t)to_procs[i]] = to_starts[i + 1] - to_starts[i];
    h->cnt.assign((size_t)comm->size * (size_t)comm->size, 0);
    int rc = phip_comm_allgather_host(comm, row.data(), h->cnt.data(), sizeof(int32_t) * (size_t)comm->size);
    if (rc) return rc;
  }
  PHIP_CHECK_HIP(hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming));
  PHIP_CHECK_HIP(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
  /* PHIP_HALO_IPC (every rank alike): 1 = the peer-write exchange between the processes of one node (backends rccl and shm), an error if
     it cannot be set up or its trial exchanges fail; unset or 0 = the backend's own exchange (grouped ncclSend/ncclRecv).  Opt-in only:
     the path has run between processes sharing one GPU, never yet between two GPUs, and three good trial exchanges do not prove that
     no later exchange reads a stale line -- it stays off until it has been checked on real xGMI (bench.py's matmult_check is the gate). */
  {
    const char *e = getenv("PHIP_HALO_IPC");
    const int   mode = e ? (atoi(e) ? 1 : 0) : -1;
    const bool  can = comm->size > 1 && (comm->backend == BACKEND_RCCL || comm->backend == BACKEND_SHM);
    const bool  want = can && mode == 1;
    if (want) {
      int rc = halo_ipc_setup(h);
      if (rc) return rc;
      if (!h->ipc_on && mode == 1) PHIP_REQUIRE(0, PHIP_ERR_LIB, "phip_halo_create: PHIP_HALO_IPC=1 but the peer-write exchange is not available: %s", h->ipc_why ? h->ipc_why : "?");
      if (!h->ipc_on && getenv("PHIP_VERBOSE")) fprintf(stderr, "[phip] rank %d: peer-write ghost exchange not used (%s); RCCL send/recv instead\n", comm->rank, h->ipc_why ? h->ipc_why : "?");
    }
  }
  *out = h;
  return 0;
}

extern "C" int phip_halo_destroy(phip_halo h)
{
  if (!h) return 0;
  if (h->comm->ctx->device < 0) { delete h; return 0; }
  hipSetDevice(h->comm->ctx->device);
  hipStreamSynchronize(h->comm->ctx->stream);
  hipStreamSynchronize(h->comm->ctx->comm_stream);
  hipFree(h->d_to_indices); hipFree(h->d_sendbuf); hipFree(h->d_tgt); hipFree(h->d_tptr); hipFree(h->d_tpos);
  if (h->ipc_on) phip_comm_barrier(h->comm); /* no neighbour stores into a buffer that is about to go */
  for (void *q : h->ipc_opened) hipIpcCloseMemHandle(q);
  hipFree(h->ipc_recv); hipFree(h->ipc_sync); hipFree(h->d_send_seg); hipFree(h->d_to_starts); hipFree(h->d_peer_recv); hipFree(h->d_peer_flag); hipFree(h->d_peer_ack);
  hipEventDestroy(h->ev_packed); hipEventDestroy(h->ev_done);
  delete h;
  return 0;
}

/* shm backend: one exchange staged through the host.  forward: every rank contributes its packed send buffer and picks, from each
   neighbour's, the segment meant for it (into lvec).  reverse: every rank contributes its lvec and picks, from each neighbour's, the
   segment of entries it owns (into the send buffer, which halo_unpack_add_kernel then adds into y). */
static int shm_exchange(phip_halo h, int reverse, const double *d_src, size_t nsrc, double *d_dst)
{
  phip_comm c = h->comm;
  phip_ctx  ctx = c->ctx;
  const size_t P = (size_t)c->size, me = (size_t)c->rank;
  std::vector<double> mine(nsrc + 1);
  if (nsrc) PHIP_CHECK_HIP(hipMemcpyAsync(mine.data(), d_src, sizeof(double) * nsrc, hipMemcpyDeviceToHost, ctx->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<std::vector<unsigned char>> all;
  int rc = shm_allgatherv(c, mine.data(), sizeof(double) * nsrc, all);
  if (rc) return rc;
  const int                   npeer = reverse ? h->to_n : h->from_n;
  const std::vector<int32_t> &procs = reverse ? h->to_procs : h->from_procs, &starts = reverse ? h->to_starts : h->from_starts;
  for (int i = 0; i < npeer; i++) {
    const size_t  p = (size_t)procs[(size_t)i];
    const int32_t cnt = starts[(size_t)i + 1] - starts[(size_t)i];
    size_t        off = 0; /* forward: what p sends to the ranks before me; reverse: what p received from the ranks before me */
    for (size_t q = 0; q < me; q++) off += (size_t)(reverse ? h->cnt[q * P + p] : h->cnt[p * P + q]);
    const int32_t have = reverse ? h->cnt[me * P + p] : h->cnt[p * P + me];
    PHIP_REQUIRE(have == cnt && sizeof(double) * (off + (size_t)cnt) <= all[p].size(), PHIP_ERR_ARG_SIZ, "halo exchange: rank %d expects %d entries from rank %d, which holds %d for it", c->rank, cnt, (int)p, have);
    if (cnt) PHIP_CHECK_HIP(hipMemcpyAsync(d_dst + starts[(size_t)i], all[p].data() + sizeof(double) * off, sizeof(double) * (size_t)cnt, hipMemcpyHostToDevice, ctx->stream));
  }
  PHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream)); /* the staging buffers go out of scope */
  return 0;
}

extern "C" int phip_halo_begin(phip_halo h, const double *d_x_local, double *d_lvec)
{
  PHIP_REQUIRE(h && (d_x_local || !h->nsend) && (d_lvec || !h->nrecv), PHIP_ERR_ARG_NULL, "phip_halo_begin: null argument");
  phip_comm c = h->comm;
  phip_ctx  ctx = c->ctx;
  h->lvec = d_lvec;
  if (c->size == 1) return 0;
  PHIP_NEED_DEVICE(ctx, "phip_halo_begin");
  PHIP_FLUSH(ctx);
  PHIP_REQUIRE(c->backend != BACKEND_CALLBACKS, PHIP_ERR_SUP, "phip_halo_begin: the callbacks backend carries set-up traffic only");
  if (h->ipc_on) { /* one launch: wait for the neighbours' acknowledgement of the last exchange, store into their buffers, raise the flags */
    h->ipc_seq++;
    g_ipc_exchanges++;
    hipLaunchKernelGGL(halo_pack_ipc_kernel, dim3(phip_grid_for(h->nsend > 0 ? h->nsend : 1, 1, 64)), dim3(PHIP_THREADS), 0, ctx->stream, h->nsend, h->to_n, h->d_to_starts,
                       h->d_send_seg, h->d_to_indices, d_x_local, h->d_peer_recv, h->d_peer_flag, h->ipc_sync + h->from_n, h->ipc_seq,
                       reinterpret_cast<unsigned int *>(h->ipc_sync + h->from_n + h->to_n), ctx->h_err_dev);
    PHIP_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (h->nsend) {
    hipLaunchKernelGGL(halo_pack_kernel, dim3(phip_grid_for(h->nsend, 1, 1024)), dim3(PHIP_THREADS), 0, ctx->stream, h->nsend, h->d_to_indices, d_x_local, h->d_sendbuf);
    PHIP_CHECK_HIP(hipGetLastError());
  }
  if (c->backend == BACKEND_RCCL) {
    /* the exchange may start once the pack (and everything before it on the compute stream, e.g. the previous
       product that still reads lvec) has finished.  Order of the two streams over consecutive products: the pack of product k + 1
       overwrites d_sendbuf only after phip_halo_end of product k made the compute stream wait for ev_done, i.e. for the sends of
       product k; the receives of product k + 1 overwrite lvec only after ev_packed of k + 1, recorded behind the off-diagonal
       product of k on the compute stream.  Neighbours with nothing to exchange are skipped on both sides (the counts of a plan
       agree pairwise, phip_halo_create's callers build them from one all-to-all), an empty group is legal. */
    PHIP_CHECK_HIP(hipSetDevice(ctx->device)); /* RCCL launches on the current device: the caller's thread may have another one current */
    PHIP_CHECK_HIP(hipEventRecord(h->ev_packed, ctx->stream));
    PHIP_CHECK_HIP(hipStreamWaitEvent(ctx->comm_stream, h->ev_packed, 0));
    if (h->to_n || h->from_n) {
      PHIP_CHECK_NCCL(g_rccl.GroupStart());
      for (int i = 0; i < h->to_n; i++) {
        const size_t cnt = (size_t)(h->to_starts[(size_t)i + 1] - h->to_starts[(size_t)i]);
        if (cnt) PHIP_CHECK_NCCL(g_rccl.Send(h->d_sendbuf + h->to_starts[(size_t)i], cnt, ncclDouble, h->to_procs[(size_t)i], c->nccl, ctx->comm_stream));
      }
      for (int i = 0; i < h->from_n; i++) {
        const size_t cnt = (size_t)(h->from_starts[(size_t)i + 1] - h->from_starts[(size_t)i]);
        if (cnt) PHIP_CHECK_NCCL(g_rccl.Recv(d_lvec + h->from_starts[(size_t)i], cnt, ncclDouble, h->from_procs[(size_t)i], c->nccl, ctx->comm_stream));
      }
      PHIP_CHECK_NCCL(g_rccl.GroupEnd());
    }
    PHIP_CHECK_HIP(hipEventRecord(h->ev_done, ctx->comm_stream));
    return 0;
  }
  if (c->backend == BACKEND_SHM) return shm_exchange(h, 0, h->d_sendbuf, (size_t)h->nsend, d_lvec);
  /* threads: publish the packed buffer; the copies happen in phip_halo_end */
  PHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  c->tg->slot_a[(size_t)c->rank] = h;
  return 0;
}

extern "C" int phip_halo_end(phip_halo h)
{
  PHIP_REQUIRE(h, PHIP_ERR_ARG_NULL, "phip_halo_end: null halo");
  phip_comm c = h->comm;
  phip_ctx  ctx = c->ctx;
  if (c->size == 1) return 0;
  PHIP_FLUSH(ctx);
  if (h->ipc_on) { /* wait for the neighbours' flags, copy what they stored to lvec, acknowledge */
    hipLaunchKernelGGL(halo_wait_ipc_kernel, dim3(phip_grid_for(h->nrecv > 0 ? h->nrecv : 1, 1, 64)), dim3(PHIP_THREADS), 0, ctx->stream, h->nrecv, h->from_n, h->ipc_sync,
                       h->ipc_seq, h->ipc_recv, h->lvec, h->d_peer_ack, reinterpret_cast<unsigned int *>(h->ipc_sync + h->from_n + h->to_n) + 1, ctx->h_err_dev);
    PHIP_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (c->backend == BACKEND_RCCL) {
    PHIP_CHECK_HIP(hipStreamWaitEvent(ctx->stream, h->ev_done, 0));
    return 0;
  }
  if (c->backend == BACKEND_SHM) return 0; /* done in phip_halo_begin */
  ThreadGroup *g = c->tg;
  g->barrier();
  for (int i = 0; i < h->from_n; i++) {
    const phip_halo_s *peer = (const phip_halo_s *)g->slot_a[(size_t)h->from_procs[(size_t)i]];
    const int32_t      cnt = h->from_starts[(size_t)i + 1] - h->from_starts[(size_t)i];
    int                seg = -1;
    for (int k = 0; k < peer->to_n; k++) if (peer->to_procs[(size_t)k] == c->rank) seg = k;
    PHIP_REQUIRE(seg >= 0 && peer->to_starts[(size_t)seg + 1] - peer->to_starts[(size_t)seg] == cnt, PHIP_ERR_ARG_SIZ,
                 "phip_halo_end: rank %d expects %d entries from rank %d, plan mismatch", c->rank, cnt, h->from_procs[(size_t)i]);
    if (cnt) PHIP_CHECK_HIP(hipMemcpyAsync(h->lvec + h->from_starts[(size_t)i], peer->d_sendbuf + peer->to_starts[(size_t)seg], sizeof(double) * (size_t)cnt, hipMemcpyDeviceToDevice, ctx->stream));
  }
  PHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  g->barrier();
  return 0;
}

/* ---- reverse ghost exchange with ADD_VALUES (the scatter of MatMultTranspose_MPIAIJ, mpiaij.c:1041-1046, 1111-1115) ------------------
   The roles swap: every segment of lvec goes back to the rank that owns those entries (VecScatterBegin_1 with SCATTER_REVERSE,
   vpscat.h:24-30 swaps to/from), and the owner adds what it receives into y[to.indices] (UnPack_1 with ADD_VALUES, vpscat.c:512-515).
   The reference adds the messages in arrival order (MPI_Waitany, vpscat.h:165); here each target entry adds its contributions in
   ascending neighbour rank, one thread per distinct target: deterministic and free of atomics. */
__global__ __launch_bounds__(PHIP_THREADS) void halo_unpack_add_kernel(int32_t nt, const int32_t *__restrict__ tgt, const int32_t *__restrict__ tptr,
                                                                     const int32_t *__restrict__ tpos, const double *__restrict__ buf, double *__restrict__ y)
{
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t t = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; t < nt; t += stride) {
    double s = y[tgt[t]];
    for (int32_t k = tptr[t]; k < tptr[t + 1]; k++) s += buf[tpos[k]];
    y[tgt[t]] = s;
  }
}

extern "C" int phip_halo_reverse_begin(phip_halo h, const double *d_lvec)
{
  PHIP_REQUIRE(h && (d_lvec || !h->nrecv), PHIP_ERR_ARG_NULL, "phip_halo_reverse_begin: null argument");
  phip_comm c = h->comm;
  phip_ctx  ctx = c->ctx;
  h->rev_lvec = d_lvec;
  if (c->size == 1) return 0;
  PHIP_NEED_DEVICE(ctx, "phip_halo_reverse_begin");
  PHIP_FLUSH(ctx);
  PHIP_REQUIRE(c->backend != BACKEND_CALLBACKS, PHIP_ERR_SUP, "phip_halo_reverse_begin: the callbacks backend carries set-up traffic only");
  if (c->backend == BACKEND_RCCL) {
    /* lvec is complete once everything queued on the compute stream so far has run; the receive buffer is the forward send buffer
       (one scatter context carries one exchange at a time, vscat.c:1670) */
    PHIP_CHECK_HIP(hipSetDevice(ctx->device));
    PHIP_CHECK_HIP(hipEventRecord(h->ev_packed, ctx->stream));
    PHIP_CHECK_HIP(hipStreamWaitEvent(ctx->comm_stream, h->ev_packed, 0));
    if (h->to_n || h->from_n) {
      PHIP_CHECK_NCCL(g_rccl.GroupStart());
      for (int i = 0; i < h->from_n; i++) {
        const size_t cnt = (size_t)(h->from_starts[(size_t)i + 1] - h->from_starts[(size_t)i]);
        if (cnt) PHIP_CHECK_NCCL(g_rccl.Send(d_lvec + h->from_starts[(size_t)i], cnt, ncclDouble, h->from_procs[(size_t)i], c->nccl, ctx->comm_stream));
      }
      for (int i = 0; i < h->to_n; i++) {
        const size_t cnt = (size_t)(h->to_starts[(size_t)i + 1] - h->to_starts[(size_t)i]);
        if (cnt) PHIP_CHECK_NCCL(g_rccl.Recv(h->d_sendbuf + h->to_starts[(size_t)i], cnt, ncclDouble, h->to_procs[(size_t)i], c->nccl, ctx->comm_stream));
      }
      PHIP_CHECK_NCCL(g_rccl.GroupEnd());
    }
    PHIP_CHECK_HIP(hipEventRecord(h->ev_done, ctx->comm_stream));
    return 0;
  }
  if (c->backend == BACKEND_SHM) return shm_exchange(h, 1, d_lvec, (size_t)h->nrecv, h->d_sendbuf);
  /* threads: lvec must be complete before a peer copies from it in phip_halo_reverse_end */
  PHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  c->tg->slot_a[(size_t)c->rank] = h;
  return 0;
}

extern "C" int phip_halo_reverse_end(phip_halo h, double *d_y)
{
  PHIP_REQUIRE(h && (d_y || !h->nsend), PHIP_ERR_ARG_NULL, "phip_halo_reverse_end: null argument");
  phip_comm c = h->comm;
  phip_ctx  ctx = c->ctx;
  if (c->size == 1) return 0;
  PHIP_FLUSH(ctx);
  if (c->backend == BACKEND_RCCL) PHIP_CHECK_HIP(hipStreamWaitEvent(ctx->stream, h->ev_done, 0));
  else if (c->backend == BACKEND_THREADS) {
    ThreadGroup *g = c->tg;
    g->barrier();
    for (int i = 0; i < h->to_n; i++) {
      const phip_halo_s *peer = (const phip_halo_s *)g->slot_a[(size_t)h->to_procs[(size_t)i]];
      const int32_t      cnt = h->to_starts[(size_t)i + 1] - h->to_starts[(size_t)i];
      int                seg = -1;
      for (int k = 0; k < peer->from_n; k++) if (peer->from_procs[(size_t)k] == c->rank) seg = k;
      PHIP_REQUIRE(seg >= 0 && peer->from_starts[(size_t)seg + 1] - peer->from_starts[(size_t)seg] == cnt, PHIP_ERR_ARG_SIZ,
                   "phip_halo_reverse_end: rank %d expects %d entries back from rank %d, plan mismatch", c->rank, cnt, h->to_procs[(size_t)i]);
      if (cnt) PHIP_CHECK_HIP(hipMemcpyAsync(h->d_sendbuf + h->to_starts[(size_t)i], peer->rev_lvec + peer->from_starts[(size_t)seg], sizeof(double) * (size_t)cnt, hipMemcpyDeviceToDevice, ctx->stream));
    }
    PHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    g->barrier(); /* every rank has read its peers' lvec before anyone overwrites it */
  }
  if (h->ntgt) {
    hipLaunchKernelGGL(halo_unpack_add_kernel, dim3(phip_grid_for(h->ntgt, 1, 1024)), dim3(PHIP_THREADS), 0, ctx->stream, h->ntgt, h->d_tgt, h->d_tptr, h->d_tpos, h->d_sendbuf, d_y);
    PHIP_CHECK_HIP(hipGetLastError());
  }
  return 0;
}
