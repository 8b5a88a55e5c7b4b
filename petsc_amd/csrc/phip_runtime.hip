// Context, memory and error plumbing of libpetschip (see include/petschip.h).
#include "phip_internal.h"
#include "phip_reduce.h"
#include <stdarg.h>
#include <stdlib.h>
#include <vector>

struct phip_prof_s {
  std::vector<hipEvent_t> ev[4][2]; /* [kind][start|stop]; kinds: include/petschip.h phip_profile_read */
  size_t                  used[4];
};

static thread_local char g_err[512] = "";

void phip_set_error(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *phip_last_error(void) { return g_err; }

extern "C" int phip_finalize(phip_ctx c);
extern "C" int phip_init(int device, phip_ctx *out)
{
  PHIP_REQUIRE(out, PHIP_ERR_ARG_NULL, "phip_init: null output");
  if (device == -1) { /* planning-only context: see include/petschip.h */
    phip_ctx p = (phip_ctx)calloc(1, sizeof(*p));
    PHIP_REQUIRE(p, PHIP_ERR_MEM, "phip_init: out of host memory");
    p->device = -1;
    *out = p;
    return 0;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    phip_set_error("phip_init: no usable HIP device (%s); libpetschip has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "count 0");
    return PHIP_ERR_LIB;
  }
  PHIP_REQUIRE(device >= 0 && device < ndev, PHIP_ERR_ARG_OUTOFRANGE, "phip_init: device %d of %d", device, ndev);
  PHIP_CHECK_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  PHIP_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  phip_ctx c = (phip_ctx)calloc(1, sizeof(*c));
  PHIP_REQUIRE(c, PHIP_ERR_MEM, "phip_init: out of host memory");
  c->device   = device;
  c->cu_count = prop.multiProcessorCount;
  PHIP_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  PHIP_CHECK_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  PHIP_CHECK_HIP(hipMalloc((void **)&c->d_partials, sizeof(unsigned long long) * 2 * PHIP_MAX_MVEC * PHIP_RED_BLOCKS));
  PHIP_CHECK_HIP(hipMemsetAsync(c->d_partials, 0, sizeof(unsigned long long) * 2 * PHIP_MAX_MVEC * PHIP_RED_BLOCKS, c->stream));
  c->pseq = 0;
  c->ew_n = 0; c->ew_batching = !(getenv("PHIP_EW_BATCH") && !atoi(getenv("PHIP_EW_BATCH")));
  PHIP_CHECK_HIP(hipMalloc((void **)&c->d_result, sizeof(double) * PHIP_MAX_MVEC));
  PHIP_CHECK_HIP(hipHostMalloc((void **)&c->h_result, sizeof(double) * PHIP_MAX_MVEC, hipHostMallocDefault));
  PHIP_CHECK_HIP(hipHostMalloc((void **)&c->h_ll, sizeof(unsigned long long) * 2 * PHIP_MAX_MVEC, hipHostMallocMapped | hipHostMallocCoherent));
  PHIP_CHECK_HIP(hipHostGetDevicePointer((void **)&c->h_ll_dev, c->h_ll, 0));
  memset(c->h_ll, 0, sizeof(unsigned long long) * 2 * PHIP_MAX_MVEC);
  PHIP_CHECK_HIP(hipHostMalloc((void **)&c->h_err, 64, hipHostMallocMapped | hipHostMallocCoherent));
  PHIP_CHECK_HIP(hipHostGetDevicePointer((void **)&c->h_err_dev, c->h_err, 0));
  *c->h_err = 0;
  c->seq = 0;
  PHIP_CHECK_HIP(hipMalloc((void **)&c->d_ring, sizeof(double) * PHIP_RING * PHIP_RING_VALS));
  PHIP_CHECK_HIP(hipMemsetAsync(c->d_ring, 0, sizeof(double) * PHIP_RING * PHIP_RING_VALS, c->stream));
  PHIP_CHECK_HIP(hipHostMalloc((void **)&c->h_ring, sizeof(phip_ring_slot) * PHIP_RING, hipHostMallocMapped | hipHostMallocCoherent));
  PHIP_CHECK_HIP(hipHostGetDevicePointer((void **)&c->h_ring_dev, c->h_ring, 0));
  memset(c->h_ring, 0, sizeof(phip_ring_slot) * PHIP_RING);
  c->ring_seq = 0;
  PHIP_CHECK_HIP(hipMalloc((void **)&c->d_flag, sizeof(int) * 16));
  PHIP_CHECK_HIP(hipHostMalloc((void **)&c->h_flag, sizeof(int) * 16, hipHostMallocDefault));
  PHIP_CHECK_HIP(hipMemsetAsync(c->d_flag, 0, sizeof(int) * 16, c->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(c->stream));
  { /* phip_vec_mdot_static's rows: the symbol look-up happens here, with this device current, not inside a caller's capture */
    const int rc = phip_static_scratch_setup(NULL);
    if (rc) { phip_finalize(c); return rc; } /* the message stays in phip_last_error; the half-built context does not leak */
  }
  *out = c;
  return 0;
}

extern "C" int phip_finalize(phip_ctx c)
{
  if (!c) return 0;
  if (c->device < 0) { free(c); return 0; }
  PHIP_CHECK_HIP(hipSetDevice(c->device));
  PHIP_FLUSH(c);
  PHIP_CHECK_HIP(hipStreamSynchronize(c->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(c->comm_stream));
  hipHostFree(c->h_ll); hipHostFree(c->h_err); hipFree(c->d_ring); hipHostFree(c->h_ring);
  hipFree(c->d_partials); hipFree(c->d_result); hipHostFree(c->h_result); hipFree(c->d_flag); hipHostFree(c->h_flag);
  if (c->prof) {
    for (int k = 0; k < 4; k++) for (int w = 0; w < 2; w++) for (hipEvent_t e : c->prof->ev[k][w]) hipEventDestroy(e);
    delete c->prof;
  }
  hipStreamDestroy(c->stream); hipStreamDestroy(c->comm_stream);
  free(c);
  return 0;
}

extern "C" int phip_device_count(int *count)
{
  PHIP_REQUIRE(count, PHIP_ERR_ARG_NULL, "phip_device_count: null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { n = 0; (void)hipGetLastError(); }
  *count = n;
  return 0;
}

extern "C" int phip_device_info(phip_ctx c, int *device, int *cu_count, size_t *hbm_bytes, char *name, int name_len)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_device_info: null context");
  if (c->device < 0) {
    if (device) *device = -1;
    if (cu_count) *cu_count = 0;
    if (hbm_bytes) *hbm_bytes = 0;
    if (name && name_len > 0) { strncpy(name, "planning-only", (size_t)name_len - 1); name[name_len - 1] = 0; }
    return 0;
  }
  hipDeviceProp_t prop;
  PHIP_CHECK_HIP(hipGetDeviceProperties(&prop, c->device));
  if (device) *device = c->device;
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  if (name && name_len > 0) { strncpy(name, prop.gcnArchName, (size_t)name_len - 1); name[name_len - 1] = 0; }
  return 0;
}

extern "C" int phip_malloc(phip_ctx c, size_t bytes, void **d_ptr)
{
  PHIP_REQUIRE(c && d_ptr, PHIP_ERR_ARG_NULL, "phip_malloc: null argument");
  *d_ptr = NULL;
  if (c->device < 0) { *d_ptr = PHIP_PLANNING_PTR; return 0; }
  if (!bytes) bytes = 16; /* zero-length local vectors are legal (veccusp.cu:1017-1020) */
  hipError_t e = hipMalloc(d_ptr, bytes);
  if (e != hipSuccess) { phip_set_error("phip_malloc(%zu): %s", bytes, hipGetErrorString(e)); return PHIP_ERR_MEM; }
  return 0;
}

extern "C" int phip_free(phip_ctx c, void *d_ptr)
{
  if (c && c->device < 0) return 0;
  if (c) PHIP_FLUSH(c); /* a pending operation may still name this buffer */
  if (d_ptr) PHIP_CHECK_HIP(hipFree(d_ptr));
  return 0;
}

/* A kernel that gave up a bounded wait says so in the context's pinned error word and goes on without the data it waited for: a
   reduction kernel (phip_reduce.h) stores its launch flag and publishes NaN, the peer-write ghost exchange (phip_comm.hip) stores
   0x1bc0000{1,2} and fills lvec with NaN.  Every host-visible point -- the calls that fetch a reduction result, phip_sync,
   phip_device_sync, phip_memcpy_d2h, phip_read_scalars -- reads and clears the word and fails with PHIP_ERR_LIB, so a failed wait
   cannot pass for a good result. */
int phip_check_device_error(phip_ctx c, const char *what)
{
  if (!c || !c->h_err) return 0;
  const unsigned int e = __atomic_load_n(c->h_err, __ATOMIC_ACQUIRE);
  if (!e) return 0;
  __atomic_store_n(c->h_err, 0u, __ATOMIC_RELAXED);
  if ((e & 0xfffffff0u) == 0x1bc00000u)
    phip_set_error("%s: the peer-write ghost exchange gave up waiting for a neighbour's %s; the ghost values of that product are NaN", what,
                   e == 0x1bc00001u ? "acknowledgement of the previous exchange (nothing was sent)" : "flag (its segment never arrived)");
  else
    phip_set_error("%s: a reduction kernel (launch flag %u) gave up waiting for the partial result of one of its workgroups and published NaN; the device is overloaded or a workgroup died", what, e);
  return PHIP_ERR_LIB;
}

extern "C" int phip_memcpy_h2d(phip_ctx c, void *d_dst, const void *h_src, size_t bytes)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_memcpy_h2d: null context");
  if (!bytes || c->device < 0) return 0;
  PHIP_FLUSH(c);
  PHIP_CHECK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(c->stream)); /* pageable host source may be reused by the caller at once */
  return 0;
}

extern "C" int phip_memcpy_d2h(phip_ctx c, void *h_dst, const void *d_src, size_t bytes)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_memcpy_d2h: null context");
  PHIP_NEED_DEVICE(c, "phip_memcpy_d2h");
  PHIP_FLUSH(c);
  if (bytes) PHIP_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(c->stream));
  return phip_check_device_error(c, "phip_memcpy_d2h");
}

extern "C" int phip_memcpy_d2d(phip_ctx c, void *d_dst, const void *d_src, size_t bytes)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_memcpy_d2d: null context");
  PHIP_NEED_DEVICE(c, "phip_memcpy_d2d");
  PHIP_FLUSH(c);
  if (bytes && d_dst != d_src) PHIP_CHECK_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

extern "C" int phip_sync(phip_ctx c)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_sync: null context");
  if (c->device < 0) return 0;
  PHIP_FLUSH(c);
  PHIP_CHECK_HIP(hipStreamSynchronize(c->stream));
  return phip_check_device_error(c, "phip_sync");
}

extern "C" int phip_device_sync(phip_ctx c)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_device_sync: null context");
  if (c->device < 0) return 0;
  PHIP_FLUSH(c);
  PHIP_CHECK_HIP(hipSetDevice(c->device));
  PHIP_CHECK_HIP(hipDeviceSynchronize());
  return phip_check_device_error(c, "phip_device_sync");
}

extern "C" void *phip_stream(phip_ctx c)
{
  if (c && c->ew_n) phip_ew_flush(c); /* the caller may queue work of its own behind ours */
  return c ? (void *)c->stream : NULL;
}

/* copy n device scalars into the pinned slot as flagged words: one tiny kernel instead of a copy + stream sync */
__global__ void publish_kernel(const double *__restrict__ src, int n, unsigned long long *ll, unsigned long long seq)
{
  const int t = threadIdx.x;
  if (t < n) publish_ll(ll, t, src[t], seq);
}

int phip_wait_ll(phip_ctx c, const unsigned long long *ll, int n, unsigned long long seq, double *out, const char *what)
{
  const unsigned long long flag = seq & 0xFFFFFFFFull;
  int k = 0; /* words 0 .. 2n-1 arrive in any order; wait for them one after the other */
  for (unsigned long spin = 1; k < 2 * n; spin++) {
    const unsigned long long w = __atomic_load_n(&ll[k], __ATOMIC_RELAXED);
    if ((w >> 32) == flag) { k++; continue; }
    if ((spin & 0x3FFF) == 0) { /* every ~16k polls: is the stream still healthy? */
      hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) {
        if ((__atomic_load_n(&ll[k], __ATOMIC_RELAXED) >> 32) == flag) continue;
        phip_set_error("%s: the stream drained but result %llu was never published (was the ticket opened without publish?)", what, seq);
        return PHIP_ERR_ARG_WRONGSTATE;
      }
      if (e != hipErrorNotReady) { phip_set_error("%s: %s", what, hipGetErrorString(e)); return PHIP_ERR_LIB; }
    }
    __builtin_ia32_pause();
  }
  for (int i = 0; i < n; i++) {
    const unsigned long long bits = (__atomic_load_n(&ll[2 * i], __ATOMIC_RELAXED) & 0xFFFFFFFFull) | (__atomic_load_n(&ll[2 * i + 1], __ATOMIC_RELAXED) << 32);
    memcpy(&out[i], &bits, 8);
  }
  return phip_check_device_error(c, what);
}

extern "C" int phip_read_scalars(phip_ctx c, const double *d_src, int32_t n, double *h_dst)
{
  PHIP_REQUIRE(c && h_dst, PHIP_ERR_ARG_NULL, "phip_read_scalars: null argument");
  PHIP_REQUIRE(n >= 0 && n <= PHIP_MAX_MVEC, PHIP_ERR_ARG_OUTOFRANGE, "phip_read_scalars: n=%d", n);
  PHIP_NEED_DEVICE(c, "phip_read_scalars");
  PHIP_FLUSH(c);
  if (!n) { PHIP_CHECK_HIP(hipStreamSynchronize(c->stream)); return phip_check_device_error(c, "phip_read_scalars"); }
  const unsigned long long seq = phip_next_seq(&c->seq);
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(PHIP_MAX_MVEC), 0, c->stream, d_src, n, c->h_ll_dev, seq);
  PHIP_CHECK_HIP(hipGetLastError());
  return phip_wait_ll(c, c->h_ll, n, seq, h_dst, "phip_read_scalars");
}

extern "C" int phip_event_create(phip_ctx c, void **event)
{
  PHIP_REQUIRE(c && event, PHIP_ERR_ARG_NULL, "phip_event_create: null argument");
  PHIP_NEED_DEVICE(c, "phip_event_create");
  hipEvent_t e;
  PHIP_CHECK_HIP(hipEventCreate(&e));
  *event = (void *)e;
  return 0;
}

extern "C" int phip_event_record(phip_ctx c, void *event)
{
  PHIP_REQUIRE(c && event, PHIP_ERR_ARG_NULL, "phip_event_record: null argument");
  PHIP_FLUSH(c);
  PHIP_CHECK_HIP(hipEventRecord((hipEvent_t)event, c->stream));
  return 0;
}

extern "C" int phip_event_elapsed_ms(phip_ctx c, void *start, void *stop, float *ms)
{
  PHIP_REQUIRE(c && start && stop && ms, PHIP_ERR_ARG_NULL, "phip_event_elapsed_ms: null argument");
  PHIP_CHECK_HIP(hipEventSynchronize((hipEvent_t)stop));
  PHIP_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}

extern "C" int phip_event_destroy(phip_ctx c, void *event)
{
  (void)c;
  if (event) PHIP_CHECK_HIP(hipEventDestroy((hipEvent_t)event));
  return 0;
}

/* ---- per-launch device timing of the SpMV kernel (bench.py's roofline leg) ------------------------- */

int phip_prof_open(phip_ctx c, int kind, hipEvent_t *start, hipEvent_t *stop)
{
  if (!c->prof) { c->prof = new phip_prof_s(); for (int k = 0; k < 4; k++) c->prof->used[k] = 0; }
  phip_prof_s *p = c->prof;
  if (p->used[kind] == p->ev[kind][0].size()) {
    hipEvent_t a, b;
    PHIP_CHECK_HIP(hipEventCreate(&a));
    PHIP_CHECK_HIP(hipEventCreate(&b));
    p->ev[kind][0].push_back(a); p->ev[kind][1].push_back(b);
  }
  *start = p->ev[kind][0][p->used[kind]];
  *stop = p->ev[kind][1][p->used[kind]];
  p->used[kind]++;
  return 0;
}

extern "C" int phip_profile_enable(phip_ctx c, int on)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_profile_enable: null context");
  PHIP_NEED_DEVICE(c, "phip_profile_enable");
  c->prof_on = on ? 1 : 0;
  if (c->prof) for (int k = 0; k < 4; k++) c->prof->used[k] = 0;
  return 0;
}

extern "C" int phip_profile_read(phip_ctx c, int kind, int32_t *count, double *sum_ms, double *min_ms, double *max_ms)
{
  PHIP_REQUIRE(c, PHIP_ERR_ARG_NULL, "phip_profile_read: null context");
  PHIP_REQUIRE(kind >= 0 && kind <= 3, PHIP_ERR_ARG_OUTOFRANGE, "phip_profile_read: kind %d", kind);
  PHIP_NEED_DEVICE(c, "phip_profile_read");
  PHIP_FLUSH(c);
  PHIP_CHECK_HIP(hipStreamSynchronize(c->stream));
  double  s = 0.0, lo = 0.0, hi = 0.0;
  int32_t n = 0;
  if (c->prof) {
    for (size_t k = 0; k < c->prof->used[kind]; k++) {
      float ms = 0.f;
      PHIP_CHECK_HIP(hipEventElapsedTime(&ms, c->prof->ev[kind][0][k], c->prof->ev[kind][1][k]));
      s += ms;
      if (!n || ms < lo) lo = ms;
      if (!n || ms > hi) hi = ms;
      n++;
    }
  }
  if (count) *count = n;
  if (sum_ms) *sum_ms = s;
  if (min_ms) *min_ms = lo;
  if (max_ms) *max_ms = hi;
  return 0;
}

/* the individual durations (ms) of the launches recorded since enable, in launch order: at most cap of them are written; *count receives
   the number recorded (bench.py: median and p10 / p90 of >= 200 timed launches) */
extern "C" int phip_profile_durations(phip_ctx c, int kind, int32_t cap, float *h_ms, int32_t *count)
{
  PHIP_REQUIRE(c && count && (h_ms || cap <= 0), PHIP_ERR_ARG_NULL, "phip_profile_durations: null argument");
  PHIP_REQUIRE(kind >= 0 && kind <= 3, PHIP_ERR_ARG_OUTOFRANGE, "phip_profile_durations: kind %d", kind);
  PHIP_NEED_DEVICE(c, "phip_profile_durations");
  PHIP_FLUSH(c);
  PHIP_CHECK_HIP(hipStreamSynchronize(c->stream));
  const size_t n = c->prof ? c->prof->used[kind] : 0;
  for (size_t k = 0; k < n && (int64_t)k < (int64_t)cap; k++) PHIP_CHECK_HIP(hipEventElapsedTime(&h_ms[k], c->prof->ev[kind][0][k], c->prof->ev[kind][1][k]));
  *count = (int32_t)n;
  return 0;
}

/* ---- by-product tickets -------------------------------------------------------------------------- */
int phip_ticket_open(phip_ctx c, int nvals, int publish, phip_ticket *t, RedCtl *ctl)
{
  PHIP_REQUIRE(t && ctl && nvals >= 1 && nvals < PHIP_RING_VALS, PHIP_ERR_ARG_OUTOFRANGE, "phip_ticket_open: nvals %d", nvals);
  const unsigned long long seq = phip_next_seq(&c->ring_seq);
  const int                slot = (int)(seq % PHIP_RING);
  t->slot = slot; t->nvals = nvals; t->seq = seq;
  ctl->partial = c->d_partials; ctl->pflag = phip_next_pflag(c); ctl->d_out = c->d_ring + (size_t)slot * PHIP_RING_VALS;
  ctl->kofs = 0; ctl->accumulate = 0; ctl->npub = 0; ctl->h_ll = NULL; ctl->seq = 0; ctl->err = c->h_err_dev;
  if (publish) { ctl->h_ll = c->h_ring_dev[slot].ll; ctl->seq = seq; ctl->npub = nvals; }
  return 0;
}

static int ticket_live(phip_ctx c, const phip_ticket *t)
{
  return t && t->seq && t->seq <= c->ring_seq && c->ring_seq - t->seq < PHIP_RING && t->slot == (int)(t->seq % PHIP_RING) && t->nvals >= 1 && t->nvals < PHIP_RING_VALS;
}

int phip_ticket_accumulate(phip_ctx c, const phip_ticket *t, int publish, RedCtl *ctl)
{
  PHIP_REQUIRE(ticket_live(c, t), PHIP_ERR_ARG_WRONGSTATE, "phip_ticket_accumulate: the ticket has expired");
  ctl->partial = c->d_partials; ctl->pflag = phip_next_pflag(c); ctl->d_out = c->d_ring + (size_t)t->slot * PHIP_RING_VALS;
  ctl->kofs = 0; ctl->accumulate = 1; ctl->npub = 0; ctl->h_ll = NULL; ctl->seq = 0; ctl->err = c->h_err_dev;
  if (publish) { ctl->h_ll = c->h_ring_dev[t->slot].ll; ctl->seq = t->seq; ctl->npub = t->nvals; }
  return 0;
}

extern "C" int phip_ticket_values(phip_ctx c, const phip_ticket *t, double **d_vals)
{
  PHIP_REQUIRE(c && d_vals, PHIP_ERR_ARG_NULL, "phip_ticket_values: null argument");
  PHIP_NEED_DEVICE(c, "phip_ticket_values");
  PHIP_FLUSH(c);
  PHIP_REQUIRE(ticket_live(c, t), PHIP_ERR_ARG_WRONGSTATE, "phip_ticket_values: the ticket has expired");
  *d_vals = c->d_ring + (size_t)t->slot * PHIP_RING_VALS;
  return 0;
}

extern "C" int phip_ticket_wait(phip_ctx c, const phip_ticket *t, double *h_vals)
{
  PHIP_REQUIRE(c && h_vals, PHIP_ERR_ARG_NULL, "phip_ticket_wait: null argument");
  PHIP_NEED_DEVICE(c, "phip_ticket_wait");
  PHIP_REQUIRE(ticket_live(c, t), PHIP_ERR_ARG_WRONGSTATE, "phip_ticket_wait: the ticket has expired");
  return phip_wait_ll(c, c->h_ring[t->slot].ll, t->nvals, t->seq, h_vals, "phip_ticket_wait");
}
