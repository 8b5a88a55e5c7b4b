// Device-side reduction helpers shared by the vector and SpMV kernels of libpetschip (not part of the C ABI).
#pragma once
#include "phip_internal.h"

struct RedCtl {
  unsigned long long *partial; /* [PHIP_MAX_MVEC][PHIP_RED_BLOCKS][2] flagged words */
  unsigned int        pflag;   /* flag of this launch's partials: unique per launch, never 0 */
  double             *d_out;   /* device results, indexed like the partial rows */
  unsigned long long *h_ll;    /* pinned host slot (device address) or NULL: publish d_out[0..npub) there as flagged words ... */
  unsigned long long  seq;     /* ... whose flag is the low 32 bits of seq */
  int                 npub;
  int                 kofs;    /* first partial row / d_out index of this launch */
  int                 accumulate; /* add to d_out[] (written by an earlier kernel of the stream) instead of overwriting it */
  unsigned int       *err = nullptr; /* the context's pinned error word (device address) or NULL: a collector that gave up waiting leaves its flag there */
};

/* value k of a result slot, as two flagged 8-byte words (phip_internal.h) */
__device__ __forceinline__ void publish_ll(unsigned long long *ll, int k, double v, unsigned long long seq)
{
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v), flag = (seq & 0xFFFFFFFFull) << 32;
  __hip_atomic_store(&ll[2 * k], flag | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&ll[2 * k + 1], flag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

/* block-level sum of NV values; result valid in thread 0 */
template <int NV>
__device__ __forceinline__ void block_sum(double (&acc)[NV], double *lds /* [4*NV] */)
{
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NV; k++) acc[k] = wave_sum(acc[k]);
  if ((t & 63) == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) lds[(t >> 6) * NV + k] = acc[k];
  }
  __syncthreads();
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = (lds[k] + lds[NV + k]) + (lds[2 * NV + k] + lds[3 * NV + k]);
  }
}

/* one flagged value (two 8-byte words, value half + flag) in device memory, visible to every XCD */
__device__ __forceinline__ void partial_put(unsigned long long *p, double v, unsigned int flag)
{
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v), f = (unsigned long long)flag << 32;
  __hip_atomic_store(&p[0], f | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&p[1], f | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* B flagged values p[2*stride*j], j < B, all requested at once; the batch is requested again until every flag matches (entries i >= n
   are skipped and read as 0).  *ok goes to 0 if they do not come (a workgroup that died): the caller then publishes NaN. */
template <int B>
__device__ __forceinline__ void partial_get_batch(const unsigned long long *p, int stride, int i0, int n, unsigned int flag, double (&v)[B], int *ok)
{
  unsigned long long w0[B], w1[B];
  for (int spin = 0;; spin++) {
#pragma unroll
    for (int j = 0; j < B; j++) {
      const int i = i0 + stride * j;
      if (i < n) {
        w0[j] = __hip_atomic_load(&p[2 * (size_t)i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w1[j] = __hip_atomic_load(&p[2 * (size_t)i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    bool all = true;
#pragma unroll
    for (int j = 0; j < B; j++)
      if (i0 + stride * j < n) all = all && (unsigned int)(w0[j] >> 32) == flag && (unsigned int)(w1[j] >> 32) == flag;
    if (all) break;
    if (spin > (1 << 22)) { *ok = 0; break; } /* seconds: never in a healthy launch, and the kernel still ends */
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int j = 0; j < B; j++) v[j] = (i0 + stride * j < n) ? __longlong_as_double((long long)((w0[j] & 0xFFFFFFFFull) | (w1[j] << 32))) : 0.0;
}

/* the collector gave up (partial_get_batch): besides publishing NaN it leaves the launch's flag in the context's error word, ahead of
   the results, so that the host call that fetches them fails with PHIP_ERR_LIB instead of handing NaN to the solver */
__device__ __forceinline__ void reduce_report_timeout(const RedCtl &c)
{
  if (!c.err) return;
  __hip_atomic_store(c.err, c.pflag ? c.pflag : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __atomic_thread_fence(__ATOMIC_RELEASE);
}

/* Tail of every reduction kernel.  acc[] holds the workgroup's partials in thread 0.  MODE 0: NV sums.  MODE 1: acc[0] = max,
   acc[1] = NaN flag -> one result (NaN if any flag, bvec2.c:253-258).
   Every workgroup leaves its partials as flagged words (the flag is unique to the launch) and is done.  The workgroup with the highest
   index collects them: its lanes wait for the flags of the entries they sum -- no arrival counter, no store-acknowledge wait, no
   atomics; the chain after the slowest workgroup's store is one load, the sum and the publication.  Workgroups are dispatched in
   index order (per XCD), so every workgroup the collector waits for is running or finished when it starts to wait. */
template <int NV, int MODE>
__device__ __forceinline__ void reduce_tail(double (&acc)[NV], double *lds /* unused */, const RedCtl c)
{
  const int t = threadIdx.x;
  const int nb = gridDim.x;
  (void)lds;
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) partial_put(c.partial + 2 * ((size_t)(c.kofs + k) * PHIP_RED_BLOCKS + blockIdx.x), acc[k], c.pflag);
  }
  if ((int)blockIdx.x != nb - 1) return;
  /* collector: wave w sums partial rows w, w+4, ...: lane l adds the values l, l+64, l+128, ... in that order (8 values requested at a
     time), then the 64-lane tree -- a fixed order for a given grid */
  const int w = t >> 6, lane = t & 63;
  int       ok = 1;
  if (MODE == 1) {
    if (w == 0) {
      double m = 0.0, f = 0.0;
      for (int j0 = 0; j0 * 64 < nb; j0 += 8) {
        double vm[8], vf[8];
        partial_get_batch<8>(c.partial + 2 * ((size_t)c.kofs * PHIP_RED_BLOCKS), 64, lane + 64 * j0, nb, c.pflag, vm, &ok);
        partial_get_batch<8>(c.partial + 2 * ((size_t)(c.kofs + 1) * PHIP_RED_BLOCKS), 64, lane + 64 * j0, nb, c.pflag, vf, &ok);
#pragma unroll
        for (int j = 0; j < 8; j++) { m = fmax(m, vm[j]); f = fmax(f, vf[j]); }
      }
      m = wave_max(m); f = wave_max(f);
      ok = __all(ok);
      if (lane == 0) {
        if (!ok) reduce_report_timeout(c);
        const double r = (f > 0.0 || !ok) ? __builtin_nan("") : m;
        c.d_out[c.kofs] = r;
        if (c.h_ll && c.kofs < c.npub) publish_ll(c.h_ll, c.kofs, r, c.seq);
      }
    }
  } else {
    for (int k = w; k < NV; k += PHIP_THREADS / 64) {
      const unsigned long long *row = c.partial + 2 * ((size_t)(c.kofs + k) * PHIP_RED_BLOCKS);
      double                    s = 0.0;
      for (int j0 = 0; j0 * 64 < nb; j0 += 8) {
        double v[8];
        partial_get_batch<8>(row, 64, lane + 64 * j0, nb, c.pflag, v, &ok);
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[j];
      }
      s = wave_sum(s);
      ok = __all(ok);
      if (lane == 0) { /* the result goes to the host straight from the register: no round trip through d_out */
        double r = c.accumulate ? c.d_out[c.kofs + k] + s : s;
        if (!ok) { r = __builtin_nan(""); reduce_report_timeout(c); }
        c.d_out[c.kofs + k] = r;
        if (c.h_ll && c.kofs + k < c.npub) publish_ll(c.h_ll, c.kofs + k, r, c.seq);
      }
    }
  }
  /* results of earlier launches of the same call (the first groups of a long VecMDot) ride along: written by earlier kernels of the stream */
  if (c.h_ll && t < c.kofs && t < c.npub) publish_ll(c.h_ll, t, c.d_out[t], c.seq);
}

/* Tail for launches that must leave nothing behind in their partial rows (phip_vec_mdot_static: the rows are a file-scope __device__
   array that every call on the GPU shares, and a captured launch may be replayed with the same arguments).  No flags: thread 0 of every
   workgroup stores its partials write-through, then adds 1 to ONE arrival counter with release/acquire at agent scope; the workgroup
   whose add returns nb - 1 is the last, sums the rows in the order of reduce_tail's collector (wave w rows w, w+4, ...; lane l the
   values l, l+64, ...; then the 64-lane tree) -- the same bits -- and puts the counter back to 0 for the next launch.  No workgroup
   waits for another one.  (MI355X_MICROARCH.md, hand-offs: one lane per workgroup stores sc1 and adds to one unsharded counter, the
   last arriver loads sc1 after its add returned, the other waves after a workgroup barrier.) */
template <int NV>
__device__ __forceinline__ void reduce_tail_counted(double (&acc)[NV], unsigned long long *rows /* [NV][PHIP_RED_BLOCKS][2] */, unsigned int *counter, double *d_out)
{
  __shared__ int last;
  const int      t = threadIdx.x, nb = gridDim.x;
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++)
      __hip_atomic_store(&rows[2 * ((size_t)k * PHIP_RED_BLOCKS + blockIdx.x)], (unsigned long long)__double_as_longlong(acc[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)(nb - 1);
  }
  __syncthreads();
  if (!last) return;
  const int w = t >> 6, lane = t & 63;
  for (int k = w; k < NV; k += PHIP_THREADS / 64) {
    const unsigned long long *row = rows + 2 * ((size_t)k * PHIP_RED_BLOCKS);
    double                    s = 0.0;
    for (int j0 = 0; j0 * 64 < nb; j0 += 8) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int i = lane + 64 * (j0 + j);
        v[j] = i < nb ? __longlong_as_double((long long)__hip_atomic_load(&row[2 * (size_t)i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 8; j++) s += v[j];
    }
    s = wave_sum(s);
    if (lane == 0) d_out[k] = s;
  }
  if (t == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
