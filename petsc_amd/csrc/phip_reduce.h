// Device-side reduction helpers shared by the vector and SpMV kernels of libpetschip (not part of the C ABI).
#pragma once
#include "phip_internal.h"

struct RedCtl {
  double             *partial; /* [PHIP_MAX_MVEC][PHIP_RED_BLOCKS] */
  unsigned int       *ticket;  /* arrival counters [PHIP_RED_BLOCKS/32 + 1][32] (one used word per 128-byte line), zero on entry and on exit */
  double             *d_out;   /* device results, indexed like the partial rows */
  unsigned long long *h_ll;    /* pinned host slot (device address) or NULL: publish d_out[0..npub) there as flagged words ... */
  unsigned long long  seq;     /* ... whose flag is the low 32 bits of seq */
  int                 npub;
  int                 kofs;    /* first partial row / d_out index of this launch */
  int                 accumulate; /* add to d_out[] (written by an earlier kernel of the stream) instead of overwriting it */
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

/* Tail of every reduction kernel.  acc[] holds the workgroup's partials in thread 0.  MODE 0: NV sums.  MODE 1: acc[0] = max,
   acc[1] = NaN flag -> one result (NaN if any flag, bvec2.c:253-258). */
template <int NV, int MODE>
__device__ __forceinline__ void reduce_tail(double (&acc)[NV], double *lds /* >= 4 doubles */, const RedCtl c)
{
  __shared__ int s_last;
  const int      t = threadIdx.x;
  const int      nb = gridDim.x;
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) __hip_atomic_store(&c.partial[(size_t)(c.kofs + k) * PHIP_RED_BLOCKS + blockIdx.x], acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    /* two-level arrival count: 32 workgroups per group counter (each on a 128-byte line of its own), then one top counter, so that
       no word takes more than 32 serialised atomics (~12 ns each on one address); the last arrivers put the counters back to zero */
    const int          g = blockIdx.x >> 5;
    const unsigned int gsize = (unsigned int)((nb - (g << 5)) < 32 ? (nb - (g << 5)) : 32);
    int                last = 0;
    const unsigned int tk = __hip_atomic_fetch_add(&c.ticket[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == gsize - 1) {
      __hip_atomic_store(&c.ticket[g * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int ngroups = (unsigned int)((nb + 31) >> 5);
      const unsigned int tk2 = __hip_atomic_fetch_add(&c.ticket[(PHIP_RED_BLOCKS / 32) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tk2 == ngroups - 1) {
        __hip_atomic_store(&c.ticket[(PHIP_RED_BLOCKS / 32) * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  /* last workgroup: wave w sums partial rows w, w+4, ...: lane l adds the values l, l+64, l+128, ... in that order (16 loads in flight
     at a time), then the 64-lane tree -- a fixed order for a given grid, whichever workgroup arrives last */
  const int w = t >> 6, lane = t & 63;
  if (MODE == 1) {
    if (w == 0) {
      double m = 0.0, f = 0.0;
      for (int j0 = 0; j0 * 64 < nb; j0 += 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int i = lane + 64 * (j0 + j);
          if (i < nb) {
            m = fmax(m, __hip_atomic_load(&c.partial[(size_t)c.kofs * PHIP_RED_BLOCKS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            f = fmax(f, __hip_atomic_load(&c.partial[(size_t)(c.kofs + 1) * PHIP_RED_BLOCKS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          }
        }
      }
      m = wave_max(m); f = wave_max(f);
      if (lane == 0) {
        const double r = (f > 0.0) ? __builtin_nan("") : m;
        c.d_out[c.kofs] = r;
        if (c.h_ll && c.kofs < c.npub) publish_ll(c.h_ll, c.kofs, r, c.seq);
      }
    }
  } else {
    for (int k = w; k < NV; k += PHIP_THREADS / 64) {
      const double *row = c.partial + (size_t)(c.kofs + k) * PHIP_RED_BLOCKS;
      double        s = 0.0;
      for (int j0 = 0; j0 * 64 < nb; j0 += 16) {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int i = lane + 64 * (j0 + j);
          v[j] = (i < nb) ? __hip_atomic_load(&row[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) s += v[j];
      }
      s = wave_sum(s);
      if (lane == 0) { /* the result goes to the host straight from the register: no round trip through d_out */
        const double r = c.accumulate ? c.d_out[c.kofs + k] + s : s;
        c.d_out[c.kofs + k] = r;
        if (c.h_ll && c.kofs + k < c.npub) publish_ll(c.h_ll, c.kofs + k, r, c.seq);
      }
    }
  }
  /* results of earlier launches of the same call (the first groups of a long VecMDot) ride along: written by earlier kernels of the stream */
  if (c.h_ll && t < c.kofs && t < c.npub) publish_ll(c.h_ll, t, c.d_out[t], c.seq);
}
