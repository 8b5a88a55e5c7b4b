// Micro-benchmark: what does a READ-ONLY stream out of HBM reach on this part, in the shapes the SpMV kernels read their matrices in?
//   shape 0: grid-stride, 16 bytes a lane, consecutive workgroups read consecutive 4 KiB (the vector kernels' shape)
//   shape 1: every workgroup owns one contiguous chunk (CHUNK bytes) and walks it in 4 KiB steps, U steps requested at a time
//            (the SELL-dict kernel's shape: a slice of 512 rows x L slots is one chunk of 4 KiB x L)
// over 1.25 GiB and 4 GiB (the 7-point and 27-point 256^3 matrices), non-temporal loads, nothing written but one double a lane.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/read_stream.hip -o scripts/micro/read_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ldnt(const double2 *p)
{
  const f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p));
  return make_double2(v.x, v.y);
}
#define CHECK(c) do { hipError_t e = (c); if (e != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void grid_stride_kernel(const double2 *__restrict__ a, size_t n16, double *__restrict__ out)
{
  const size_t stride = (size_t)gridDim.x * 256;
  double       s = 0.0;
  size_t       i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const double2 p = ldnt(a + i), q = ldnt(a + i + stride);
    const double2 r = ldnt(a + i + 2 * stride), t = ldnt(a + i + 3 * stride);
    s += p.x + p.y + q.x + q.y + r.x + r.y + t.x + t.y;
  }
  for (; i < n16; i += stride) { const double2 p = ldnt(a + i); s += p.x + p.y; }
  if (s == 12345.678) out[0] = s;
}

template <int U>
__global__ __launch_bounds__(256) void chunk_kernel(const double2 *__restrict__ a, int steps, double *__restrict__ out)
{
  const double2 *p = a + (size_t)blockIdx.x * steps * 256 + threadIdx.x;
  double         s = 0.0;
  int            k = 0;
  for (; k + U <= steps; k += U) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ldnt(p + (size_t)(k + u) * 256);
#pragma unroll
    for (int u = 0; u < U; u++) s += v[u].x + v[u].y;
  }
  for (; k < steps; k++) { const double2 v = ldnt(p + (size_t)k * 256); s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}

/* the SELL-dict kernel's two streams: per step 4 KiB of values (16 bytes a lane) and 512 bytes of codes (2 bytes a lane), either as two
   arrays (MERGED = false: values at a, codes at c) or as one chunk of 4.5 KiB a step (MERGED = true) */
template <int U, bool MERGED, int LDS_KB = 0>
__global__ __launch_bounds__(256) void two_stream_kernel(const double2 *__restrict__ a, const unsigned short *__restrict__ c, int steps, double *__restrict__ out)
{
  /* LDS_KB > 0: that much LDS a workgroup, touched once, only to hold the occupancy down (160 KiB a CU: 32 -> 5, 48 -> 3 workgroups) */
  __shared__ double hold[LDS_KB > 0 ? LDS_KB * 128 : 1];
  if (LDS_KB > 0) { hold[threadIdx.x] = (double)steps; __syncthreads(); if (hold[(threadIdx.x + 1) & 255] < 0.0) out[1] = 1.0; }
  const size_t vstep = MERGED ? 288 : 256;   /* double2 units between the values of consecutive steps */
  const size_t cstep = MERGED ? 2304 : 256;  /* ushort units between the codes of consecutive steps */
  const double2        *p = a + (size_t)blockIdx.x * steps * vstep + threadIdx.x;
  const unsigned short *q = MERGED ? reinterpret_cast<const unsigned short *>(a + (size_t)blockIdx.x * steps * vstep + 256) + threadIdx.x
                                   : c + (size_t)blockIdx.x * steps * cstep + threadIdx.x;
  double s = 0.0;
  int    k = 0;
  for (; k + U <= steps; k += U) {
    double2        v[U];
    unsigned short w[U];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = ldnt(p + (size_t)(k + u) * vstep); w[u] = __builtin_nontemporal_load(q + (size_t)(k + u) * cstep); }
#pragma unroll
    for (int u = 0; u < U; u++) s += v[u].x + v[u].y + (double)w[u];
  }
  for (; k < steps; k++) { const double2 v = ldnt(p + (size_t)k * vstep); s += v.x + v.y + (double)__builtin_nontemporal_load(q + (size_t)k * cstep); }
  if (s == 12345.678) out[0] = s;
}

/* the chunk stream plus, per 4 KiB step, EXTRA_KB KiB of plain loads out of a 1 MiB table that stays in every L2 (every step another piece of
   it, so the vector L1 misses): does a line that L2 serves cost the stream as much as a line from HBM? */
template <int U, int EXTRA_KB>
__global__ __launch_bounds__(256) void stream_plus_l2_kernel(const double2 *__restrict__ a, const double2 *__restrict__ table, int steps, double *__restrict__ out)
{
  const double2 *p = a + (size_t)blockIdx.x * steps * 256 + threadIdx.x;
  double         s = 0.0;
  int            k = 0;
  for (; k + U <= steps; k += U) {
    double2 v[U], e[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ldnt(p + (size_t)(k + u) * 256);
#pragma unroll
    for (int u = 0; u < U; u++) {
      /* piece (workgroup * steps + step) of the table, EXTRA_KB KiB = EXTRA_KB * 64 lanes of 16 bytes; 1 MiB = 1024 pieces of 1 KiB */
      const unsigned piece = ((unsigned)blockIdx.x * (unsigned)steps + (unsigned)(k + u)) * EXTRA_KB % (1024u - EXTRA_KB);
      e[u] = make_double2(0.0, 0.0);
      if (threadIdx.x < EXTRA_KB * 64) e[u] = table[(size_t)piece * 64 + threadIdx.x];
    }
#pragma unroll
    for (int u = 0; u < U; u++) s += v[u].x + v[u].y + e[u].x + e[u].y;
  }
  for (; k < steps; k++) { const double2 v = ldnt(p + (size_t)k * 256); s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}

/* the chunk stream plus, per 4 KiB step, G loads of 8 bytes a lane that HIT in L1 (a window of 8 KiB of the table per workgroup, read
   over and over): the SELL-dict kernel's mix of requests -- one 16-byte stream load and two 8-byte gathers per step and wave -- without
   its arithmetic.  Does the address path's time for requests that hit slow the stream down? */
template <int U, int G>
__global__ __launch_bounds__(256) void stream_plus_l1_kernel(const double2 *__restrict__ a, const double *__restrict__ table, int steps, double *__restrict__ out)
{
  const double2 *p = a + (size_t)blockIdx.x * steps * 256 + threadIdx.x;
  const double  *w = table + (size_t)(blockIdx.x & 127) * 1024 + (threadIdx.x & 63); /* 8 KiB window of this workgroup (1 MiB table) */
  double         s = 0.0;
  int            k = 0;
  for (; k + U <= steps; k += U) {
    double2 v[U];
    double  e[U][G > 0 ? G : 1];
    asm volatile("" : "+v"(w)); /* the window loads stay in the loop */
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ldnt(p + (size_t)(k + u) * 256);
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int g = 0; g < G; g++) e[u][g] = w[(((k + u) * G + g) & 7) * 64 + (threadIdx.x >> 6) * 128]; /* within the 1024 doubles of the window */
#pragma unroll
    for (int u = 0; u < U; u++) {
      s += v[u].x + v[u].y;
#pragma unroll
      for (int g = 0; g < G; g++) s += e[u][g];
    }
  }
  for (; k < steps; k++) { const double2 v = ldnt(p + (size_t)k * 256); s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}

int main()
{
  const size_t cap = (size_t)4 << 30;
  double      *d_a, *d_out;
  CHECK(hipMalloc((void **)&d_a, cap));
  CHECK(hipMalloc((void **)&d_out, 4096));
  CHECK(hipMemset(d_a, 0, cap));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
#define TIME(LABEL, BYTES, LAUNCH)                                                                                      \
  {                                                                                                                     \
    LAUNCH; LAUNCH;                                                                                                     \
    CHECK(hipDeviceSynchronize());                                                                                      \
    CHECK(hipEventRecord(e0, 0));                                                                                       \
    for (int r = 0; r < 10; r++) { LAUNCH; }                                                                            \
    CHECK(hipEventRecord(e1, 0));                                                                                       \
    CHECK(hipEventSynchronize(e1));                                                                                     \
    float ms;                                                                                                           \
    CHECK(hipEventElapsedTime(&ms, e0, e1));                                                                            \
    printf("{\"shape\": \"%s\", \"bytes\": %zu, \"us\": %.1f, \"GBps\": %.0f}\n", LABEL, (size_t)(BYTES), ms * 100.0,   \
           (double)(BYTES) / (ms / 10.0 * 1e-3) / 1e9);                                                                 \
    fflush(stdout);                                                                                                     \
  }
  for (size_t bytes : {(size_t)1280 << 20, (size_t)4 << 30}) {
    const size_t n16 = bytes / 16;
    for (int grid : {2048, 8192, 65536})
      TIME(grid == 2048 ? "grid-stride 2048 wg" : grid == 8192 ? "grid-stride 8192 wg" : "grid-stride 65536 wg", bytes,
           hipLaunchKernelGGL(grid_stride_kernel, dim3(grid), dim3(256), 0, 0, (const double2 *)d_a, n16, d_out));
    for (int steps : {7, 27}) {
      const unsigned nwg = (unsigned)(bytes / ((size_t)steps * 4096));
      const size_t   moved = (size_t)nwg * steps * 4096;
      char           label[96];
      snprintf(label, sizeof label, "chunks of %d x 4 KiB, 4 at a time", steps);
      TIME(label, moved, hipLaunchKernelGGL(chunk_kernel<4>, dim3(nwg), dim3(256), 0, 0, (const double2 *)d_a, steps, d_out));
      snprintf(label, sizeof label, "chunks of %d x 4 KiB, 8 at a time", steps);
      TIME(label, moved, hipLaunchKernelGGL(chunk_kernel<8>, dim3(nwg), dim3(256), 0, 0, (const double2 *)d_a, steps, d_out));
      /* the table: the last 1 MiB of the buffer (the stream stays in front of it) */
      {
        const unsigned  nwg3 = (unsigned)((bytes - ((size_t)1 << 20)) / ((size_t)steps * 4096));
        const size_t    moved3 = (size_t)nwg3 * steps * 4096;
        const double2  *table = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(d_a) + bytes - ((size_t)1 << 20));
        snprintf(label, sizeof label, "chunks of %d x 4 KiB, 8 at a time + 1 KiB a step out of L2 (rate on the stream bytes)", steps);
        TIME(label, moved3, hipLaunchKernelGGL((stream_plus_l2_kernel<8, 1>), dim3(nwg3), dim3(256), 0, 0, (const double2 *)d_a, table, steps, d_out));
        snprintf(label, sizeof label, "chunks of %d x 4 KiB, 8 at a time + 2 KiB a step out of L2 (rate on the stream bytes)", steps);
        TIME(label, moved3, hipLaunchKernelGGL((stream_plus_l2_kernel<8, 2>), dim3(nwg3), dim3(256), 0, 0, (const double2 *)d_a, table, steps, d_out));
      }
      {
        const unsigned nwg3 = (unsigned)((bytes - ((size_t)1 << 20)) / ((size_t)steps * 4096));
        const size_t   moved3 = (size_t)nwg3 * steps * 4096;
        const double  *table = reinterpret_cast<const double *>(reinterpret_cast<const char *>(d_a) + bytes - ((size_t)1 << 20));
        snprintf(label, sizeof label, "chunks of %d x 4 KiB, 8 at a time + 2 L1-hitting 8-byte loads a step and wave", steps);
        TIME(label, moved3, hipLaunchKernelGGL((stream_plus_l1_kernel<8, 2>), dim3(nwg3), dim3(256), 0, 0, (const double2 *)d_a, table, steps, d_out));
        snprintf(label, sizeof label, "chunks of %d x 4 KiB, 8 at a time + 4 L1-hitting 8-byte loads a step and wave", steps);
        TIME(label, moved3, hipLaunchKernelGGL((stream_plus_l1_kernel<8, 4>), dim3(nwg3), dim3(256), 0, 0, (const double2 *)d_a, table, steps, d_out));
      }
      /* values + codes: 4608 bytes a step; the two-array form keeps the codes in the last ninth of the buffer */
      const unsigned nwg2 = (unsigned)(bytes / ((size_t)steps * 4608));
      const size_t   moved2 = (size_t)nwg2 * steps * 4608;
      const unsigned short *codes = reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(d_a) + (size_t)nwg2 * steps * 4096);
      snprintf(label, sizeof label, "values + codes in two arrays, %d steps, 8 at a time", steps);
      TIME(label, moved2, hipLaunchKernelGGL((two_stream_kernel<8, false>), dim3(nwg2), dim3(256), 0, 0, (const double2 *)d_a, codes, steps, d_out));
      snprintf(label, sizeof label, "values + codes in two arrays, %d steps, 8 at a time, 5 workgroups a CU", steps);
      TIME(label, moved2, hipLaunchKernelGGL((two_stream_kernel<8, false, 32>), dim3(nwg2), dim3(256), 0, 0, (const double2 *)d_a, codes, steps, d_out));
      snprintf(label, sizeof label, "values + codes in two arrays, %d steps, 8 at a time, 3 workgroups a CU", steps);
      TIME(label, moved2, hipLaunchKernelGGL((two_stream_kernel<8, false, 48>), dim3(nwg2), dim3(256), 0, 0, (const double2 *)d_a, codes, steps, d_out));
      snprintf(label, sizeof label, "values + codes in one 4.5 KiB chunk a step, %d steps, 8 at a time", steps);
      TIME(label, moved2, hipLaunchKernelGGL((two_stream_kernel<8, true>), dim3(nwg2), dim3(256), 0, 0, (const double2 *)d_a, codes, steps, d_out));
    }
  }
  return 0;
}
