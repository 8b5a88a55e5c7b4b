// Micro-test: global_load_lds_dwordx4 (direct-to-LDS loads) on gfx950 -- where does lane t's 16 bytes land, also under a partial
// execution mask, and does a FIFO of such loads keep a read-only stream at the rate of plain loads?
//   check: every wave stages 8 slots of 1 KiB (lane t: 16 bytes at LDS base + 16 t); what the wave then reads back from LDS must be
//          the source.  (A first form of this check skipped some lanes of the odd slots: the compiler merged the tails of the divergent
//          branches, made the LDS base -- which the instruction takes from M0, one value per wave -- differ between lanes and read it
//          from the first lane: slot 5's data in slot 2's place.  The builtin is only safe where control flow is uniform.)
//   rate:  chunk stream (as scripts/micro/read_stream.hip) with the next 8 steps staged in LDS while the current 8 are summed.
// Build: hipcc --offload-arch=gfx950 -O3 -DLDS_AUX=2 scripts/micro/lds_dma.hip -o scripts/micro/lds_dma   (LDS_AUX: cache policy bits of the loads, 2 = nt)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CHECK(c) do { hipError_t e = (c); if (e != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void       *lptr_t;

__global__ __launch_bounds__(256) void check_kernel(const double2 *__restrict__ a, double2 *__restrict__ out)
{
  __shared__ double2 stage[4][8][64];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int s = 0; s < 8; s++) stage[w][s][l] = make_double2(-1.0, -2.0); /* marker */
  __syncthreads();
  const double2 *p = a + (size_t)blockIdx.x * 8 * 256 + threadIdx.x;
#pragma unroll
  for (int s = 0; s < 8; s++)
    __builtin_amdgcn_global_load_lds((gptr_t)(p + s * 256), (lptr_t)&stage[w][s][0], 16, 0, 0);
  /* the compiler waits for the loads before the first read of the staged array */
#pragma unroll
  for (int s = 0; s < 8; s++) out[((size_t)blockIdx.x * 8 + s) * 256 + threadIdx.x] = stage[w][s][l];
}

__global__ __launch_bounds__(256) void stream_kernel(const double2 *__restrict__ a, int steps, double *__restrict__ out)
{
  __shared__ double2 stage[4][8][64];
  const int      w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const double2 *p = a + (size_t)blockIdx.x * steps * 256 + threadIdx.x;
  double         s = 0.0;
  {
#pragma unroll
    for (int u = 0; u < 8; u++) if (u < steps) __builtin_amdgcn_global_load_lds((gptr_t)(p + (size_t)u * 256), (lptr_t)&stage[w][u][0], 16, 0, LDS_AUX);
    const unsigned la = (unsigned)(size_t)(lptr_t)&stage[w][0][l]; /* this lane's 16 bytes of slot 0; slot u is 1 KiB further */
    for (int k = 0; k < steps; k += 8) {
      /* the staged batch comes out of LDS in one block the compiler cannot take apart or move loads into: it models a load into LDS
         as a 16-byte store at the wave's base, so it feels free to reorder the lanes' reads around it, and it waits for EVERY
         outstanding load into LDS before any LDS read it can see */
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 r[8];
      asm volatile("s_waitcnt vmcnt(0)\n\t"
                   "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                   "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                   : "v"(la)
                   : "memory");
      double2 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = __builtin_bit_cast(double2, r[u]);
#pragma unroll
      for (int u = 0; u < 8; u++) if (k + 8 + u < steps) __builtin_amdgcn_global_load_lds((gptr_t)(p + (size_t)(k + 8 + u) * 256), (lptr_t)&stage[w][u][0], 16, 0, LDS_AUX);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; u++) if (k + u < steps) s += v[u].x + v[u].y;
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
  /* ---- check ---- */
  const int nwg = 64;
  std::vector<double> h((size_t)nwg * 8 * 256 * 2);
  for (size_t i = 0; i < h.size(); i++) h[i] = (double)i + 0.25;
  double *d_a, *d_o;
  CHECK(hipMalloc((void **)&d_a, h.size() * 8));
  CHECK(hipMalloc((void **)&d_o, h.size() * 8));
  CHECK(hipMemcpy(d_a, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(check_kernel, dim3(nwg), dim3(256), 0, 0, (const double2 *)d_a, (double2 *)d_o);
  CHECK(hipDeviceSynchronize());
  std::vector<double> o(h.size());
  CHECK(hipMemcpy(o.data(), d_o, h.size() * 8, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int b = 0; b < nwg; b++)
    for (int s = 0; s < 8; s++)
      for (int t = 0; t < 256; t++) {
        const size_t i = (((size_t)b * 8 + s) * 256 + t) * 2;
        const bool   skipped = false;
        const double e0 = skipped ? -1.0 : h[i], e1 = skipped ? -2.0 : h[i + 1];
        if (o[i] != e0 || o[i + 1] != e1) { if (bad < 24 || (b == 0 && s < 2 && t < 8)) printf("  b %d s %d t %d: got %.2f %.2f want %.2f %.2f\n", b, s, t, o[i], o[i + 1], e0, e1); bad++; }
      }
  printf("{\"test\": \"lane t of a wave lands at LDS base + 16 t, masked lanes leave their slot alone\", \"mismatches\": %zu}\n", bad);
  if (bad) return 2;
  /* the staged stream over the same data: 18 workgroups x 27 steps and 73 x 7 steps, every thread's sum against the host's */
  for (int steps : {27, 7}) {
    const int nw = nwg * 8 / steps;
    hipLaunchKernelGGL(stream_kernel, dim3(nw), dim3(256), 0, 0, (const double2 *)d_a, steps, d_o);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(o.data(), d_o, (size_t)nw * 256 * 8, hipMemcpyDeviceToHost));
    size_t bad2 = 0;
    for (int b = 0; b < nw; b++)
      for (int t = 0; t < 256; t++) {
        double e = 0.0;
        for (int k = 0; k < steps; k++) { const size_t i = (((size_t)b * steps + k) * 256 + t) * 2; e += h[i] + h[i + 1]; }
        if (o[(size_t)b * 256 + t] != e) bad2++;
      }
    printf("{\"test\": \"staged stream, %d steps: every thread's sum\", \"mismatches\": %zu}\n", steps, bad2);
    if (bad2) return 3;
  }
  CHECK(hipFree(d_a)); CHECK(hipFree(d_o));
  /* ---- rate ---- */
  const size_t bytes = (size_t)2 << 30;
  double      *d_b, *d_s;
  CHECK(hipMalloc((void **)&d_b, bytes));
  CHECK(hipMemset(d_b, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int steps : {27, 7}) {
    const unsigned n = (unsigned)(bytes / ((size_t)steps * 4096));
    CHECK(hipMalloc((void **)&d_s, (size_t)n * 256 * 8));
    {
      for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(stream_kernel, dim3(n), dim3(256), 0, 0, (const double2 *)d_b, steps, d_s);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("{\"shape\": \"chunks of %d x 4 KiB staged through LDS, 8 steps ahead\", \"GBps\": %.0f}\n", steps, (double)n * steps * 4096 / (ms / 5 * 1e-3) / 1e9);
      }
    }
    CHECK(hipFree(d_s));
  }
  return 0;
}
