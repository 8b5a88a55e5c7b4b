// Micro-benchmark: can the hottest entries of a gathered table be kept in the vector L1 by letting all OTHER gathers bypass it?
// Indices: a share `hot` of the gathers falls into the first T entries (16 KiB), the rest uniformly into a 2 MiB table (L2-resident).
// Variant 0: one plain load per gather.  Variants 1..3: two loads per gather, branch-free -- a plain (L1-allocating) load for the lanes
// whose index is below T (the other lanes read entry 0) and a load with cache policy sc1 / nt / sc0 sc1 for the rest (the hot lanes read
// entry T).  Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_l1protect.hip -o scripts/micro/gather_l1protect
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(c) do { hipError_t e = (c); if (e != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e)); return 1; } } while (0)

template <int AUX>
__device__ __forceinline__ double ld(__amdgpu_buffer_rsrc_t r, int32_t i)
{
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, i * 8, 0, AUX);
  return __builtin_bit_cast(double, w);
}

template <int AUX>
__global__ __launch_bounds__(256) void gather_kernel(int64_t n, const int32_t *__restrict__ idx, const double *x, uint32_t xbytes, int32_t T, double *__restrict__ out)
{
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)xbytes, 0x00020000);
  const int64_t stride = (int64_t)gridDim.x * 256;
  double        s = 0.0;
  int64_t       i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    int32_t q[4] = {idx[i], idx[i + stride], idx[i + 2 * stride], idx[i + 3 * stride]};
    double  v[4];
    if (AUX == 0) {
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = ld<0>(r, q[u]);
    } else {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { a[u] = ld<0>(r, q[u] < T ? q[u] : 0); b[u] = ld<AUX>(r, q[u] < T ? T : q[u]); }
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = q[u] < T ? a[u] : b[u];
    }
    s += v[0]; s += v[1]; s += v[2]; s += v[3];
  }
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
  const int64_t n = 1 << 25;
  std::vector<int32_t> h((size_t)n);
  int32_t *d_idx; double *d_x, *d_out;
  const int grid = 256 * 16;
  const int64_t m = 1 << 18; /* 2 MiB */
  const int32_t T = 2048;    /* 16 KiB */
  CHECK(hipMalloc((void **)&d_idx, sizeof(int32_t) * n));
  CHECK(hipMalloc((void **)&d_x, sizeof(double) * m));
  CHECK(hipMalloc((void **)&d_out, sizeof(double) * grid * 256));
  CHECK(hipMemset(d_x, 0, sizeof(double) * m));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  uint64_t rs = 88172645463325252ull;
  for (int hotpct : {0, 20, 40, 60}) {
    for (int64_t i = 0; i < n; i++) {
      rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
      const bool hot = (int)((rs >> 40) % 100) < hotpct;
      h[(size_t)i] = (int32_t)(hot ? (rs % (uint64_t)T) : (rs % (uint64_t)m));
    }
    CHECK(hipMemcpy(d_idx, h.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
#define RUN(AUX)                                                                                                                      \
  {                                                                                                                                   \
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(gather_kernel<AUX>, dim3(grid), dim3(256), 0, 0, n, d_idx, d_x, (uint32_t)(m * 8), T, d_out); \
    CHECK(hipDeviceSynchronize());                                                                                                    \
    float best = 1e30f;                                                                                                               \
    for (int rep = 0; rep < 3; rep++) {                                                                                               \
      CHECK(hipEventRecord(e0, 0));                                                                                                   \
      for (int k = 0; k < 5; k++) hipLaunchKernelGGL(gather_kernel<AUX>, dim3(grid), dim3(256), 0, 0, n, d_idx, d_x, (uint32_t)(m * 8), T, d_out); \
      CHECK(hipEventRecord(e1, 0));                                                                                                   \
      CHECK(hipEventSynchronize(e1));                                                                                                 \
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));                                                                              \
      if (ms / 5 < best) best = ms / 5;                                                                                               \
    }                                                                                                                                 \
    printf("{\"hot_share_pct\": %d, \"aux\": %d, \"variant\": \"%s\", \"us\": %.1f, \"G_gathers_per_s\": %.1f}\n", hotpct, AUX,              \
           AUX == 0 ? "one plain load" : (AUX == 16 ? "plain (hot) + sc1 (rest)" : (AUX == 2 ? "plain (hot) + nt (rest)" : "plain (hot) + sc0 sc1 (rest)")), best * 1e3, \
           (double)n / (best * 1e-3) / 1e9);                                                                                          \
    fflush(stdout);                                                                                                                   \
  }
    RUN(0) RUN(16) RUN(2) RUN(17)
  }
  return 0;
}
