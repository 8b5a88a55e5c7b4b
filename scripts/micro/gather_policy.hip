// Micro-benchmark: does any cache policy of a divergent 8-byte gather fetch less than a 128-byte line per L2 miss?
// Random gathers over tables of 32 MiB and 128 MiB (x of R-MAT scale 22 / 24) through raw buffer loads with every combination of the
// gfx950 cache-policy bits sc0 (1), nt (2), sc1 (16).  Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_policy.hip -o scripts/micro/gather_policy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(c) do { hipError_t e = (c); if (e != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e)); return 1; } } while (0)

typedef double v2f __attribute__((ext_vector_type(2)));
template <int AUX>
__global__ __launch_bounds__(256) void gather_kernel(int64_t n, const int32_t *__restrict__ idx, const double *x, uint32_t xbytes, double *__restrict__ out)
{
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)xbytes, 0x00020000);
  const int64_t stride = (int64_t)gridDim.x * 256;
  double        s = 0.0;
  int64_t       i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const int32_t a = idx[i], b = idx[i + stride], c = idx[i + 2 * stride], d = idx[i + 3 * stride];
    double v[4];
    int32_t q[4] = {a, b, c, d};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, q[u] * 8, 0, AUX);
      v[u] = __builtin_bit_cast(double, w);
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
  const int64_t mmax = 1 << 24;
  CHECK(hipMalloc((void **)&d_idx, sizeof(int32_t) * n));
  CHECK(hipMalloc((void **)&d_x, sizeof(double) * mmax));
  CHECK(hipMalloc((void **)&d_out, sizeof(double) * grid * 256));
  CHECK(hipMemset(d_x, 0, sizeof(double) * mmax));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  uint64_t rs = 88172645463325252ull;
  for (int64_t m : {(int64_t)1 << 19, (int64_t)1 << 22, (int64_t)1 << 24}) {
    for (int64_t i = 0; i < n; i++) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; h[(size_t)i] = (int32_t)(rs % (uint64_t)m); }
    CHECK(hipMemcpy(d_idx, h.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
#define RUN(AUX)                                                                                                                      \
  {                                                                                                                                   \
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(gather_kernel<AUX>, dim3(grid), dim3(256), 0, 0, n, d_idx, d_x, (uint32_t)(m * 8), d_out); \
    CHECK(hipDeviceSynchronize());                                                                                                    \
    float best = 1e30f;                                                                                                               \
    for (int rep = 0; rep < 3; rep++) {                                                                                               \
      CHECK(hipEventRecord(e0, 0));                                                                                                   \
      for (int k = 0; k < 5; k++) hipLaunchKernelGGL(gather_kernel<AUX>, dim3(grid), dim3(256), 0, 0, n, d_idx, d_x, (uint32_t)(m * 8), d_out); \
      CHECK(hipEventRecord(e1, 0));                                                                                                   \
      CHECK(hipEventSynchronize(e1));                                                                                                 \
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));                                                                              \
      if (ms / 5 < best) best = ms / 5;                                                                                               \
    }                                                                                                                                 \
    printf("{\"table_MiB\": %lld, \"aux\": %d, \"policy\": \"%s%s%s\", \"us\": %.1f, \"G_gathers_per_s\": %.1f}\n", (long long)(m * 8 >> 20), AUX, (AUX & 1) ? "sc0 " : "", \
           (AUX & 16) ? "sc1 " : "", (AUX & 2) ? "nt" : "", best * 1e3, (double)n / (best * 1e-3) / 1e9);                              \
    fflush(stdout);                                                                                                                   \
  }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(16) RUN(17) RUN(18) RUN(19)
  }
  return 0;
}
