// Micro-benchmark: how many 8-byte gathers per second does an MI355X sustain when the lanes of a wavefront read different cache lines?
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_rate.hip -o scripts/micro/gather_rate ; run on the GPU box.
// Each thread streams its indices (coalesced int32) and sums x[idx]; one double per thread is written at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(c) do { hipError_t e = (c); if (e != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void gather_kernel(int64_t n, const int32_t *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  double        s = 0.0;
  int64_t       i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const int32_t a = idx[i], b = idx[i + stride], c = idx[i + 2 * stride], d = idx[i + 3 * stride];
    s += x[a]; s += x[b]; s += x[c]; s += x[d];
  }
  for (; i < n; i += stride) s += x[idx[i]];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
  const int64_t n = 1 << 25;        /* gathers per launch */
  const int64_t m = 1 << 21;        /* x entries (16.8 MB) */
  std::vector<int32_t> h((size_t)n);
  int32_t *d_idx; double *d_x, *d_out;
  const int grid = 256 * 16;
  CHECK(hipMalloc((void **)&d_idx, sizeof(int32_t) * n));
  CHECK(hipMalloc((void **)&d_x, sizeof(double) * m));
  CHECK(hipMalloc((void **)&d_out, sizeof(double) * grid * 256));
  CHECK(hipMemset(d_x, 0, sizeof(double) * m));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct { const char *name; int64_t window; } cases[] = {{"sequential (coalesced)", 0}, {"random within +-4096 of i*m/n", 4096}, {"random within +-131072", 131072}, {"random over all of x", -1}};
  uint64_t rs = 88172645463325252ull;
  for (auto &cs : cases) {
    for (int64_t i = 0; i < n; i++) {
      rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
      const int64_t base = i * m / n;
      int64_t       v;
      if (cs.window == 0) v = i % m;
      else if (cs.window < 0) v = (int64_t)(rs % (uint64_t)m);
      else { v = base + (int64_t)(rs % (uint64_t)(2 * cs.window)) - cs.window; if (v < 0) v = 0; if (v >= m) v = m - 1; }
      h[(size_t)i] = (int32_t)v;
    }
    CHECK(hipMemcpy(d_idx, h.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, 0, n, d_idx, d_x, d_out);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CHECK(hipEventRecord(e0, 0));
      for (int k = 0; k < 10; k++) hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, 0, n, d_idx, d_x, d_out);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms / 10 < best) best = ms / 10;
    }
    printf("{\"case\": \"%s\", \"gathers\": %lld, \"us\": %.1f, \"G_gathers_per_s\": %.1f, \"clocks_per_gather_per_CU_at_2.4GHz\": %.2f}\n", cs.name, (long long)n, best * 1e3,
           (double)n / (best * 1e-3) / 1e9, best * 1e-3 * 2.4e9 * 256 / (double)n);
    fflush(stdout);
  }
  return 0;
}
