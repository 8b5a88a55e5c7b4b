// Micro-benchmark: how long does the address unit of a CU take for one 64-lane load that HITS in L1, by shape of the request?
//   stride-2: 8 bytes a lane, every other entry (lane t reads p[2 t]): what a lane owning two ADJACENT rows asks for, per row
//   mode 1: 8 bytes a lane, consecutive entries (lane t reads p[t])
//   mode 2: 16 bytes a lane, consecutive (lane t reads p[2 t], p[2 t + 1] in one request)
//   mode 3: 2 bytes a lane, consecutive (the code loads of the SELL-dict kernel)
//   mode 4: 16 bytes a lane, consecutive but starting 8 bytes off a 16-byte boundary
// Every wave reads a window of 8 KiB of its own workgroup's 16 KiB, so all requests hit in L1 after the first pass.
// Prints clocks per request per CU (wall time x clock / requests per CU).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/ta_rate.hip -o scripts/micro/ta_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(c) do { hipError_t e = (c); if (e != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e)); return 1; } } while (0)

typedef double f64x2a8 __attribute__((ext_vector_type(2), aligned(8)));

template <int MODE>
__global__ __launch_bounds__(256) void ta_kernel(const double *__restrict__ x, double *__restrict__ out, int iters)
{
  const int     t = threadIdx.x & 63;
  const double *p = x + (size_t)blockIdx.x * 2048 + (threadIdx.x >> 6) * 256; /* 16 KiB a workgroup */
  double        s = 0.0;
  for (int i = 0; i < iters; i++) {
    asm volatile("" : "+v"(p)); /* the loads stay in the loop */
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int o = ((i + u) & 7) * 128; /* 8 windows of 1 KiB */
      if (MODE == 1) s += p[o + t];
      if (MODE == 2) { const double2 g = *reinterpret_cast<const double2 *>(p + o + 2 * t); s += g.x + g.y; }
      if (MODE == 3) s += (double)reinterpret_cast<const unsigned short *>(p + o)[t];
      if (MODE == 4) { const f64x2a8 g = *reinterpret_cast<const f64x2a8 *>(p + o + 2 * t + 1); s += g.x + g.y; }
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void ta_kernel_stride2(const double *__restrict__ x, double *__restrict__ out, int iters)
{
  const int     t = threadIdx.x & 63;
  const double *p = x + (size_t)blockIdx.x * 2048 + (threadIdx.x >> 6) * 256;
  double        s = 0.0;
  for (int i = 0; i < iters; i++) {
    asm volatile("" : "+v"(p)); /* the loads stay in the loop */
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int o = ((i + u) & 7) * 128;
      s += p[o + 2 * t];
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
  const int grid = 256 * 8, iters = 2000;
  double   *d_x, *d_out;
  CHECK(hipMalloc((void **)&d_x, sizeof(double) * ((size_t)grid * 2048 + 4096)));
  CHECK(hipMalloc((void **)&d_out, sizeof(double) * grid * 256));
  CHECK(hipMemset(d_x, 0, sizeof(double) * ((size_t)grid * 2048 + 4096)));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const double mhz = prop.clockRate / 1e3;
  hipEvent_t   e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
#define RUN(NAME, KERNEL)                                                                                              \
  {                                                                                                                    \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, 0, d_x, d_out, iters);                                        \
    CHECK(hipDeviceSynchronize());                                                                                     \
    CHECK(hipEventRecord(e0, 0));                                                                                      \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, 0, d_x, d_out, iters);                                        \
    CHECK(hipEventRecord(e1, 0));                                                                                      \
    CHECK(hipEventSynchronize(e1));                                                                                    \
    float ms;                                                                                                          \
    CHECK(hipEventElapsedTime(&ms, e0, e1));                                                                           \
    const double req_per_cu = (double)grid * 4 * iters * 8 / prop.multiProcessorCount;                                 \
    printf("{\"shape\": \"%s\", \"ms\": %.3f, \"ns_per_request_per_cu\": %.2f, \"clocks_at_%.0f_MHz\": %.2f}\n", NAME, ms,     \
           ms * 1e6 / req_per_cu, mhz, ms * 1e3 * mhz / req_per_cu);                                                   \
  }
  RUN("8B stride-2", ta_kernel_stride2);
  RUN("8B consecutive", ta_kernel<1>);
  RUN("16B consecutive", ta_kernel<2>);
  RUN("2B consecutive", ta_kernel<3>);
  RUN("16B consecutive, 8 bytes off", ta_kernel<4>);
  return 0;
}
