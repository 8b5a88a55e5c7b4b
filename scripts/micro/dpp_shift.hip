#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
  int v = threadIdx.x + 100;
  int r = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);
  out[threadIdx.x] = r;
}
int main()
{
  int *d; hipMalloc(&d, 64 * 4);
  k<<<1, 64>>>(d);
  int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i++) printf("%d ", h[i]);
  printf("\n");
  return 0;
}
