// Which dynamic-LDS sizes still let three 256-thread blocks share a CU?  (sizes the Langevin kernel's stash is budgeted against)
//   hipcc --offload-arch=gfx950 -O2 tools/lds_occupancy.hip -o tools/lds_occupancy && tools/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned char smem[];
__global__ void __launch_bounds__(256) k(int *out) { smem[threadIdx.x] = 1; __syncthreads(); out[threadIdx.x] = smem[255 - threadIdx.x]; }
int main() {
  (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int prev = -1;
  for (int b = 50000; b <= 56000; b += 16) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, b) != hipSuccess) { printf("error at %d\n", b); return 1; }
    if (n != prev) printf("%d bytes: %d blocks per CU\n", b, n);
    prev = n;
  }
  return 0;
}
