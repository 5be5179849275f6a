// Micro-benchmark: does a DEPENDENT chain of fp64 FMAs issue slower than independent ones on this device,
// and how many waves per SIMD hide it?  (Question behind the single-wave VALU-busy fraction of k_pbl_loop.)
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_dep.hip -o /tmp/valu_dep ; run: /tmp/valu_dep
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 65536
template <int NCHAIN>
__global__ void __launch_bounds__(256) k_dep(double *out, double a, double b) {
  double r0 = a + threadIdx.x, r1 = a * 2 + threadIdx.x, r2 = a * 3, r3 = a * 5;
  const double c = b;
  for (int i = 0; i < ITER; i++) {
    if (NCHAIN == 1) {
      asm volatile("v_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\n"
                   "v_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1" : "+v"(r0) : "v"(c));
    } else if (NCHAIN == 2) {
      asm volatile("v_fma_f64 %0, %0, %2, %2\nv_fma_f64 %1, %1, %2, %2\nv_fma_f64 %0, %0, %2, %2\nv_fma_f64 %1, %1, %2, %2\n"
                   "v_fma_f64 %0, %0, %2, %2\nv_fma_f64 %1, %1, %2, %2\nv_fma_f64 %0, %0, %2, %2\nv_fma_f64 %1, %1, %2, %2" : "+v"(r0), "+v"(r1) : "v"(c));
    } else {
      asm volatile("v_fma_f64 %0, %0, %4, %4\nv_fma_f64 %1, %1, %4, %4\nv_fma_f64 %2, %2, %4, %4\nv_fma_f64 %3, %3, %4, %4\n"
                   "v_fma_f64 %0, %0, %4, %4\nv_fma_f64 %1, %1, %4, %4\nv_fma_f64 %2, %2, %4, %4\nv_fma_f64 %3, %3, %4, %4"
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
}
// the same with a 32-bit instruction between the dependent FMAs (what the compiler's v_mov / selects do)
__global__ void __launch_bounds__(256) k_dep_mixed(double *out, double a, double b) {
  double r0 = a + threadIdx.x;
  unsigned u = threadIdx.x;
  const double c = b;
  for (int i = 0; i < ITER; i++)
    asm volatile("v_fma_f64 %0, %0, %2, %2\nv_add_u32 %1, %1, %1\nv_fma_f64 %0, %0, %2, %2\nv_add_u32 %1, %1, %1\n"
                 "v_fma_f64 %0, %0, %2, %2\nv_add_u32 %1, %1, %1\nv_fma_f64 %0, %0, %2, %2\nv_add_u32 %1, %1, %1" : "+v"(r0), "+v"(u) : "v"(c));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + u;
}

template <typename F> static double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double *out;
  hipMalloc(&out, sizeof(double) * 256 * cus * 8);
  const double ghz = p.clockRate * 1e-6;
  printf("device %s, %d CUs, %.2f GHz nominal\n", p.name, cus, ghz);
  for (int wps : {1, 2, 3, 4, 6, 8}) {   // waves per SIMD = blocks of 256 threads per CU
    const int grid = cus * wps;
    const double n = 8.0 * ITER * wps;   // FMAs per SIMD
    double t1 = time_ms([&] { hipLaunchKernelGGL(k_dep<1>, dim3(grid), dim3(256), 0, 0, out, 1.0, 1.0000001); });
    double t2 = time_ms([&] { hipLaunchKernelGGL(k_dep<2>, dim3(grid), dim3(256), 0, 0, out, 1.0, 1.0000001); });
    double t4 = time_ms([&] { hipLaunchKernelGGL(k_dep<4>, dim3(grid), dim3(256), 0, 0, out, 1.0, 1.0000001); });
    double tm = time_ms([&] { hipLaunchKernelGGL(k_dep_mixed, dim3(grid), dim3(256), 0, 0, out, 1.0, 1.0000001); });
    printf("waves/SIMD %d: cycles per fp64 FMA per SIMD (nominal clock): 1 chain %.2f, 2 chains %.2f, 4 chains %.2f; dependent FMA + add_u32 pairs: %.2f per pair\n",
           wps, t1 * 1e-3 * ghz * 1e9 / n, t2 * 1e-3 * ghz * 1e9 / n, t4 * 1e-3 * ghz * 1e9 / n, tm * 1e-3 * ghz * 1e9 / (4.0 * ITER * wps));
  }
  return 0;
}
