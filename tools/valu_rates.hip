// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU
// instructions the Langevin kernel is made of, on the device it runs on.
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o /tmp/valu_rates ; run: /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define ITER 2048

#define KERNEL64(name, insn)                                                        \
  __global__ void k_##name(double *out, double a, double b) {                      \
    double r0 = a + threadIdx.x, r1 = a * 2 + threadIdx.x, r2 = a * 3, r3 = a * 5; \
    double c = b;                                                                   \
    for (int i = 0; i < ITER; i++) {                                                \
      REP8(asm volatile(insn " %0, %0, %4\n" insn " %1, %1, %4\n" insn " %2, %2, %4\n" insn " %3, %3, %4" \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));)        \
    }                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;                \
  }
#define KERNEL64_3(name, insn)                                                      \
  __global__ void k_##name(double *out, double a, double b) {                      \
    double r0 = a + threadIdx.x, r1 = a * 2 + threadIdx.x, r2 = a * 3, r3 = a * 5; \
    double c = b;                                                                   \
    for (int i = 0; i < ITER; i++) {                                                \
      REP8(asm volatile(insn " %0, %0, %4, %4\n" insn " %1, %1, %4, %4\n" insn " %2, %2, %4, %4\n" insn " %3, %3, %4, %4" \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));)        \
    }                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;                \
  }
#define KERNEL64_1(name, insn)                                                      \
  __global__ void k_##name(double *out, double a, double b) {                      \
    double r0 = a + threadIdx.x, r1 = a * 2 + threadIdx.x, r2 = a * 3, r3 = a * 5; \
    for (int i = 0; i < ITER; i++) {                                                \
      REP8(asm volatile(insn " %0, %0\n" insn " %1, %1\n" insn " %2, %2\n" insn " %3, %3" \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)                 \
    }                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;                \
  }
#define KERNEL32(name, insn)                                                        \
  __global__ void k_##name(double *out, double a, double b) {                      \
    unsigned r0 = (unsigned)a + threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u; \
    unsigned c = (unsigned)b | 1u;                                                  \
    for (int i = 0; i < ITER; i++) {                                                \
      REP8(asm volatile(insn " %0, %0, %4\n" insn " %1, %1, %4\n" insn " %2, %2, %4\n" insn " %3, %3, %4" \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));)        \
    }                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(r0 + r1 + r2 + r3);      \
  }
#define KERNEL32_3(name, insn)                                                      \
  __global__ void k_##name(double *out, double a, double b) {                      \
    unsigned r0 = (unsigned)a + threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u; \
    unsigned c = (unsigned)b | 1u;                                                  \
    for (int i = 0; i < ITER; i++) {                                                \
      REP8(asm volatile(insn " %0, %0, %4, %4\n" insn " %1, %1, %4, %4\n" insn " %2, %2, %4, %4\n" insn " %3, %3, %4, %4" \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));)        \
    }                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(r0 + r1 + r2 + r3);      \
  }
#define KERNEL32_1(name, insn)                                                      \
  __global__ void k_##name(double *out, double a, double b) {                      \
    float r0 = (float)a + threadIdx.x, r1 = r0 * 3.f, r2 = r0 * 5.f, r3 = r0 * 7.f; \
    for (int i = 0; i < ITER; i++) {                                                \
      REP8(asm volatile(insn " %0, %0\n" insn " %1, %1\n" insn " %2, %2\n" insn " %3, %3" \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)                 \
    }                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(r0 + r1 + r2 + r3);      \
  }

KERNEL64(add_f64, "v_add_f64")
KERNEL64(mul_f64, "v_mul_f64")
KERNEL64_3(fma_f64, "v_fma_f64")
KERNEL64(min_f64, "v_min_f64")
KERNEL64_1(rcp_f64, "v_rcp_f64")
KERNEL64_1(rsq_f64, "v_rsq_f64")
KERNEL64_1(sqrt_f64, "v_sqrt_f64")
KERNEL64_1(fract_f64, "v_fract_f64")
KERNEL64_1(floor_f64, "v_floor_f64")
KERNEL64_1(mov_b64, "v_mov_b64")
KERNEL32(xor_b32, "v_xor_b32")
KERNEL32(add_u32, "v_add_u32")
KERNEL32(mul_lo_u32, "v_mul_lo_u32")
KERNEL32(mul_hi_u32, "v_mul_hi_u32")
KERNEL32(mul_u32_u24, "v_mul_u32_u24")
KERNEL32(add_f32, "v_add_f32")
KERNEL32_3(fma_f32, "v_fma_f32")
KERNEL32_3(alignbit, "v_alignbit_b32")
KERNEL32_3(mad_u32_u24, "v_mad_u32_u24")
KERNEL32_1(exp_f32, "v_exp_f32")
KERNEL32_1(log_f32, "v_log_f32")
KERNEL32_1(rcp_f32, "v_rcp_f32")
KERNEL32_1(sqrt_f32, "v_sqrt_f32")
KERNEL32_1(sin_f32, "v_sin_f32")

// packed f32: two lanes' worth of f32 per 64-bit register pair and instruction (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32):
// does a two-particles-per-lane f32 fine loop issue at twice the scalar-f32 flop rate?
#define KERNELPK(name, insn, nops)                                                  \
  __global__ void k_##name(double *out, double a, double b) {                      \
    double r0 = a + threadIdx.x, r1 = a * 2 + threadIdx.x, r2 = a * 3, r3 = a * 5; \
    double c = b;                                                                   \
    for (int i = 0; i < ITER; i++) {                                                \
      REP8(asm volatile(insn " %0, %0, " nops "\n" insn " %1, %1, " nops "\n" insn " %2, %2, " nops "\n" insn " %3, %3, " nops \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));)        \
    }                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;                \
  }
KERNELPK(pk_fma_f32, "v_pk_fma_f32", "%4, %4")
KERNELPK(pk_mul_f32, "v_pk_mul_f32", "%4")
KERNELPK(pk_add_f32, "v_pk_add_f32", "%4")
KERNEL32(mul_f32, "v_mul_f32")
KERNEL32(max_f32, "v_max_f32")

// 64-bit results from 32-bit operands
__global__ void k_mad_u64_u32(double *out, double a, double b) {
  unsigned long long r0 = (unsigned long long)a + threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u;
  unsigned c = (unsigned)b | 1u;
  for (int i = 0; i < ITER; i++) {
    REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\nv_mad_u64_u32 %1, vcc, %4, %4, %1\nv_mad_u64_u32 %2, vcc, %4, %4, %2\nv_mad_u64_u32 %3, vcc, %4, %4, %3"
                      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c) : "vcc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(r0 + r1 + r2 + r3);
}
__global__ void k_cvt_f64_f32(double *out, double a, double b) {
  float f0 = (float)a + threadIdx.x, f1 = f0 * 2, f2 = f0 * 3, f3 = f0 * 5;
  double r0, r1, r2, r3;
  for (int i = 0; i < ITER; i++) {
    REP8(asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7"
                      : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
}
__global__ void k_cvt_f32_f64(double *out, double a, double b) {
  double d0 = a + threadIdx.x, d1 = d0 * 2, d2 = d0 * 3, d3 = d0 * 5;
  float r0, r1, r2, r3;
  for (int i = 0; i < ITER; i++) {
    REP8(asm volatile("v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7"
                      : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(r0 + r1 + r2 + r3);
}
// compiler-generated composites (library fp64 math as the kernel uses it)
#define KERNELC(name, expr)                                                         \
  __global__ void k_##name(double *out, double a, double b) {                      \
    double r0 = a + threadIdx.x * 1e-3, r1 = a * 1.1 + threadIdx.x * 1e-3, r2 = a * 1.2, r3 = a * 1.3; \
    for (int i = 0; i < ITER / 8; i++) {                                            \
      REP8({ double x = r0; r0 = expr; x = r1; r1 = expr; x = r2; r2 = expr; x = r3; r3 = expr; }) \
    }                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;                \
  }
KERNELC(c_div, (b / x) + 0.5)
KERNELC(c_sqrt, sqrt(x) + 1.0)
KERNELC(c_exp, exp(-x) + 0.5)
KERNELC(c_log, log(x) + 2.0)
KERNELC(c_erf, erf(x) + 0.5)
KERNELC(c_pow, pow(x, b) + 0.5)
KERNELC(c_expf, (double)__expf(-(float)x) + 0.5)

typedef void (*kern_t)(double *, double, double);
struct T { const char *name; kern_t k; double per_iter; };

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  const double clk = p.clockRate * 1e3;   // Hz
  printf("device %s  CUs %d  clock %.0f MHz\n", p.name, ncu, clk / 1e6);
  const int wavesPerSimd = 8, block = 256;                 // 4 waves per block = 1 per SIMD
  const int grid = ncu * wavesPerSimd;                     // 8 blocks per CU -> 8 waves per SIMD
  double *out; hipMalloc(&out, sizeof(double) * grid * block);
  std::vector<T> ts = {
#define E(n) {#n, k_##n, 32.0 * ITER}
#define EC(n) {#n, k_##n, 4.0 * ITER}
    E(add_f64), E(mul_f64), E(fma_f64), E(min_f64), E(rcp_f64), E(rsq_f64), E(sqrt_f64), E(fract_f64), E(floor_f64), E(mov_b64),
    E(xor_b32), E(add_u32), E(mul_lo_u32), E(mul_hi_u32), E(mul_u32_u24), E(mad_u32_u24), E(mad_u64_u32), E(alignbit),
    E(add_f32), E(mul_f32), E(max_f32), E(fma_f32), E(pk_fma_f32), E(pk_mul_f32), E(pk_add_f32), E(exp_f32), E(log_f32), E(rcp_f32), E(sqrt_f32), E(sin_f32), E(cvt_f64_f32), E(cvt_f32_f64),
    EC(c_div), EC(c_sqrt), EC(c_exp), EC(c_log), EC(c_erf), EC(c_pow), EC(c_expf)};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto &t : ts) {
    hipLaunchKernelGGL(t.k, dim3(grid), dim3(block), 0, 0, out, 1.5, 0.75);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(t.k, dim3(grid), dim3(block), 0, 0, out, 1.5, 0.75);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // each SIMD runs wavesPerSimd waves, each issuing per_iter instructions
    double cyc = ms * 1e-3 * clk / (wavesPerSimd * t.per_iter);
    printf("%-14s %8.3f ms  %7.2f cycles per wave-instruction%s\n", t.name, ms, cyc, t.per_iter < 32.0 * ITER ? " (per call)" : "");
  }
  return 0;
}
