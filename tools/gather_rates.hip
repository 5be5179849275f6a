// Micro-benchmark for the field gathers of k_prep: how fast can a wave fetch,
// for each of its 64 particles, R runs of 96 contiguous bytes (one corner column of the wind pack: 2 levels x
// 2 slots x 3 components, fp64) from a 433 MB array?
//   A  per-lane: every lane issues 6 global_load_dwordx4 per run for its own particle (what k_prep does)
//   B  quad-cooperative: the 4 lanes of a quad fetch 64 contiguous bytes of ONE particle per instruction
//      (1.5 instructions per run), and the data reach their owner through LDS (ds_write_b128 / ds_read_b128)
// Both sum all doubles fetched so that the result can be compared.  Run addresses: cell-sorted particles
// (consecutive lanes = consecutive levels of a column, as after the locality sort) or random cells.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_rates.hip -o /tmp/gather_rates ; run: /tmp/gather_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRuns = 8;          // 2 interpolations x 4 corner columns
constexpr int kChunk = 6;         // 16-byte chunks per run

__global__ void __launch_bounds__(256) k_per_lane(const double2 *__restrict__ f, const unsigned int *__restrict__ run, long long n, double *__restrict__ out) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  double s = 0;
#pragma unroll
  for (int r = 0; r < kRuns; r++) {
    const double2 *q = f + (size_t)run[p * kRuns + r] * 3;   // run index in units of 48 bytes (one level record)
#pragma unroll
    for (int c = 0; c < kChunk; c++) { const double2 v = q[c]; s += v.x + v.y; }
  }
  out[p] = s;
}

// LDS: per wave, 64 particles x 6 chunks of one run at a time (6 KB per wave), padded against bank conflicts
__global__ void __launch_bounds__(256) k_quad(const double2 *__restrict__ f, const unsigned int *__restrict__ run, long long n, double *__restrict__ out) {
  __shared__ double2 stage[4][64 * kChunk + 64];
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q4 = lane & 3, quad0 = lane & ~3;
  double2 *st = stage[wave];
  double s = 0;
  for (int r = 0; r < kRuns; r++) {
    const unsigned int mine = p < n ? run[p * kRuns + r] : 0u;
    // 4 particles of the quad x 6 chunks = 24 chunks = 6 quad-wide loads
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const int g = k * 4 + q4;              // chunk number 0..23 within the quad's 4 runs
      const int owner = g / kChunk, c = g % kChunk;
      const unsigned int base = __shfl(mine, quad0 + owner, 64);
      const double2 v = f[(size_t)base * 3 + c];
      st[(quad0 + owner) * kChunk + c + ((quad0 + owner) >> 2)] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
    for (int c = 0; c < kChunk; c++) { const double2 v = st[lane * kChunk + c + (lane >> 2)]; s += v.x + v.y; }
    __builtin_amdgcn_wave_barrier();
  }
  if (p < n) out[p] = s;
}

int main() {
  const int nx = 361, ny = 181, nz = 138;
  const size_t nrec = (size_t)nx * ny * nz;                 // 48-byte records
  const long long n = 10000000;
  std::vector<double> hf(nrec * 6);
  for (size_t i = 0; i < hf.size(); i++) hf[i] = (double)(i % 1000) * 1e-3;
  double *f; unsigned int *run; double *out;
  hipMalloc(&f, hf.size() * 8); hipMemcpy(f, hf.data(), hf.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&run, (size_t)n * kRuns * 4); hipMalloc(&out, n * 8);
  std::vector<unsigned int> hr((size_t)n * kRuns);
  std::vector<double> o1(n), o2(n);
  for (int mode = 0; mode < 2; mode++) {
    unsigned long long x = 88172645463325252ull;
    for (long long p = 0; p < n; p++) {
      size_t cell;
      if (mode == 0) cell = (size_t)((double)p / n * (nrec - (size_t)nz * (nx + 2) - 2));      // sorted: about one particle per cell-level
      else { x ^= x << 13; x ^= x >> 7; x ^= x << 17; cell = x % (nrec - (size_t)nz * (nx + 2) - 2); }
      for (int r = 0; r < kRuns; r++) {
        const size_t col = (r & 1) * nz + ((r >> 1) & 1) * (size_t)nz * nx;     // the 4 corner columns; second interpolation = same cell
        hr[p * kRuns + r] = (unsigned int)(cell + col);
      }
    }
    hipMemcpy(run, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; v++) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        if (v == 0) k_per_lane<<<(unsigned)((n + 255) / 256), 256>>>((const double2 *)f, run, n, out);
        else k_quad<<<(unsigned)((n + 255) / 256), 256>>>((const double2 *)f, run, n, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      hipMemcpy(v == 0 ? o1.data() : o2.data(), out, n * 8, hipMemcpyDeviceToHost);
      printf("%-7s %-9s %7.3f ms  %6.1f GB/s of run bytes\n", mode == 0 ? "sorted" : "random", v == 0 ? "per-lane" : "quad+LDS", best,
             (double)n * kRuns * 96 / (best * 1e-3) / 1e9);
    }
    long long bad = 0;
    for (long long p = 0; p < n; p++) if (o1[p] != o2[p]) bad++;
    printf("        results differ for %lld particles\n", bad);
  }
  return 0;
}
