// fpx_engine.hip -- HIP kernels (gfx950) and the C ABI of include/flexpart_amd.h.
//
// One process drives one GPU through one handle.  All particle state and all
// met fields stay resident in HBM between calls; a step is a single launch of
// k_advance (one thread per particle slot) on the handle's stream.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstring>
#include <rccl/rccl.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/flexpart_amd.h"
#include "fpx_tu.hpp"
#include "fpx_device.hpp"
#include "fpx_verttransform.hpp"
#include "fpx_calcpar.hpp"
#include "fpx_convect.hpp"
#include "fpx_rng_host.hpp"

namespace fpx {

// shared by the translation units (fpx_tu.hpp): the message behind fpx_last_error()
#if FPX_TU_REAL == 4 || FPX_TU_PART > 0
extern thread_local std::string g_err;
#else
thread_local std::string g_err;
#endif
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
FPX_TU_OPEN

#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(FPX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

constexpr int kBlock = 256;
// The big kernels read the fields of their first argument, the View, THROUGH the kernel-argument segment instead of from the
// by-value parameter.  A by-value aggregate is a private copy that the optimiser splits into one scalar per field, all loaded at the
// kernel's entry: in k_prep that is 40 scalar loads whose results do not fit the scalar file and travel as lanes of a vector
// register (800-1300 v_readlane per kernel; polar k_prep 0.78 instead of 0.71 ms) -- and as soon as ONE access has an address
// the optimiser cannot resolve (a per-lane subscript, a select between two fields' addresses) the whole copy lives in scratch
// (576-816 B per lane in three instance families before this).  Read in place, every field is a scalar load from constant memory
// next to its use, re-loaded rather than spilled, and no copy exists that could end up in scratch.  The View must be the kernel's
// FIRST parameter (offset 0 of the segment).
#ifndef FPX_VIEW_BY_VALUE
#define FPX_VIEW_FROM_KERNARG(V, V_arg) const View<R> &V = *(const View<R> *)(const void *)__builtin_amdgcn_kernarg_segment_ptr(); (void)V_arg
#else
#define FPX_VIEW_FROM_KERNARG(V, V_arg) const View<R> &V = V_arg
#endif
#ifndef FPX_PREP_WAVES
#define FPX_PREP_WAVES 3   // register budget (waves per SIMD) of k_prep (<= 168 VGPRs)
#endif
#ifndef FPX_INIT_PREP_3WAVES
#define FPX_INIT_PREP_3WAVES 1    // the INIT instance of k_prep at the three-wave budget too: its initialize() body runs only in waves that hold a new particle
#endif
#ifndef FPX_PREP_TWO_WAVES   // which instances of k_prep keep the two-wave register budget: none since the arguments are read in place (the nest
// instances fit 168 VGPRs without a spill, the ones with dry deposition spill 6-12 registers; round 3: every INIT / POLAR / NEST / DRYDEP instance at two waves)
#define FPX_PREP_TWO_WAVES(INIT, POLAR, NEST, DRYDEP) ((INIT && !FPX_INIT_PREP_3WAVES) || (POLAR && !FPX_POLAR_PREP_3WAVES))
#endif
#ifndef FPX_POLAR_PREP_3WAVES
#define FPX_POLAR_PREP_3WAVES 1   // the polar instance of k_prep at the three-wave register budget (the polar move is out of line: polar_move)
#endif
#ifndef FPX_FINISH_WAVES
#define FPX_FINISH_WAVES 2 // register budget of k_pbl_finish
#endif
#ifndef FPX_LOOP_WAVES_F32
#define FPX_LOOP_WAVES_F32 4   // the f32 instances fit 128 VGPRs: four waves per SIMD (measured: 168 -> 184 ms at 1e8 when they slipped to three)
#endif
#ifndef FPX_LOOP_WAVES
#define FPX_LOOP_WAVES 3   // waves per SIMD the Langevin kernel is register-budgeted for (<= 168 VGPRs)
#endif
#ifndef FPX_COST_BUCKETS
#define FPX_COST_BUCKETS -1  // the work list orders each class by the expected number of passes, longest first (k_prep): -1 = for clouds below 5e7 particles
#endif
#ifndef FPX_SLICE_SCHEDULE
#define FPX_SLICE_SCHEDULE 0   // pass budgets of the successive launches of the Langevin kernel, 0 = none; one entry = ONE launch (measured best)
#endif
#ifndef FPX_DRAIN_LANES
#define FPX_DRAIN_LANES 32   // a wave of a non-final launch whose list is used up hands its particles on when fewer lanes than this hold one
#endif
constexpr int kMaxNz = 512;
constexpr unsigned char kKeyDone = 32, kKeyNotDue = 33;   // keys of the work-list sort (6 bits); 0..31: PBL particles, class-major (k_prep)

// ---------------------------------------------------------------------------
// field repacking kernels
// ---------------------------------------------------------------------------
// 3-D field (x fastest, strides nxmax,nymax) -> out[((jy*nx+ix)*nz + k)*stride + off]
// (z fastest).  32x32 LDS tile transposes x<->z so both sides move whole rows.
template <typename H, typename R>
__global__ void k_pack3(const H *__restrict__ in, R *__restrict__ out, int nx, int ny, int nz, int nxmax, int nymax,
                        int stride, int off) {
  __shared__ R tile[32][33];
  const int jy = blockIdx.z;
  const int x0 = blockIdx.x * 32, z0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int x = x0 + threadIdx.x, z = z0 + r;
    if (x < nx && z < nz) tile[r][threadIdx.x] = (R)in[(size_t)x + (size_t)nxmax * ((size_t)jy + (size_t)nymax * z)];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int x = x0 + r, z = z0 + threadIdx.x;
    if (x < nx && z < nz) out[(((size_t)jy * nx + x) * nz + z) * stride + off] = tile[threadIdx.x][r];
  }
}

template <typename H, typename R>
__global__ void k_pack2(const H *__restrict__ in, R *__restrict__ out, int nx, int ny, int nxmax, int stride, int off) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nx * ny) return;
  int ix = i % nx, jy = i / nx;
  out[(size_t)i * stride + off] = (R)in[(size_t)ix + (size_t)nxmax * jy];
}

// hcell[jy][ix] = max of hmix over the cell's corners and both slots: the loop of
// advance.f90:238-252 (start value 0, jyp clamp of :228-231) == initialize.f90:83-90
template <typename R>
__global__ void k_hcell(const R *__restrict__ sfc, R *__restrict__ hcell, int nx, int ny) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nx * ny) return;
  int ix = i % nx, jy = i / nx;
  int ixp = min(ix + 1, nx - 1), jyp = min(jy + 1, ny - 1);
  R h = (R)0;
  for (int s = 0; s < 2; s++) {
    h = fmax(h, sfc[(((size_t)jy * nx + ix) * 2 + s) * 4 + 3]);
    h = fmax(h, sfc[(((size_t)jy * nx + ixp) * 2 + s) * 4 + 3]);
    h = fmax(h, sfc[(((size_t)jyp * nx + ix) * 2 + s) * 4 + 3]);
    h = fmax(h, sfc[(((size_t)jyp * nx + ixp) * 2 + s) * 4 + 3]);
  }
  hcell[i] = h;
}

// ---------------------------------------------------------------------------
// particle I/O kernels: host order (pid) <-> device slots
// ---------------------------------------------------------------------------
template <typename S, typename D>
__global__ void k_scatter_in(const S *__restrict__ src, D *__restrict__ dst, const unsigned int *__restrict__ slot_of_pid,
                             long long first, long long count) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  long long slot = slot_of_pid ? (long long)slot_of_pid[first + i] : first + i;
  dst[slot] = (D)src[i];
}
template <typename S, typename D>
__global__ void k_gather_out(const S *__restrict__ src, D *__restrict__ dst, const unsigned int *__restrict__ slot_of_pid,
                             long long first, long long count) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  long long slot = slot_of_pid ? (long long)slot_of_pid[first + i] : first + i;
  dst[i] = (D)src[slot];
}
template <typename D>
__global__ void k_fill(D *__restrict__ dst, D val, long long first, long long count, const unsigned int *__restrict__ slot_of_pid) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  long long slot = slot_of_pid ? (long long)slot_of_pid[first + i] : first + i;
  dst[slot] = val;
}
__global__ void k_iota_pid(unsigned int *__restrict__ pid, long long first, long long count) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  pid[first + i] = (unsigned int)(first + i);
}

// ---------------------------------------------------------------------------
// locality sort: key = (jy, ix, level), the order in which the packed fields lie in
// HBM; dead particles sort to the end.  Keeps waves homogeneous (same cell => same
// mixing height, stability regime and similar sub-step counts) and turns the field
// gathers of a wave into a few shared cache lines.
// ---------------------------------------------------------------------------
template <typename R>
__global__ void k_sort_keys(View<R> V, Parts<R> P, long long n, unsigned int *__restrict__ keys, unsigned int *__restrict__ vals,
                            unsigned int dead_key) {
  __shared__ R hgt[kMaxNz];
  for (int k = threadIdx.x; k < V.nz; k += blockDim.x) hgt[k] = V.height[k];
  __syncthreads();
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int key = dead_key;
  if (P.itra1[i] != kDead) {
    double xt = P.xt[i], yt = P.yt[i];
    if (xt >= 0. && xt <= (double)V.nxmin1 && yt >= 0. && yt <= (double)V.nymin1) {
      int ix = min((int)xt, V.nx - 1), jy = min((int)yt, V.ny - 1);
      int lev = find_level(hgt, V.nz, P.zt[i]);   // 1 .. nz-1
      key = ((unsigned int)jy * V.nx + ix) * (unsigned int)V.nz + (unsigned int)lev;
    }
  }
  keys[i] = key;
  vals[i] = (unsigned int)i;
}

// keys of the inverse operation: the particle number, so that the sort puts every particle back into the storage space
// of its number (slot = pid)
__global__ void k_sort_keys_pid(const unsigned int *__restrict__ pid, long long n, unsigned int *__restrict__ keys, unsigned int *__restrict__ vals) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = pid[i];
  vals[i] = (unsigned int)i;
}

// out[i] = in[perm[i]] for every particle array at once.  Between two re-sorts a particle stays in or next to its grid
// column, so the sources of neighbouring destination blocks share cache lines: workgroups are dealt to the eight XCDs
// round-robin, therefore block b takes tile (b % 8) * tiles_per_xcd + b / 8 -- each XCD's L2 then sees one contiguous
// eighth of the destination range and every source line is fetched from HBM by one XCD instead of by up to eight.
// The arrays go in four groups, one launch each: the source lines a workgroup touches are shared with its neighbours
// (a grid column's particles are re-shuffled among its levels), and only with a few arrays in flight does that
// working set stay in the 4 MB L2 of an XCD until the neighbours have used it (all 18 arrays in one launch: twice the
// compulsory reads).
template <typename R, int GROUP>
__global__ void k_permute(Parts<R> A, Parts<R> B, const unsigned int *__restrict__ perm, long long n, int nspec, int tiles_per_xcd) {
  const long long tile = (long long)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
  long long i = tile * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned int j = perm[i];
  if (GROUP == 0) { B.xt[i] = A.xt[j]; B.yt[i] = A.yt[j]; B.zt[i] = A.zt[j]; B.idt[i] = A.idt[j]; }
  if (GROUP == 1) { B.up[i] = A.up[j]; B.vp[i] = A.vp[j]; B.wp[i] = A.wp[j]; B.itra1[i] = A.itra1[j]; }
  if (GROUP == 2) { B.us[i] = A.us[j]; B.vs[i] = A.vs[j]; B.ws[i] = A.ws[j]; B.itramem[i] = A.itramem[j]; }
  if (GROUP == 3) {
    B.npoint[i] = A.npoint[j]; B.nclass[i] = A.nclass[j]; B.cbt[i] = A.cbt[j]; B.itrasplit[i] = A.itrasplit[j];
    B.pid[i] = A.pid[j];
    for (int ks = 0; ks < nspec; ks++) B.xmass1[(size_t)ks * B.cap + i] = A.xmass1[(size_t)ks * A.cap + j];
    if (A.xscav) for (int ks = 0; ks < nspec; ks++) B.xscav[(size_t)ks * B.cap + i] = A.xscav[(size_t)ks * A.cap + j];
  }
}

// ---- convection (fpx_convect.hpp): small kernels and the random-number sources of redist ---------------------------
template <typename S, typename D>
__global__ void k_conv_pack(const S *__restrict__ src, D *__restrict__ dst, int nx, int ny, int nlev, int nxmax, int nymax) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, n2 = (long long)nx * ny;
  if (i >= n2 * nlev) return;
  const int k = (int)(i / n2), r = (int)(i % n2), jy = r / nx, ix = r % nx;
  dst[i] = (D)src[(size_t)ix + (size_t)nxmax * ((size_t)jy + (size_t)nymax * k)];
}
template <typename H>
struct ConvRngSeq {     // the number the host drew for this particle from the serial ran3 stream
  const H *rn; const unsigned int *pid;
  __device__ H operator()(long long s) const { return rn[pid[s]]; }
};
template <typename R, typename H>
struct ConvRngCtr {     // counter generator: (seed, global particle number, step, stream of redist)
  View<R> V; const unsigned int *pid; unsigned int step;
  __device__ H operator()(long long s) const {
    Rng<R> G;
    make_rng(V, pid[s], step | 0x40000000u, G);
    return (H)(((float)(G.bits(3) >> 8) + 0.5f) * (1.0f / 16777216.0f));
  }
};

// A permutation without locality (the first sort of a freshly seeded or uploaded cloud) would fetch one memory
// transaction per 8-byte element through the direct gather (18 arrays: > 1 kB per particle).  Such a permutation goes
// through one 128-byte record per particle instead: pack (coalesced reads, whole-line writes), then gather whole
// lines and store coalesced.  Species beyond the first keep the direct gather.
template <typename R>
struct alignas(128) SortRecord {
  double xt, yt;
  R v[8];                // zt up vp wp us vs ws xmass1(1)
  int i[6];              // idt itra1 itramem npoint nclass itrasplit
  unsigned int pid;
  int cbt;
};
template <typename R>
__global__ void k_permute_pack(Parts<R> A, SortRecord<R> *__restrict__ rec, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  SortRecord<R> r;
  r.xt = A.xt[i]; r.yt = A.yt[i];
  r.v[0] = A.zt[i]; r.v[1] = A.up[i]; r.v[2] = A.vp[i]; r.v[3] = A.wp[i]; r.v[4] = A.us[i]; r.v[5] = A.vs[i]; r.v[6] = A.ws[i];
  r.v[7] = A.xmass1[i];
  r.i[0] = A.idt[i]; r.i[1] = A.itra1[i]; r.i[2] = A.itramem[i]; r.i[3] = A.npoint[i]; r.i[4] = A.nclass[i]; r.i[5] = A.itrasplit[i];
  r.pid = A.pid[i]; r.cbt = A.cbt[i];
  rec[i] = r;
}
template <typename R>
__global__ void k_permute_unpack(const SortRecord<R> *__restrict__ rec, Parts<R> A, Parts<R> B, const unsigned int *__restrict__ perm,
                                 long long n, int nspec) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned int j = perm[i];
  const SortRecord<R> r = rec[j];
  B.xt[i] = r.xt; B.yt[i] = r.yt;
  B.zt[i] = r.v[0]; B.up[i] = r.v[1]; B.vp[i] = r.v[2]; B.wp[i] = r.v[3]; B.us[i] = r.v[4]; B.vs[i] = r.v[5]; B.ws[i] = r.v[6];
  B.xmass1[i] = r.v[7];
  B.idt[i] = r.i[0]; B.itra1[i] = r.i[1]; B.itramem[i] = r.i[2]; B.npoint[i] = r.i[3]; B.nclass[i] = r.i[4]; B.itrasplit[i] = r.i[5];
  B.pid[i] = r.pid; B.cbt[i] = (short)r.cbt;
  for (int ks = 1; ks < nspec; ks++) B.xmass1[(size_t)ks * B.cap + i] = A.xmass1[(size_t)ks * A.cap + j];
  if (A.xscav) for (int ks = 0; ks < nspec; ks++) B.xscav[(size_t)ks * B.cap + i] = A.xscav[(size_t)ks * A.cap + j];
}
// number of neighbours in the new order whose sources lie more than `window` storage spaces apart
__global__ void k_perm_disorder(const unsigned int *__restrict__ perm, long long n, unsigned int window, unsigned long long *__restrict__ count) {
  __shared__ unsigned int part[kBlock / 64];
  unsigned int mine = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (i > 0) {
      const long long d = (long long)perm[i] - (long long)perm[i - 1];
      mine += (d < 0 ? -d : d) > (long long)window;
    }
  }
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += part[k];
    if (t) atomicAdd(count, t);
  }
}

// particle number -> storage space; built on demand (release, splitting, up/download by particle number), not by
// every re-sort: a random 4-byte scatter costs a whole memory transaction per particle
__global__ void k_slot_map(const unsigned int *__restrict__ pid, long long cap, unsigned int *__restrict__ slot_of_pid) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) slot_of_pid[pid[i]] = (unsigned int)i;
}

// ---------------------------------------------------------------------------
// synthetic cloud on the device (benchmarks): same SplitMix64 counters and the
// same arithmetic as flexpart_amd/synthetic.py:make_particles
// ---------------------------------------------------------------------------
__device__ __forceinline__ double splitmix_u01(unsigned long long idx1, unsigned long long seed) {
  unsigned long long z = idx1 * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <typename R>
__global__ void k_seed(View<R> V, Parts<R> P, long long n, unsigned long long seed, double frac_pbl, double zmax,
                       double lat_margin, int itime0) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  {
#pragma clang fp contract(off)
  const unsigned long long gi = (unsigned long long)i + V.pid_base + 1ull;   // number in the global cloud
  double ux = splitmix_u01(gi, seed + 1), uy = splitmix_u01(gi, seed + 2);
  double uz = splitmix_u01(gi, seed + 3), us = splitmix_u01(gi, seed + 4);
  const double eps = 361.0 / 3.0e5;
  double x = eps + ux * ((double)(V.nx - 1) - 2.0 * eps);
  double y = lat_margin + uy * ((double)(V.ny - 1) - 2.0 * lat_margin);
  int ix = min((int)x, V.nx - 2), jy = min((int)y, V.ny - 2);
  double hloc = (double)V.hcell[(size_t)jy * V.nx + ix];
  double z = us < frac_pbl ? 10.0 + uz * fmax(hloc - 20.0, 1.0) : hloc + 10.0 + uz * (zmax - hloc - 10.0);
  P.xt[i] = x; P.yt[i] = y; P.zt[i] = (R)z;
  P.up[i] = 0; P.vp[i] = 0; P.wp[i] = 0; P.us[i] = 0; P.vs[i] = 0; P.ws[i] = 0;
  P.idt[i] = 0; P.itra1[i] = itime0; P.itramem[i] = itime0; P.npoint[i] = 1; P.nclass[i] = 1; P.itrasplit[i] = V.ldirect * 999999999;   // "never", signed like itra1 + ldirect*itsplit (releaseparticles.f90:181)
  P.cbt[i] = 1; P.pid[i] = (unsigned int)i;
  for (int ks = 0; ks < V.nspec; ks++) P.xmass1[(size_t)ks * P.cap + i] = (R)1;
  if (P.xscav) for (int ks = 0; ks < V.nspec; ks++) P.xscav[(size_t)ks * P.cap + i] = (R)-1;
  }
}

// ---------------------------------------------------------------------------
// TABLE_SEQ helper: which particles are due / new / take initialize()'s CBL draw
// ---------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(kBlock) k_classify(View<R> V, Parts<R> P, long long numpart, int itime, unsigned char *__restrict__ flags) {
  __shared__ R hgt[kMaxNz];
  for (int k = threadIdx.x; k < V.nz; k += blockDim.x) hgt[k] = V.height[k];
  __syncthreads();
  long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= numpart) return;
  unsigned char f = 0;
  if (P.itra1[s] == itime) {
    f |= 1;
    if (P.itramem[s] == itime || itime == 0) {
      f |= 2;
      double xt = P.xt[s], yt = P.yt[s];
      if (xt >= 0. && xt <= (double)V.nxmin1 && yt >= 0. && yt <= (double)V.nymin1)
        if (initialize_needs_cbl_draws(V, hgt, itime, xt, yt, P.zt[s])) f |= 4;
    }
  }
  flags[P.pid[s]] = f;
}

// ---------------------------------------------------------------------------
// the hot kernels: one pass of the particle loop timemanager.f90:531-712
//
//   k_prep : every due particle -- initialize() if newly released, grid/cell/PBL test;
//            particles above the PBL finish here (one interpol_wind step + Petterssen:
//            gather/stream bound); PBL particles are appended to a work list.
//   k_pbl  : persistent waves over the work list.  Each lane owns one PBL particle and runs
//            passes of the Langevin loop; a lane whose particle has finished its lsynctime
//            immediately refills from the list, so lanes never idle on the data-dependent
//            trip count (15-150 passes per particle).
// ---------------------------------------------------------------------------
template <typename R, typename RNG>
__device__ __forceinline__ void make_rng(const View<R> &V, unsigned int pid, unsigned int step, RNG &G) {
  G.tab = V.rannumb; G.maxrand = V.maxrand; G.mode = V.rng_mode;
  G.pid = pid + V.pid_base; G.step = step;
  G.k0 = (unsigned int)V.seed; G.k1 = (unsigned int)(V.seed >> 32);
  G.cblk = 0xffffffffu;
}

template <typename R>
__device__ __forceinline__ int advance_start_index(const View<R> &V, const SeqRng &S, const Rng<R> &G, unsigned int pid) {
  if (V.rng_mode == 0) return S.nrand_adv[pid];      // advance.f90:153, serial stream replayed by the host
  if (V.rng_mode == 1) return G.start_index(0);
  return 1;
}

// epilogue timemanager.f90:630-708 + write-back of the particle
// TURB: also write up, vp, cbt -- changed only by initialize() and by the Langevin loop; a particle that
// was above the mixing layer for the whole step keeps them (advance.f90:629-708 does not touch them)
// What the epilogue reads of the particle itself (release point, masses): fetched by the caller together with its last
// gather (EpiPre::load inside a `late` hook), so that the epilogue starts with them instead of with two more dependent
// memory round trips at the tail of a latency-bound kernel.
// (the release point and the first species' mass: three registers; the kernels have none to spare for more)
template <typename R>
struct EpiPre {
  int npoint;
  R xm0;
  template <bool DRYDEP>
  __device__ __forceinline__ void load(const View<R> &V, const GridP<R> &Gp, const Parts<R> &P, long long s) {
    const bool massfract = V.mdomainfill == 0 && V.mquasilag == 0;
    npoint = ((massfract && V.numpoint > 1) || (DRYDEP && Gp.on)) ? P.npoint[s] : 1;
    xm0 = P.xmass1[s];
  }
};

// the dry deposition of one particle's step, handed back to a caller that scatters it where its wave is convergent again
// (k_pbl_finish: the neighbourhood sums of wave_kernel_add need every lane of the wave)
template <typename R>
struct DryDep {
  float dep[kMaxSpec];   // drydeposit(ks), timemanager.f90:650-656,690-693 (0: nothing)
  R x, y;                // the particle's position the kernel is centred on (xtra1, ytra1 after the step)
  int nage, kp, nclass;
};

template <typename R, bool DRYDEP, bool TURB = true>
__device__ __forceinline__ void epilogue_store(const View<R> &V, const GridP<R> &Gp, Parts<R> &P, long long s, int itime, int itramem,
                                               int nstop, const PState<R> &ps, const R *prob, Stats *st, const EpiPre<R> *pre = nullptr,
                                               DryDep<R> *defer = nullptr) {
  int itra1;
  if (nstop > 1) {
    itra1 = kDead;
    atomicAdd(&st->n_left, 1ull);
  } else {
    itra1 = itime + V.lsynctime;
    R xmassfract = (R)0;
    // release point of the particle: xmass(npoint(j),ks), npart(npoint(j)), timemanager.f90:663-666
    const bool massfract = V.mdomainfill == 0 && V.mquasilag == 0;
    // (with one release point every index into the point tables is clamped to it: npoint(j) is not needed, and the
    // table loads below do not wait for it)
    const int npoint = pre ? pre->npoint : ((massfract && V.numpoint > 1) || (DRYDEP && Gp.on)) ? P.npoint[s] : 1;
    const int kr = release_index(V, npoint);
    // everything the species loop reads -- the release point's table entries and the particle's masses -- in ONE round
    // trip (the asm ties the loads together; otherwise each is issued where it is first needed: three dependent trips)
    int npart_i = massfract ? V.rel_npart[kr] : 0;
    R xmr_all[kMaxSpec], xm_all[kMaxSpec];
#pragma unroll
    for (int ks = 0; ks < kMaxSpec; ks++) {
      xmr_all[ks] = (massfract && ks < V.nspec) ? V.rel_xmass[(size_t)ks * V.numpoint + kr] : (R)0;
      xm_all[ks] = ks < V.nspec ? ((pre && ks == 0) ? pre->xm0 : P.xmass1[(size_t)ks * P.cap + s]) : (R)0;
    }
    static_assert(kMaxSpec == 5, "the tie below names five species");
    asm volatile("" : "+v"(npart_i), "+v"(xmr_all[0]), "+v"(xmr_all[1]), "+v"(xmr_all[2]), "+v"(xmr_all[3]), "+v"(xmr_all[4]),
                      "+v"(xm_all[0]), "+v"(xm_all[1]), "+v"(xm_all[2]), "+v"(xm_all[3]), "+v"(xm_all[4]));
    const R npart_r = (R)npart_i;
#pragma unroll
    for (int ks = 0; ks < kMaxSpec; ks++) {
      if (ks < V.nspec) {
        R decfact = V.decay[ks] > (R)0 ? m_exp(-(R)abs(V.lsynctime) * V.decay[ks]) : (R)1;
        R xm = xm_all[ks];
        if (DRYDEP && V.drydepspec[ks]) {
          // timemanager.f90:650-656; drydeposit is real(dep_prec): 4 bytes in every build
          float drydeposit = (float)(xm * prob[ks] * decfact);
          xm = xm * ((R)1 - prob[ks]) * decfact;
          if (Gp.on && V.ldirect == 1) {   // timemanager.f90:690-696
            if (V.decay[ks] > (R)0) {
              const int ldeltat = itime < Gp.loutnext ? itime - (Gp.loutnext - Gp.loutstep) : itime - Gp.loutnext;   // :513-517
              drydeposit = (float)((R)drydeposit * m_exp((R)abs(ldeltat) * V.decay[ks]));
            }
            const int nage = ageclass(Gp, abs(itime - itramem));
            const int kp = Gp.ioutputforeachrelease == 1 ? npoint : 1;
            if (defer) {
              defer->dep[ks] = drydeposit; defer->x = (R)ps.xt; defer->y = (R)ps.yt; defer->nage = nage; defer->kp = kp; defer->nclass = P.nclass[s];
            } else {
              drydepo_particle(V, Gp, P.nclass[s], drydeposit, ks, (R)ps.xt, (R)ps.yt, nage, kp);
              if (Gp.nested) drydepo_particle(V, Gp, P.nclass[s], drydeposit, ks, (R)ps.xt, (R)ps.yt, nage, kp, true);   // timemanager.f90:694-696
            }
          }
        } else xm = xm * decfact;
        if (DRYDEP || V.decay[ks] > (R)0) P.xmass1[(size_t)ks * P.cap + s] = xm;
        if (massfract) {   // timemanager.f90:663-666
          const R xmr = xmr_all[ks];
          if (xmr > (R)0) xmassfract = m_max(xmassfract, npart_r * xm / xmr);
        } else {
          xmassfract = (R)1;
        }
      }
    }
    if (xmassfract < (R)0.0001) {   // minmass, par_mod.f90:213
      itra1 = kDead;
      atomicAdd(&st->n_minmass, 1ull);
    } else if (abs(itra1 - itramem) >= V.lage_last) {
      itra1 = kDead;
      atomicAdd(&st->n_maxage, 1ull);
    }
  }
  P.xt[s] = ps.xt; P.yt[s] = ps.yt; P.zt[s] = ps.zt;
  if (TURB) { P.up[s] = ps.up; P.vp[s] = ps.vp; P.cbt[s] = ps.icbt; }
  P.wp[s] = ps.wp;
  P.us[s] = ps.usigold; P.vs[s] = ps.vsigold; P.ws[s] = ps.wsigold;
  P.idt[s] = ps.ldt; P.itra1[s] = itra1;
}

// Per-slot hand-over record between the three PBL kernels: ONE contiguous, aligned record per particle (128 bytes in the
// fp64 build = one cache line, 64 bytes in f32), so that a lane that retires its particle in the Langevin kernel -- about
// two lanes of a wave per pass -- writes one line instead of a sector in each of sixteen arrays, and a refilling lane
// reads one.  k_prep fills the surface-layer part (FRESH), k_pbl_loop writes the particle's state at the end of its last
// pass (DONE / ESCAPED) -- or, when the launch's pass budget is used up (time slices, see k_pbl_loop), the state a later
// launch continues from (CONTINUE); k_pbl_finish reads it and writes the particle arrays, coalesced.
enum { PBL_FRESH = 3 };   // fourth value of the record's state next to PBL_CONTINUE / PBL_DONE / PBL_ESCAPED
template <typename R>
struct alignas(16 * sizeof(R)) PblRecord {
  // v[0..3]  = dxsave, dysave, dawsave, dcwsave                                  (CONTINUE, DONE, ESCAPED)
  // v[4..7]  = zt, up, vp, wp                                                    (CONTINUE, DONE, ESCAPED)
  // v[8..10] = ust, wst, ol (interpol_all.f90:80-107; ust as hanna.f90:43 floored it)  (FRESH, CONTINUE)
  //          | interpol_mod u, v, w of the last pass                             (DONE, ESCAPED)
  // v[11]    = CBL transition (cbl.f90:79-81), v[12] = mixing height of the cell (advance.f90:236-262): written by k_prep only
  // i[0] = nrand, i[1] = ldt, i[2] = state | icbt < 0 | indz | ngrid | |itimec - itime|  (pbl_pack)
  R v[13];
  int i[3];
};
static_assert(sizeof(PblRecord<double>) == 128 && sizeof(PblRecord<float>) == 64, "one cache line / half a line per record");
// state: bits 0-1, icbt = -1: bit 2, indz (<= 511): bits 3-11, ngrid + 2 (-2 .. kMaxNests): bits 12-14, |itimec - itime| (<= |lsynctime| <= 65535): bits 15-30
__device__ __forceinline__ int pbl_pack(int state, int icbt, int indz, int ngrid, int elapsed) {
  return state | (icbt < 0 ? 4 : 0) | (indz << 3) | ((ngrid + 2) << 12) | (elapsed << 15);
}
__device__ __forceinline__ int pbl_state(int pk) { return pk & 3; }
__device__ __forceinline__ int pbl_icbt(int pk) { return (pk & 4) ? -1 : 1; }
__device__ __forceinline__ int pbl_indz(int pk) { return (pk >> 3) & 511; }
__device__ __forceinline__ int pbl_ngrid(int pk) { return ((pk >> 12) & 7) - 2; }
__device__ __forceinline__ int pbl_elapsed(int pk) { return (pk >> 15) & 65535; }
static_assert(kMaxNz <= 512 && kMaxNests + 2 <= 7, "pbl_pack field widths");
template <typename R>
struct PblRec {
  PblRecord<R> *rec;       // [cap]
  R *tdep;                 // [cap], DRYDEP only: seconds of the step the particle spent below 2*href (advance.f90:582-599, see pbl_pass)
};

// Register budget: the steady-state kernel of a run on the mother lat-lon grid is built for FPX_PREP_WAVES waves per SIMD
// (<= 168 VGPRs, no scratch); the variants that also carry initialize(), the polar maps or the nest table would spill at
// that budget and keep two waves.
// The particle's step from the point where it is known to be due and inside the grid: initialize() if new, the boundary-layer
// test, the set-up of a boundary-layer particle or the whole step of one above it.  A function of its own so that k_prep can
// hold two instances: on a grid with polar caps (POLAR) the waves none of whose particles sits in a cap run the mother-grid
// instance (POLAR = false: compile-time ngrid = 0, field pointers in scalar registers, no stereographic code) with CAPCHECK
// (the test of advance.f90:843 still sees a particle that has moved INTO a cap); only the waves of the caps run the polar one.
template <typename R, bool DRYDEP, bool INIT, bool POLAR, bool NEST, bool CAPCHECK>
__device__ __forceinline__ void prep_body(const View<R> &V, const GridP<R> &Gp, Parts<R> &P, const SeqRng &S, const PblRec<R> &Q, long long s, int itime,
                                          unsigned int step, Stats *st, unsigned char *__restrict__ pbl_flag, const R *hgt,
                                          PState<R> &ps, int itramem, unsigned int pid) {
  constexpr bool MOTHER = !POLAR && !NEST;
  Rng<R> G;
  make_rng(V, pid, step, G);
  const bool is_new = INIT && ((itramem == itime) || (itime == 0));   // timemanager.f90:553
  if (is_new) {
    int nrand_i;
    R dcas = (R)0, dcas1 = (R)0;
    Rng<R> Gi = G;
    if (V.rng_mode == 0) {
      nrand_i = S.nrand_init[pid];
      if (sizeof(R) == 4) { dcas = (R)S.cbl_dcas[pid]; dcas1 = (R)S.cbl_dcas1[pid]; }
      else { dcas = (R)S.cbl_dcas_d[pid]; dcas1 = (R)S.cbl_dcas1_d[pid]; }
    } else {
      Gi.step = step | 0x80000000u;           // a stream of its own for initialize()
      nrand_i = V.rng_mode == 1 ? Gi.start_index(0) : 1;
      dcas = (R)(((float)(Gi.bits(1) >> 8) + 0.5f) * (1.0f / 16777216.0f));
      Rng<R> Gn = Gi;
      Gn.mode = 2;
      dcas1 = Gn.at(7);
    }
    initialize_particle(V, hgt, Gi, nrand_i, itime, ps, dcas, dcas1);
    atomicAdd(&st->n_init, 1ull);
  }

  AdvCtx<R> A;
  const TimeW<R> W = time_weights(V, itime);
  const bool in_pbl = adv_begin<R, MOTHER, true>(V, ps.xt, ps.yt, ps.zt, itime, advance_start_index(V, S, G, pid), A);
  if (V.lsettling) A.nsp = settling_species(V, P.npoint[s]);   // advance.f90:518-524 with nrelpoint = npoint(j)
  if (in_pbl) {
    if (is_new) {   // the PBL kernels read the state from HBM (an old particle's is there already)
      P.up[s] = ps.up; P.vp[s] = ps.vp; P.wp[s] = ps.wp;
      P.us[s] = ps.usigold; P.vs[s] = ps.vsigold; P.ws[s] = ps.wsigold;
      P.idt[s] = ps.ldt; P.cbt[s] = ps.icbt;
    }
    // first-pass set-up of interpol_all (ust, wst, ol: interpol_all.f90:80-107) done here, where
    // the lanes are convergent; the Langevin kernel then starts from five numbers per particle
    PblCtx<R> B;
    pbl_begin(V, ps.xt, ps.yt, W, A, B);
    {
      PblRecord<R> &r = Q.rec[s];
      r.v[8] = B.ust; r.v[9] = B.wst; r.v[10] = B.ol; r.v[11] = B.transition;
      r.v[12] = A.h;
      r.i[0] = A.nrand; r.i[2] = pbl_pack(PBL_FRESH, 1, 0, A.ngrid, 0);
    }
    // Regime class of the particle's PBL passes (hanna.f90:42,59,91 and advance.f90:405-406).  The
    // work list is the slots stably sorted by this 3-bit key: class by class, each class in slot
    // (= cell) order, so the lanes of a wave run the same branch of the turbulence scheme.
    unsigned char cls;
    if (V.cblflag == 1 && V.turbswitch && (-A.h / B.ol > (R)5)) cls = 1;      // skewed CBL scheme
    else if (A.h / m_abs(B.ol) < (R)1) cls = 3;                               // neutral
    else if (B.ol < (R)0) cls = 2;                                            // unstable, Gaussian: next to the CBL class, whose
                                                                              // hanna_short branch it shares (waves at a class boundary run both classes)
    else cls = 4;                                                             // stable
    // (measured and not kept: the reverse order -- 297.3 instead of 293.7 ms at 1e8, 47.9 instead of 42.2 ms at an eighth
    // of the cloud: the launch must not END on the CBL class, whose particles make the most passes; a cursor per class
    // with the waves dealt to the classes by their work -- 46.7 ms at an eighth: every class then ends at the end of the
    // launch; a cost bucket (octaves of (ustar+wstar)/h) below the class in the key: no measurable change)
    // Below the class: a cost bucket, the most expensive first.  The number of passes a particle makes is about |lsynctime| /
    // (its time step); its last time step of the previous step (idt) predicts that well where the time step is set by the
    // cell (ust, ol, h: the stable class keeps one value for the whole step).  With the long particles of a class at the head
    // of its segment, what the waves still hold when the cursor leaves the class -- and at the end of the launch -- are short
    // particles: less of the launch runs with half-empty or two-class waves.  Each bucket stays in slot (= cell) order.
    unsigned char bucket = 0;
    if (V.pbl_cost_buckets) {
      const int idt = is_new ? ps.ldt : P.idt[s];
      const int est = abs(V.lsynctime) / max(idt, 1);          // passes, if the time step stayed
      if (V.pbl_cost_buckets == 1) bucket = est >= 128 ? 3 : est >= 64 ? 2 : est >= 32 ? 1 : 0;
      else if (V.pbl_cost_buckets == 2) bucket = est >= 64 ? 1 : 0;
      else bucket = est >= 256 ? 7 : est >= 128 ? 6 : est >= 96 ? 5 : est >= 64 ? 4 : est >= 48 ? 3 : est >= 32 ? 2 : est >= 16 ? 1 : 0;
    }
    pbl_flag[s] = (unsigned char)((cls - 1) * 8 + (7 - bucket));
    return;
  }
  // Above the mixing layer for the whole step (advance.f90:629-708 -> 99): wp and ldt are set, not read; up, vp and cbt
  // are not touched; the mesoscale velocities are fetched with the last column of the 48-value gather (same round
  // trip, not live across the gather: the asm ties the loads to that point of the program).
  R usig, vsig, wsig;
  pbl_flag[s] = kKeyDone;
  auto late = [&]() {
    if (!is_new) {
      asm volatile("" : "+v"(s));
      ps.usigold = P.us[s]; ps.vsigold = P.vs[s]; ps.wsigold = P.ws[s];
    }
  };
  above_step<R, Rng<R>, true>(V, hgt, G, W, itime, ps.xt, ps.yt, ps.zt, ps.wp, ps.ldt, A, usig, vsig, wsig, late);
  // (fetching the epilogue's npoint / xmass1 with the Petterssen gather was tried here: the three registers spill at the
  // three-wave budget of this kernel; k_pbl_finish has them)
  const int nstop = adv_finish<R, Rng<R>, POLAR, MOTHER, NoLate, CAPCHECK>(V, hgt, G, itime, ps, A, usig, vsig, wsig);
  R prob[kMaxSpec];
#pragma unroll
  for (int ks = 0; ks < kMaxSpec; ks++) prob[ks] = (R)0;
  if (is_new) epilogue_store<R, DRYDEP, true>(V, Gp, P, s, itime, itramem, nstop, ps, prob, st);
  else epilogue_store<R, DRYDEP, false>(V, Gp, P, s, itime, itramem, nstop, ps, prob, st);
}

template <typename R, bool DRYDEP, bool INIT, bool POLAR, bool NEST>
__global__ void __launch_bounds__(kBlock, FPX_PREP_TWO_WAVES(INIT, POLAR, NEST, DRYDEP) ? 2 : FPX_PREP_WAVES) k_prep(View<R> V_arg, GridP<R> Gp_arg, Parts<R> P_arg, SeqRng S_arg, PblRec<R> Q_arg, long long numpart, int itime,
                                                 unsigned int step, Stats *st, unsigned char *__restrict__ pbl_flag,
                                                 unsigned int *__restrict__ pbl_count) {
#ifndef FPX_VIEW_BY_VALUE
  struct KArgs { View<R> V; GridP<R> Gp; Parts<R> P; SeqRng S; PblRec<R> Q; };   // the leading parameters as the segment holds them
  const KArgs &ka = *(const KArgs *)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
  const View<R> &V = ka.V; const GridP<R> &Gp = ka.Gp; Parts<R> &P = const_cast<Parts<R> &>(ka.P); const SeqRng &S = ka.S; const PblRec<R> &Q = ka.Q;
  (void)V_arg; (void)Gp_arg; (void)P_arg; (void)S_arg; (void)Q_arg;
#else
  const View<R> &V = V_arg; const GridP<R> &Gp = Gp_arg; Parts<R> &P = P_arg; const SeqRng &S = S_arg; const PblRec<R> &Q = Q_arg;
#endif
  __shared__ R hgt[kMaxNz];
  for (int k = threadIdx.x; k < V.nz; k += blockDim.x) hgt[k] = V.height[k];
  __syncthreads();
  long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= numpart) return;
  // the position travels in the same memory round trip as the due test (nearly every particle is due)
  PState<R> ps;
  const int itra1_in = P.itra1[s];
  ps.xt = P.xt[s]; ps.yt = P.yt[s]; ps.zt = P.zt[s];
  int itramem = P.itramem[s];
  unsigned int pid = P.pid[s];
  // ONE round trip: without this the compiler sinks each load into the branch that first needs it (itra1 -> wait -> xt ->
  // wait -> yt -> wait -> zt ...: five dependent round trips at the head of a latency-bound kernel)
  {
    int due_key = itra1_in;
    asm volatile("" : "+v"(due_key), "+v"(ps.xt), "+v"(ps.yt), "+v"(ps.zt), "+v"(itramem), "+v"(pid));
    if (due_key != itime) { pbl_flag[s] = kKeyNotDue; return; }    // timemanager.f90:537
  }
  // key kKeyNotDue = not due; kKeyDone = due, finished in this kernel (above the PBL); 8 (c - 1) + 0..7 = PBL particle of regime
  // class c = 1..4, cost bucket 7..0 (see below).
  // The counts (particles due, length of the PBL work list) are read off the sorted keys by
  // k_list_counts: one atomic per wave on a single address costs more than the whole kernel.

  // a non-finite or out-of-grid position would index outside the fields (the reference
  // would read arbitrary memory): terminate the particle instead
  if (!(ps.xt >= 0. && ps.xt <= (double)V.nxmin1 && ps.yt >= 0. && ps.yt <= (double)V.nymin1) || !(ps.zt == ps.zt)) {
    P.itra1[s] = kDead;
    pbl_flag[s] = kKeyDone;
    atomicAdd(&st->n_badpos, 1ull);
    return;
  }

  // wave-uniform: is any particle of this wave new (timemanager.f90:553)?  The INIT instance runs whenever particles MAY have
  // been released since the last step -- every step of a run with a continuous release -- but only the waves that do hold a
  // new particle need the body with initialize() in it (a dynamically indexed array in scratch, more registers); the others
  // run the steady-state body, so that the instance costs what the steady-state kernel costs.
  if (INIT && __any((itramem == itime) || (itime == 0))) {
    prep_body<R, DRYDEP, true, POLAR, NEST, false>(V, Gp, P, S, Q, s, itime, step, st, pbl_flag, hgt, ps, itramem, pid);
    return;
  }
  if (POLAR && !NEST) {
    // wave-uniform: does any particle of this wave start in a polar cap (advance.f90:161-164)?  The slots are cell-sorted, so
    // five waves in six of a global run do not
    if (!__any(pick_polar(V, ps.yt) != 0)) {
      prep_body<R, DRYDEP, false, false, false, true>(V, Gp, P, S, Q, s, itime, step, st, pbl_flag, hgt, ps, itramem, pid);
      return;
    }
  }
  prep_body<R, DRYDEP, false, POLAR, NEST, false>(V, Gp, P, S, Q, s, itime, step, st, pbl_flag, hgt, ps, itramem, pid);
}

// ---------------------------------------------------------------------------
// partoutput.f90:63-190 -- the binary particle dump.  The device builds the
// record stream of the file (Fortran sequential unformatted: 4-byte length, payload, 4-byte
// length) for the particles due at itime, in particle-number order, in the host's real kind H;
// the host only streams the bytes to disk.  Arithmetic in H with FMA contraction off: the file
// is byte-identical to the reference's.
// ---------------------------------------------------------------------------
template <typename H>
struct DiagP {
  const H *oro, *tropo[2];   // host layout (ix,jy), stride nxmax
  const H *d3;               // [jy][ix][iz][slot][3] = (pv, qv, tt): z fastest like the wind pack, one 96-byte run per corner column
  int nxmax, nymax;
  H dx, dy, xlon0, ylat0;
};

// Both kernels run over the device slots (cell-sorted after a locality sort: coalesced state reads, gathers with
// the locality of the particle step); the file order is the particle number, so the selection flag and the
// record position are indexed by P.pid.
template <typename R>
__global__ void k_po_flags(Parts<R> P, long long n, int itime, unsigned int *__restrict__ flags) {
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  flags[P.pid[s]] = P.itra1[s] == itime ? 1u : 0u;
}

__device__ __forceinline__ unsigned int *po_put(unsigned int *w, float v) { *w = __float_as_uint(v); return w + 1; }
__device__ __forceinline__ unsigned int *po_put(unsigned int *w, double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  w[0] = (unsigned int)b; w[1] = (unsigned int)(b >> 32);
  return w + 2;
}

template <typename R, typename H>
__global__ void __launch_bounds__(kBlock) k_partoutput(View<R> V, Parts<R> P, DiagP<H> D, const unsigned int *__restrict__ recidx,
                                                       long long n, int itime, unsigned int *__restrict__ out) {
#pragma clang fp contract(off)
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n || P.itra1[s] != itime) return;
  const unsigned int pid = P.pid[s];
  const int nx = V.nx, ny = V.ny, nz = V.nz, nspec = V.nspec;
  const int reclen = 8 + (10 + nspec) * (int)sizeof(H);
  const H dt1 = (H)(itime - V.memtime0), dt2 = (H)(V.memtime1 - itime);   // partoutput.f90:69-71
  const H dtt = (H)1. / (dt1 + dt2);
  const double xt = P.xt[s], yt = P.yt[s];
  const H zt = (H)P.zt[s];
  const H xlon = (H)((double)D.xlon0 + xt * (double)D.dx);
  const H ylat = (H)((double)D.ylat0 + yt * (double)D.dy);
  const int ix = (int)xt, jy = (int)yt;
  int ixp = ix + 1, jyp = jy + 1;
  const H ddx = (H)(xt - (double)(H)ix), ddy = (H)(yt - (double)(H)jy);
  const H rddx = (H)1. - ddx, rddy = (H)1. - ddy;
  const H p1 = rddx * rddy, p2 = ddx * rddy, p3 = rddx * ddy, p4 = ddx * ddy;
  if (jyp >= D.nymax) jyp = jyp - 1;                                      // :119-121
  if (ixp >= D.nxmax) ixp = D.nxmax - 1;                                  // guard (weight 0 there)
  auto h2 = [&](const H *f, int i, int j) { return f[(size_t)i + (size_t)D.nxmax * (size_t)j]; };
  // component c of (pv, qv, tt) at level k, slot h; elements of the host's padding read as 0 like rho below
  auto d3 = [&](int c, int i, int j, int k, int h) -> H {
    if (i >= nx || j >= ny) return (H)0;
    return D.d3[((((size_t)j * nx + i) * nz + (k - 1)) * 2 + h) * 3 + c];
  };
  // rho and hmix live in the gather packs (compact nx, ny; elements of the host's padding read as 0)
  auto rho_at = [&](int i, int j, int k, int slot) -> H {
    if (i >= nx || j >= ny) return (H)0;
    return (H)V.r2[(((size_t)j * nx + i) * nz + (k - 1)) * 4 + slot * 2];
  };
  auto hmix_at = [&](int i, int j, int slot) -> H {
    if (i >= nx || j >= ny) return (H)0;
    return (H)V.sfc[((size_t)j * nx + i) * 8 + slot * 4 + 3];
  };
  const H topo = p1 * h2(D.oro, ix, jy) + p2 * h2(D.oro, ixp, jy) + p3 * h2(D.oro, ix, jyp) + p4 * h2(D.oro, ixp, jyp);
  int indz = nz - 1, indzp = nz;   // the reference keeps the previous particle's indices when zt >= height(nz); cannot happen after advance()
  for (int il = 2; il <= nz; il++)
    if ((H)V.height[il - 1] > zt) { indz = il - 1; indzp = il; break; }
  const H dz1 = zt - (H)V.height[indz - 1], dz2 = (H)V.height[indzp - 1] - zt;
  const H dz = (H)1. / (dz1 + dz2);
  const int slot[2] = {V.m1, V.m2};
  H pvprof[2], qvprof[2], ttprof[2], rhoprof[2];
#pragma unroll
  for (int l = 0; l < 2; l++) {
    const int ind = indz + l;
    H pv1[2], qv1[2], tt1[2], rho1[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      const int h = slot[m];
      pv1[m] = p1 * d3(0, ix, jy, ind, h) + p2 * d3(0, ixp, jy, ind, h) + p3 * d3(0, ix, jyp, ind, h) + p4 * d3(0, ixp, jyp, ind, h);
      qv1[m] = p1 * d3(1, ix, jy, ind, h) + p2 * d3(1, ixp, jy, ind, h) + p3 * d3(1, ix, jyp, ind, h) + p4 * d3(1, ixp, jyp, ind, h);
      tt1[m] = p1 * d3(2, ix, jy, ind, h) + p2 * d3(2, ixp, jy, ind, h) + p3 * d3(2, ix, jyp, ind, h) + p4 * d3(2, ixp, jyp, ind, h);
      rho1[m] = p1 * rho_at(ix, jy, ind, h) + p2 * rho_at(ixp, jy, ind, h) + p3 * rho_at(ix, jyp, ind, h) + p4 * rho_at(ixp, jyp, ind, h);
    }
    pvprof[l] = (pv1[0] * dt2 + pv1[1] * dt1) * dtt;
    qvprof[l] = (qv1[0] * dt2 + qv1[1] * dt1) * dtt;
    ttprof[l] = (tt1[0] * dt2 + tt1[1] * dt1) * dtt;
    rhoprof[l] = (rho1[0] * dt2 + rho1[1] * dt1) * dtt;
  }
  const H pvi = (dz1 * pvprof[1] + dz2 * pvprof[0]) * dz;
  const H qvi = (dz1 * qvprof[1] + dz2 * qvprof[0]) * dz;
  const H tti = (dz1 * ttprof[1] + dz2 * ttprof[0]) * dz;
  const H rhoi = (dz1 * rhoprof[1] + dz2 * rhoprof[0]) * dz;
  H tr[2], hm[2];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const int h = slot[m];
    tr[m] = p1 * h2(D.tropo[h], ix, jy) + p2 * h2(D.tropo[h], ixp, jy) + p3 * h2(D.tropo[h], ix, jyp) + p4 * h2(D.tropo[h], ixp, jyp);
    hm[m] = p1 * hmix_at(ix, jy, h) + p2 * hmix_at(ixp, jy, h) + p3 * hmix_at(ix, jyp, h) + p4 * hmix_at(ixp, jyp, h);
  }
  const H hmixi = (hm[0] * dt2 + hm[1] * dt1) * dtt;
  const H tri = (tr[0] * dt2 + tr[1] * dt1) * dtt;
  // the record, :177-179 (32-bit words: the payload of an 8-byte build is only 4-byte aligned in the file)
  unsigned int *w = out + (size_t)recidx[pid] * (size_t)(reclen / 4 + 2);
  *w++ = (unsigned int)reclen;
  *w++ = (unsigned int)P.npoint[s];
  w = po_put(w, xlon); w = po_put(w, ylat); w = po_put(w, zt);
  *w++ = (unsigned int)P.itramem[s];
  w = po_put(w, topo); w = po_put(w, pvi); w = po_put(w, qvi); w = po_put(w, rhoi);
  w = po_put(w, hmixi); w = po_put(w, tri); w = po_put(w, tti);
  for (int ks = 0; ks < nspec; ks++) w = po_put(w, (H)P.xmass1[(size_t)ks * P.cap + s]);
  *w++ = (unsigned int)reclen;
}

// ---------------------------------------------------------------------------
// releaseparticles.f90:133-375 and the splitting block timemanager.f90:473-504
// ---------------------------------------------------------------------------
// per release point and call, prepared by the host in the host's real kind H (releaseparticles.f90:63-131)
template <typename H>
struct RelPoints {
  const long long *first;    // [numpoint+1] exclusive prefix of numrel: particle k of this call belongs to point i with first[i] <= k < first[i+1]
  const long long *gfirst;   // [numpoint] number, in the release count of the whole run over ALL ranks, of this rank's first particle of point i
  const H *xp1, *xaux, *yp1, *yaux, *zp1, *zaux;   // [numpoint]
  const H *mass;             // [nspec][numpoint]: xmass(i,k)/real(npart(i))*timecorrect(k)/average_timecorrect
  const short *kindz;        // [numpoint]
  int numpoint;
};

// free[pid] = the storage space of particle number pid+1 is vacant: itra1 /= itime (:135)
template <typename R>
__global__ void k_rel_flags(Parts<R> P, const unsigned int *__restrict__ slot_of_pid, long long cap, int itime, unsigned int *__restrict__ flags) {
  const long long pid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= cap) return;
  const long long s = slot_of_pid ? (long long)slot_of_pid[pid] : pid;
  flags[pid] = P.itra1[s] != itime ? 1u : 0u;
}
// target[k] = particle number (0-based) of the k-th vacant storage space
__global__ void k_rel_targets(const unsigned int *__restrict__ flags, const unsigned int *__restrict__ rank, long long cap, long long ntotal,
                              unsigned int *__restrict__ target) {
  const long long pid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= cap || !flags[pid]) return;
  const unsigned int r = rank[pid];
  if ((long long)r < ntotal) target[r] = (unsigned int)pid;
}

// a nested wind field as releaseparticles reads it (:196-226,231-341): orography and tt of time slot 2 in the host's real
// kind, compact; rho of slot 2 from the nest's gather pack
template <typename R, typename H>
struct RelNests {
  int n;
  H eps;                              // nxmax/3.e5 (:61)
  struct { int nx, ny; H xl, yl, xr, yr, xres, yres; const H *oro, *tt2; const R *r2; } g[kMaxNests];
};

template <typename R, typename H>
__global__ void __launch_bounds__(kBlock) k_release(View<R> V, Parts<R> P, DiagP<H> D, RelNests<R, H> NS, RelPoints<H> RP, const unsigned int *__restrict__ target,
                                                    const unsigned int *__restrict__ slot_of_pid, long long ntotal, int itime,
                                                    const H *__restrict__ uniforms /* [4][ntotal] serial ran1 stream, or NULL: counter RNG */,
                                                    int numparticlecount0, int mintime, int itsplit, int ind_rel, int nclassunc, int mquasilag,
                                                    H *__restrict__ rho_rel, unsigned int *__restrict__ maxpid) {
#pragma clang fp contract(off)
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ntotal) return;
  // release point of particle k: last i with first[i] <= k
  int lo = 0, hi = RP.numpoint;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (RP.first[mid] <= k) lo = mid; else hi = mid; }
  const int i = lo;
  const unsigned int pid = target[k];
  const long long s = slot_of_pid ? (long long)slot_of_pid[pid] : (long long)pid;
  H u[4];
  if (uniforms) {
#pragma unroll
    for (int d = 0; d < 4; d++) u[d] = uniforms[(size_t)d * ntotal + k];
  } else {   // four uniforms in [0,1) keyed on (seed, number of the particle in the run's release count over all ranks)
    unsigned int o[4];
    const unsigned long long gk = (unsigned long long)(RP.gfirst[i] + (k - RP.first[i]));
    philox4x32((unsigned int)gk, 0x52454c45u /* "RELE" */, (unsigned int)(gk >> 32), 0u,
               (unsigned int)V.seed, (unsigned int)(V.seed >> 32), o);
#pragma unroll
    for (int d = 0; d < 4; d++) u[d] = (H)((float)(o[d] >> 8) * (1.0f / 16777216.0f));
  }
  const int nx = V.nx, ny = V.ny, nz = V.nz;
  double xt = (double)(RP.xp1[i] + u[0] * RP.xaux[i]);                       // :139
  if (V.xglobal) {
    if (xt > (double)(H)V.nxmin1) xt = xt - (double)(H)V.nxmin1;
    if (xt < 0.) xt = xt + (double)(H)V.nxmin1;
  }
  const double yt = (double)(RP.yp1[i] + u[1] * RP.yaux[i]);                 // :146
  int nc = (int)(u[2] * (H)nclassunc) + 1;                                   // :168-169
  nc = nc < nclassunc ? nc : nclassunc;
  H zt = RP.zp1[i] + u[3] * RP.zaux[i];                                      // :183
  // the nest we are in (:196-206), grid coordinates and weights (:211-226)
  int ngrid = 0;
  for (int q = NS.n; q >= 1; q--)
    if (xt > (double)(NS.g[q - 1].xl + NS.eps) && xt < (double)(NS.g[q - 1].xr - NS.eps) && yt > (double)(NS.g[q - 1].yl + NS.eps) &&
        yt < (double)(NS.g[q - 1].yr - NS.eps)) { ngrid = q; break; }
  int ix, jy, gnx = nx, gny = ny, gnxmax = D.nxmax, gnymax = D.nymax;
  H ddx, ddy;
  const H *g_oro = D.oro, *g_tt = nullptr;
  const R *g_r2 = V.r2;
  if (ngrid > 0) {
    const auto &N = NS.g[ngrid - 1];
    const H xtn = (H)((xt - (double)N.xl) * (double)N.xres), ytn = (H)((yt - (double)N.yl) * (double)N.yres);
    ix = (int)xtn; jy = (int)ytn;
    ddy = ytn - (H)jy; ddx = xtn - (H)ix;
    gnx = N.nx; gny = N.ny; gnxmax = N.nx + 1; gnymax = N.ny + 1;      // one padding column / row reads as 0
    g_oro = N.oro; g_tt = N.tt2; g_r2 = N.r2;
  } else {
    ix = (int)xt; jy = (int)yt;
    ddy = (H)(yt - (double)(H)jy); ddx = (H)(xt - (double)(H)ix);
  }
  int ixp = ix + 1, jyp = jy + 1;
  // guards (the reference would read outside its arrays): a position on the upper edges
  ix = min(max(ix, 0), gnx - 1); jy = min(max(jy, 0), gny - 1); ixp = min(max(ixp, 0), gnxmax - 1); jyp = min(max(jyp, 0), gnymax - 1);
  const H rddx = (H)1. - ddx, rddy = (H)1. - ddy;
  const H p1 = rddx * rddy, p2 = ddx * rddy, p3 = rddx * ddy, p4 = ddx * ddy;
  // mother: host layout with stride nxmax; nest: compact [nyn][nxn], the padding reads as 0
  auto oro = [&](int a, int b) -> H {
    if (ngrid > 0) return (a >= gnx || b >= gny) ? (H)0 : g_oro[(size_t)a + (size_t)gnx * (size_t)b];
    return g_oro[(size_t)a + (size_t)D.nxmax * (size_t)b];
  };
  const H topo = p1 * oro(ix, jy) + p2 * oro(ixp, jy) + p3 * oro(ix, jyp) + p4 * oro(ixp, jyp);
  // rho(.,.,kz,2): the r2 pack, physical slot 2; tt(.,.,kz,2): component 2 of the d3 pack (mother) / the nest's compact array
  auto rho2 = [&](int a, int b, int kz) -> H { return (a >= gnx || b >= gny) ? (H)0 : (H)g_r2[(((size_t)b * gnx + a) * nz + (kz - 1)) * 4 + 2]; };
  auto tt2 = [&](int a, int b, int kz) -> H {
    if (a >= gnx || b >= gny) return (H)0;
    if (ngrid > 0) return g_tt[(size_t)a + (size_t)gnx * ((size_t)b + (size_t)gny * (size_t)(kz - 1))];
    return D.d3[((((size_t)b * nx + a) * nz + (kz - 1)) * 2 + 1) * 3 + 2];
  };
  const int kz3 = RP.kindz[i];
  if (kz3 == 3) {                                                            // :231-273
    const H presspart = zt;
    H pressold = (H)0;
    for (int kz = 1; kz <= nz; kz++) {
      const H r = p1 * rho2(ix, jy, kz) + p2 * rho2(ixp, jy, kz) + p3 * rho2(ix, jyp, kz) + p4 * rho2(ixp, jyp, kz);
      const H t = p1 * tt2(ix, jy, kz) + p2 * tt2(ixp, jy, kz) + p3 * tt2(ix, jyp, kz) + p4 * tt2(ixp, jyp, kz);
      const H press = r * (H)287.05 * t / (H)100.;
      if (kz == 1) pressold = press;
      if (press < presspart) {
        if (kz == 1) zt = (H)V.height[0] / (H)2.;
        else {
          const H dz1 = pressold - presspart, dz2 = presspart - press;
          zt = ((H)V.height[kz - 2] * dz2 + (H)V.height[kz - 1] * dz1) / (dz1 + dz2);
        }
        break;
      }
      pressold = press;
    }
  }
  if (kz3 == 2) zt = zt - topo;                                              // :278
  if (zt < (H)1.e-6) zt = (H)1.e-6;
  if (zt > (H)V.height[nz - 1] - (H)0.5) zt = (H)V.height[nz - 1] - (H)0.5;
  H rhoout = (H)1.;
  const bool dens = ind_rel == 1 || ind_rel == 3 || ind_rel == 4;
  if (dens) {                                                                // :300-341
    int indz = nz - 1, indzp = nz;
    for (int ii = 2; ii <= nz; ii++)
      if ((H)V.height[ii - 1] > zt) { indz = ii - 1; indzp = ii; break; }
    const H dz1 = zt - (H)V.height[indz - 1], dz2 = (H)V.height[indzp - 1] - zt, dz = (H)1. / (dz1 + dz2);
    H rhoaux[2];
#pragma unroll
    for (int n = 0; n < 2; n++)
      rhoaux[n] = p1 * rho2(ix, jy, indz + n) + p2 * rho2(ixp, jy, indz + n) + p3 * rho2(ix, jyp, indz + n) + p4 * rho2(ixp, jyp, indz + n);
    rhoout = (dz2 * rhoaux[0] + dz1 * rhoaux[1]) * dz;
    if (rho_rel && k == RP.first[i + 1] - 1) rho_rel[i] = rhoout;            // rho_rel(i) keeps the last particle's value
  }
  for (int ks = 0; ks < V.nspec; ks++) {
    H m = RP.mass[(size_t)ks * RP.numpoint + i];
    if (dens) m = m * rhoout;
    P.xmass1[(size_t)ks * P.cap + s] = (R)m;
    if (P.xscav) P.xscav[(size_t)ks * P.cap + s] = (R)-1;      // releaseparticles.f90:167-171: not yet scavenged
  }
  P.xt[s] = xt; P.yt[s] = yt; P.zt[s] = (R)zt;
  P.nclass[s] = nc;
  P.npoint[s] = mquasilag == 0 ? i + 1 : (int)(numparticlecount0 + k + 1);   // :171-175
  P.idt[s] = mintime; P.itra1[s] = itime; P.itramem[s] = itime;
  P.itrasplit[s] = itime + V.ldirect * itsplit;
  // the turbulent state of the storage space is whatever its last owner left; initialize() overwrites it
  // before the first advance() (timemanager.f90:553), as in the reference
  atomicMax(maxpid, pid);
}

// splitting, timemanager.f90:473-504
template <typename R>
__global__ void k_split_flags(Parts<R> P, const unsigned int *__restrict__ slot_of_pid, long long numpart, int itime, int ldirect,
                              unsigned int *__restrict__ flags) {
  const long long pid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= numpart) return;
  const long long s = slot_of_pid ? (long long)slot_of_pid[pid] : pid;
  flags[pid] = (ldirect * itime >= ldirect * P.itrasplit[s]) ? 1u : 0u;
}
template <typename R>
__global__ void k_split(Parts<R> P, const unsigned int *__restrict__ slot_of_pid, const unsigned int *__restrict__ flags,
                        const unsigned int *__restrict__ rank, long long numpart, long long room, int nspec) {
#pragma clang fp contract(off)
  const long long pid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= numpart || !flags[pid] || (long long)rank[pid] >= room) return;
  const long long j = slot_of_pid ? (long long)slot_of_pid[pid] : pid;
  const long long n = numpart + rank[pid];          // storage spaces behind numpart are identity-numbered
  const int itm = P.itramem[j];
  // itrasplit(j) = 2*(itrasplit(j)-itramem(j))+itramem(j), timemanager.f90:486 -- in 64 bits, saturated at the "never" value
  // (the reference's default integer would overflow once the doubled interval passes 2^31 s)
  long long its64 = 2ll * ((long long)P.itrasplit[j] - itm) + itm;
  its64 = its64 > 999999999ll ? 999999999ll : (its64 < -999999999ll ? -999999999ll : its64);
  const int its = (int)its64;
  P.itrasplit[j] = its; P.itrasplit[n] = its;
  P.itramem[n] = itm; P.itra1[n] = P.itra1[j]; P.idt[n] = P.idt[j]; P.npoint[n] = P.npoint[j]; P.nclass[n] = P.nclass[j];
  P.xt[n] = P.xt[j]; P.yt[n] = P.yt[j]; P.zt[n] = P.zt[j];
  P.up[n] = P.up[j]; P.vp[n] = P.vp[j]; P.wp[n] = P.wp[j];
  P.us[n] = P.us[j]; P.vs[n] = P.vs[j]; P.ws[n] = P.ws[j];
  P.cbt[n] = P.cbt[j];
  for (int ks = 0; ks < nspec; ks++) {
    const R m = P.xmass1[(size_t)ks * P.cap + j] / (R)2;
    P.xmass1[(size_t)ks * P.cap + j] = m;
    P.xmass1[(size_t)ks * P.cap + n] = m;
    if (P.xscav) P.xscav[(size_t)ks * P.cap + n] = P.xscav[(size_t)ks * P.cap + j];   // the copy inherits the receptor's scavenged fraction
  }
}

// Particle redistribution between ranks, mpi_mod.f90:661-856 (mpif_redist_part).  The message is what the reference sends,
// in its order, as ONE buffer: nclass, npoint, itra1, idt, itramem, itrasplit (int32[n] each), xtra1, ytra1 (f64[n]), ztra1
// (R[n]), xmass1 (R[nspec][n]).  As in the reference the turbulent velocities, cbt and xscav_frac1 do NOT travel.
template <typename R>
struct RedistBuf {
  int *nclass, *npoint, *itra1, *idt, *itramem, *itrasplit;
  double *xt, *yt;
  R *zt, *xmass;   // xmass: [nspec][n]
  long long n;
  __host__ __device__ static size_t bytes(long long n, int nspec) { return (size_t)n * (6 * 4 + 2 * 8 + (size_t)(1 + nspec) * sizeof(R)); }
  __host__ __device__ static RedistBuf at(void *base, long long n, int nspec) {
    RedistBuf b;
    unsigned char *q = (unsigned char *)base;
    b.xt = (double *)q; q += (size_t)n * 8;            // the 8-byte arrays first: every array stays aligned for any n
    b.yt = (double *)q; q += (size_t)n * 8;
    b.zt = (R *)q; q += (size_t)n * sizeof(R);
    b.xmass = (R *)q; q += (size_t)n * nspec * sizeof(R);
    b.nclass = (int *)q; q += (size_t)n * 4;
    b.npoint = (int *)q; q += (size_t)n * 4;
    b.itra1 = (int *)q; q += (size_t)n * 4;
    b.idt = (int *)q; q += (size_t)n * 4;
    b.itramem = (int *)q; q += (size_t)n * 4;
    b.itrasplit = (int *)q;
    b.n = n;
    return b;
  }
};
// sender, :700-745: the storage spaces ll..ul = numpart-num_trans+1 .. numpart go into the message and are terminated
template <typename R>
__global__ void k_redist_pack(Parts<R> P, const unsigned int *__restrict__ slot_of_pid, long long first_pid, RedistBuf<R> B, int nspec) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B.n) return;
  const long long pid = first_pid + i;
  const long long s = slot_of_pid ? (long long)slot_of_pid[pid] : pid;
  B.nclass[i] = P.nclass[s]; B.npoint[i] = P.npoint[s]; B.itra1[i] = P.itra1[s]; B.idt[i] = P.idt[s];
  B.itramem[i] = P.itramem[s]; B.itrasplit[i] = P.itrasplit[s];
  B.xt[i] = P.xt[s]; B.yt[i] = P.yt[s]; B.zt[i] = P.zt[s];
  for (int ks = 0; ks < nspec; ks++) B.xmass[(size_t)ks * B.n + i] = P.xmass1[(size_t)ks * P.cap + s];
  P.itra1[s] = kDead;                                  // :744
}
// receiver, :808-835: valid[i] = the i-th received particle is alive at itime ("we may have transferred invalid particles")
__global__ void k_redist_valid(const int *__restrict__ itra1_tmp, long long n, int itime, unsigned int *__restrict__ valid) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) valid[i] = itra1_tmp[i] == itime ? 1u : 0u;
}
// the k-th valid received particle goes into the k-th storage space of 1..maxnumpart with itra1 /= itime (target[k], as
// k_rel_targets builds it); everything else of that space stays what its last owner left
template <typename R>
__global__ void k_redist_unpack(Parts<R> P, const unsigned int *__restrict__ slot_of_pid, RedistBuf<R> B, int nspec, const unsigned int *__restrict__ valid,
                                const unsigned int *__restrict__ vrank, const unsigned int *__restrict__ target, unsigned int *__restrict__ maxpid) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B.n || !valid[i]) return;
  const unsigned int pid = target[vrank[i]];
  const long long s = slot_of_pid ? (long long)slot_of_pid[pid] : (long long)pid;
  P.itra1[s] = B.itra1[i]; P.npoint[s] = B.npoint[i]; P.nclass[s] = B.nclass[i]; P.idt[s] = B.idt[i];
  P.itramem[s] = B.itramem[i]; P.itrasplit[s] = B.itrasplit[i];
  P.xt[s] = B.xt[i]; P.yt[s] = B.yt[i]; P.zt[s] = B.zt[i];
  for (int ks = 0; ks < nspec; ks++) P.xmass1[(size_t)ks * P.cap + s] = B.xmass[(size_t)ks * B.n + i];
  atomicMax(maxpid, pid);
}

// readpartpositions.f90:115-148 -- warm start: the records of a dump (as partoutput writes them) -> particle SoA.
// One lane per record; arithmetic of the coordinate conversion in the host's real kind H.
template <typename R, typename H>
__global__ void __launch_bounds__(kBlock) k_readpart(Parts<R> P, const unsigned int *__restrict__ raw, long long n, int nspec,
                                                     H dx, H dy, H xlon0, H ylat0, double jul_header, double bdate, int mintime, int itrasplit0,
                                                     const int *__restrict__ nclass_in, int *__restrict__ status /* [0] error, [1] max npoint */) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int reclen = 8 + (10 + nspec) * (int)sizeof(H);
  const unsigned int *w = raw + (size_t)i * (size_t)(reclen / 4 + 2);
  auto get = [&](const unsigned int *q) -> H {
    if (sizeof(H) == 4) return (H)__uint_as_float(q[0]);
    return (H)__longlong_as_double((long long)(((unsigned long long)q[1] << 32) | (unsigned long long)q[0]));
  };
  const int hw = (int)sizeof(H) / 4;
  if ((int)w[0] != reclen || (int)w[reclen / 4 + 1] != reclen) { atomicExch(&status[0], 1); return; }
  const int npoint = (int)w[1];
  const H xlonin = get(w + 2), ylatin = get(w + 2 + hw), zin = get(w + 2 + 2 * hw);
  const int itramem_in = (int)w[2 + 3 * hw];
  if (xlonin == (H)-9999.9) { atomicExch(&status[0], 2); return; }     // a closing record inside the file: several dumps
  P.xt[i] = (double)((xlonin - xlon0) / dx);                              // :121-122
  P.yt[i] = (double)((ylatin - ylat0) / dy);
  P.zt[i] = (R)zin;
  P.npoint[i] = npoint;
  atomicMax(&status[1], npoint);                                          // numparticlecount, :123
  const double julpartin = jul_header + (double)itramem_in / 86400.;      // :140-147
  P.itramem[i] = (int)llround((julpartin - bdate) * (double)(H)86400.);
  P.idt[i] = mintime;
  P.itrasplit[i] = itrasplit0;                                            // :117
  P.itra1[i] = 0;
  P.nclass[i] = nclass_in ? nclass_in[i] : 1;
  P.up[i] = (R)0; P.vp[i] = (R)0; P.wp[i] = (R)0; P.us[i] = (R)0; P.vs[i] = (R)0; P.ws[i] = (R)0;
  P.cbt[i] = 1; P.pid[i] = (unsigned int)i;
  const unsigned int *x = w + 3 + 3 * hw + 7 * hw;
  for (int ks = 0; ks < nspec; ks++) P.xmass1[(size_t)ks * P.cap + i] = (R)get(x + ks * hw);
  if (P.xscav) for (int ks = 0; ks < nspec; ks++) P.xscav[(size_t)ks * P.cap + i] = (R)-1;   // as after a release; the dump does not carry it
}

// ---------------------------------------------------------------------------
// concoutput.f90:296-447 -- the sparse grid_conc writer.  For one (species, pointspec,
// age class): class mean of the sampling grid, the non-zero test, and the run-length compression (start index
// of every run of non-zero cells; values with a sign that flips from run to run) as two scans and two
// element-wise kernels.  Arithmetic in float, the reference's own kind for this routine, contraction off.
// ---------------------------------------------------------------------------
template <typename G>
__global__ void __launch_bounds__(kBlock) k_co_values(const G *__restrict__ grid, size_t class_stride, int nclass, long long n,
                                                      float *__restrict__ val, unsigned int *__restrict__ nz, unsigned int *__restrict__ rs) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto value = [&](long long j) -> float {          // mean_mod.f90:mean_sp, times nclassunc (concoutput.f90:323-325)
    float xl = 0.f;
    for (int l = 0; l < nclass; l++) xl = xl + (float)grid[(size_t)l * class_stride + j];
    return (xl / (float)nclass) * (float)nclass;
  };
  const float v = value(i);
  const bool on = v > 1.17549435e-38f;               // smallnum = tiny(0.0)
  const bool prev = i > 0 && value(i - 1) > 1.17549435e-38f;
  val[i] = v;
  nz[i] = on ? 1u : 0u;
  rs[i] = (on && !prev) ? 1u : 0u;                   // sp_zer: a run starts here
}

__global__ void __launch_bounds__(kBlock) k_co_write(const float *__restrict__ val, const unsigned int *__restrict__ nz, const unsigned int *__restrict__ rs,
                                                     const unsigned int *__restrict__ rpos /* exclusive scan of nz */,
                                                     const unsigned int *__restrict__ runid /* inclusive scan of rs */, long long n,
                                                     const float *__restrict__ scale /* volume (conc, pptv) or area, per cell */, int conc, float outnum, float tot_mu,
                                                     int idx0, int *__restrict__ wi, float *__restrict__ wr,
                                                     const float *__restrict__ dens = nullptr, float weightmolar = 1.f) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !nz[i]) return;
  const unsigned int run = runid[i];
  const float sp_fact = (run & 1u) ? 1.f : -1.f;     // sp_fact starts at -1 and flips at every run start
  if (rs[i]) wi[run - 1] = (int)i + idx0;
  float r;
  if (conc == 2) {                                     // mixing ratio, concoutput.f90:575-579
    r = sp_fact * 1.e12f * val[i] / scale[i] / outnum * 28.97f / weightmolar / dens[i];
  } else if (conc) {
    const float f3 = 1.e12f / scale[i] / outnum;     // factor3d, concoutput.f90:226 (ldirect = 1)
    r = sp_fact * val[i] * f3 / tot_mu;
  } else r = sp_fact * 1.e12f * val[i] / scale[i];
  wr[rpos[i]] = r;
}

// densityoutgrid, concoutput.f90:176-205: air density at the centre of every output cell from the met density of slot
// memind(2) (the r2 pack), nearest column, linear between the two z levels around the mid height of the output layer
template <typename R>
__global__ void __launch_bounds__(kBlock) k_co_density(View<R> V, int nxg, int nyg, int nzg, const float *__restrict__ outheight,
                                                       float dxout, float dyout, float outlon0, float outlat0, float dx, float dy, float xlon0, float ylat0,
                                                       float *__restrict__ dens) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)nxg * nyg * nzg) return;
  const int ix = (int)(i % nxg), jy = (int)((i / nxg) % nyg), kz = 1 + (int)(i / ((long long)nxg * nyg));
  const float halfheight = kz == 1 ? outheight[0] / 2.f : (outheight[kz - 1] + outheight[kz - 2]) / 2.f;
  int kzz;
  for (kzz = 2; kzz <= V.nz; kzz++)
    if ((float)V.height[kzz - 2] < halfheight && (float)V.height[kzz - 1] > halfheight) break;
  kzz = max(min(kzz, V.nz), 2);
  const float dz1 = halfheight - (float)V.height[kzz - 2], dz2 = (float)V.height[kzz - 1] - halfheight, dz = dz1 + dz2;
  float xl = outlon0 + (float)ix * dxout, yl = outlat0 + (float)jy * dyout;
  xl = (xl - xlon0) / dx;
  yl = (yl - ylat0) / dy;
  const int iix = max(min((int)lroundf(xl), V.nxmin1), 0), jjy = max(min((int)lroundf(yl), V.nymin1), 0);
  const size_t col = ((size_t)jjy * V.nx + iix) * V.nz;
  const float r1 = (float)V.r2[(col + (kzz - 1)) * 4 + V.m2 * 2], r0 = (float)V.r2[(col + (kzz - 2)) * 4 + V.m2 * 2];
  dens[i] = (r1 * dz1 + r0 * dz2) / dz;
}

// The wind pack blended in time for the step in flight: out0 at itime, out1 (may be NULL) at itime + lsynctime*ldirect.
// (y(memind(1))*dt2 + y(memind(2))*dt1)*dtt, interpol_wind.f90:189-191, per grid point instead of per particle.
template <typename R>
__global__ void __launch_bounds__(256) k_blend_w3(const R *__restrict__ w3, const R *__restrict__ r2, long long npoint, int m1, int m2, R dt1a, R dt2a, R dtta,
                                                  R dt1b, R dt2b, R dttb, R *__restrict__ out0, R *__restrict__ out1, R *__restrict__ rout0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npoint) return;
  const R *p = w3 + i * 6;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const R y1 = p[m1 * 3 + k], y2 = p[m2 * 3 + k];
    out0[i * 3 + k] = (y1 * dt2a + y2 * dt1a) * dtta;
    if (out1) out1[i * 3 + k] = (y1 * dt2b + y2 * dt1b) * dttb;
  }
  const R *q = r2 + i * 4;                     // (rho, drhodz) of the two slots, interpol_all.f90:189-198
#pragma unroll
  for (int k = 0; k < 2; k++) rout0[i * 2 + k] = (q[m1 * 2 + k] * dt2a + q[m2 * 2 + k] * dt1a) * dtta;
}

// After the stable sort of the slots by their 3-bit key: list length = number of keys <= 4 (PBL
// classes), particles due = number of keys <= 6.  One wave, two 64-ary searches (5 dependent
// loads each at 1e8 keys).
__device__ __forceinline__ long long count_le_sorted(const unsigned char *__restrict__ keys, long long lo, long long hi, unsigned char bound) {
  const int lane = threadIdx.x & 63;
  while (hi > lo) {   // keys[< lo] <= bound < keys[>= hi]
    const long long step = (hi - lo + 63) / 64;
    const long long pos = lo + lane * step;
    const bool le = pos < hi && keys[pos] <= bound;
    const int cnt = __popcll(__ballot(le));    // the keys are sorted: the lanes that see <= bound are the first cnt
    if (cnt == 0) { hi = lo; break; }
    const long long nlo = lo + (long long)(cnt - 1) * step + 1;
    const long long nhi = lo + (long long)cnt * step;
    lo = nlo;
    hi = nhi < hi ? nhi : hi;
  }
  return lo;
}
// live particles (itra1 /= -999999999) among the first n storage spaces: one atomic per block
__global__ void __launch_bounds__(256) k_count_live(const int *__restrict__ itra1, long long n, unsigned long long *__restrict__ out) {
  __shared__ unsigned int part[4];
  unsigned int c = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) c += itra1[i] != kDead;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(part[0] + part[1] + part[2] + part[3]));
}
// the latest time of birth (itramem, in the run's time direction) among the live particles of the first n storage spaces
__global__ void __launch_bounds__(256) k_birth_max(const int *__restrict__ itra1, const int *__restrict__ itramem, long long n, int ldirect, long long *__restrict__ out) {
  long long m = LLONG_MIN;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if (itra1[i] != kDead) m = max(m, (long long)ldirect * itramem[i]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m != LLONG_MIN) atomicMax(out, m);
}
// Counters of the step's work list (unsigned ints, zeroed at the start of every step):
//   [0]     length of the whole list (all PBL particles; k_pbl_finish)
//   [kCtrBase + 12 j + ...]: launch j of the Langevin kernel (ONE pointer gives a launch everything it needs):
//     + c,     c = 0..3: particles of stability class c + 1 in its list (they sit at the head of the class's segment)
//     + 4 + c: the chunk cursor of that class
//     + 8 + c: first entry of the class's segment in the list (the list is sorted by class; the same in every launch)
constexpr int kMaxSlices = 16, kCtrBase = 8, kCtrStride = 12, kCtrWords = kCtrBase + (kMaxSlices + 1) * kCtrStride;
__global__ void k_list_counts(const unsigned char *__restrict__ sorted_keys, long long n, unsigned int *__restrict__ ctr, Stats *st) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  long long upto[5];
  upto[0] = 0;
  for (int c = 1; c <= 4; c++) upto[c] = count_le_sorted(sorted_keys, upto[c - 1], n, (unsigned char)(8 * c - 1));
  const long long npbl = upto[4];
  const long long ndue = count_le_sorted(sorted_keys, npbl, n, kKeyDone);
  if (threadIdx.x == 0) {
    ctr[0] = (unsigned int)npbl;
    for (int c = 1; c <= 4; c++) {
      ctr[kCtrBase + c - 1] = (unsigned int)(upto[c] - upto[c - 1]);
      for (int j = 0; j <= kMaxSlices; j++) ctr[kCtrBase + kCtrStride * j + 8 + c - 1] = (unsigned int)upto[c - 1];
    }
    st->n_due += (unsigned long long)ndue;
  }
}

// The Langevin kernel: persistent waves, lane refill (see the header comment above).
// Only the pass loop lives here; set-up and completion run in k_prep / k_pbl_finish
// where all lanes are active.
// LEAN: no dry deposition, no settling (gases).  TSW / CBLF / RNGM: see pbl_pass.
//
// The kernel runs as a short sequence of launches per step (Engine::step).  A lane can SUSPEND its particle between two
// passes: the state goes into the particle's hand-over record (state PBL_CONTINUE) and the slot is appended to the next
// launch's list, class by class (one atomic per wave and event); the next launch continues it.  Three rules, all decided
// per wave:
//  * class purity: the list is sorted by stability class, every class has a chunk cursor of its own and a wave draws from
//    ONE class at a time.  When that class has no chunk left the wave runs on with what it holds, without refilling,
//    until fewer than drain_lanes lanes are busy, suspends those particles and moves to the next class with all lanes
//    free -- so the lanes of a wave always run ONE branch of hanna_short / cbl (before: for up to 300 passes after the
//    cursor had crossed a class boundary the waves ran both classes' code for the sake of a few long-lived particles:
//    12 % of the hanna_short and 7 % of the CBL sub-steps at 1.25e7 particles);
//  * drain: the same rule after the last class -- the wave hands its last particles on and ends; the next launch packs
//    the leftovers of all waves densely (before: the last tenth of a launch ran with waves of a lane or two);
//  * pass budget (cap_passes > 0, optional): at most cap_passes passes per particle and launch.
// The last launch of the sequence has no next list (drain_lanes = 0, cap_passes = 0): it runs everything to the end.
// A particle's random numbers are keyed on (particle number, step, draw index) and the draw index travels in the record:
// suspending does not change a bit of the result (tests/test_gpu_parity.py::test_time_slices_do_not_change_a_bit).
// SUSP = false: the instance of a step with ONE launch -- nothing can be suspended or resumed, and the refill, which runs with two
// lanes of the wave, carries none of that (the hand-over of suspended particles cost 2 % of the kernel's instructions at 1e8).
template <typename R, bool LEAN, int TSW, int CBLF, int RNGM, bool SUSP = false>
__global__ void __launch_bounds__(kBlock, sizeof(R) == 4 ? FPX_LOOP_WAVES_F32 : FPX_LOOP_WAVES) k_pbl_loop(View<R> V_arg, Parts<R> P_arg, PblRec<R> Q_arg, int itime, unsigned int step, Stats *st,
                                                     const unsigned int *__restrict__ pbl_list,
                                                     unsigned int *__restrict__ blk,
                                                     int cap_passes, int drain_lanes,
                                                     unsigned int *__restrict__ next_list) {
  // (the View alone: k_pbl_loop<double, true, 1, 1, 2> 157 -> 153 VGPRs, 49 -> 19 spilled SGPRs, 24 B -> no scratch; -0.5 to -1 % of the launch)
#if !defined(FPX_VIEW_BY_VALUE) && !defined(FPX_LOOP_ONLY_VIEW)
  struct KArgs { View<R> V; Parts<R> P; PblRec<R> Q; };
  const KArgs &ka = *(const KArgs *)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
  const View<R> &V = ka.V; const Parts<R> &P = ka.P; const PblRec<R> &Q = ka.Q;
  (void)V_arg; (void)P_arg; (void)Q_arg;
#else
  FPX_VIEW_FROM_KERNARG(V, V_arg);
  const Parts<R> &P = P_arg; const PblRec<R> &Q = Q_arg;
#endif
  // blk: this launch's block of the counters (see k_list_counts): per class c the first blk[c] entries of the class's segment
  // [blk[8 + c], ...) of pbl_list, the class's chunk cursor blk[4 + c]; the next launch's block follows
  const unsigned int *mine = blk;
  unsigned int *next_cnt = blk + kCtrStride;
  // (only the current class's length and segment start live in scalar registers, not all four: twelve registers through the
  // whole kernel were taken from the polynomial constants of the fine loop, which then came back through v_readlane in
  // every sub-step)
  // A short list (the later launches): only as many blocks as it fills take part, so that its waves are spread over the
  // CUs one block each instead of three to a SIMD on some and none on others (a pass of a wave alone on its SIMD takes a
  // third of the time).
  const unsigned int nlist_all = mine[0] + mine[1] + mine[2] + mine[3];
  if ((unsigned long long)blockIdx.x * kBlock >= (unsigned long long)nlist_all) return;
  const bool can_suspend = SUSP && (drain_lanes > 0 || cap_passes > 0);
  // dynamic LDS: [S_COUNT][kBlock] stash (per-lane pass-level state, see Stash) + the height column,
  // sized by the host (loop_smem_bytes) so that three blocks fit one CU for the usual nz
  extern __shared__ __align__(16) unsigned char fpx_loop_smem[];
  static_assert(kStashStride == kBlock, "stash layout is one column per thread of the block");
  R *stash_mem = reinterpret_cast<R *>(fpx_loop_smem);
  R *hgt = stash_mem + stash_slots<R>(LEAN, SUSP) * kStashStride;   // (the host sizes the block's LDS alike: loop_smem_bytes)
  for (int k = threadIdx.x; k < V.nz; k += blockDim.x) hgt[k] = V.height[k];
  // the block's copy of the lookup tables of the fp64 logarithm and exponential (m_log_abs, m_exp_tab): 768 B
  __shared__ double lds_tab[sizeof(R) == 8 ? kLdsTabDoubles : 1];
  if (sizeof(R) == 8)
    for (int k = threadIdx.x; k < kLdsTabDoubles; k += blockDim.x) lds_tab[k] = k < kLdsExpTabAt ? kLogTab[k >> 1][k & 1] : kExpTab[k - kLdsExpTabAt];
  __syncthreads();
  const Stash<R> S{(typename Stash<R>::lds_ptr)(stash_mem + threadIdx.x), (lds_tab_ptr)lds_tab};
  const int lane = threadIdx.x & 63;
  const TimeW<R> W = time_weights(V, itime);   // wave-uniform
  const R cap_r = cap_passes > 0 ? (R)cap_passes : (R)1e30;

  // The list is in slot order, i.e. sorted by grid cell after a locality sort.  A wave takes
  // CHUNKS of consecutive entries (one atomic per chunk) and refills its lanes from its own
  // chunk, so the particles a wave works on at any moment come from neighbouring cells
  // (same mixing height, same stability regime, shared cache lines).
  // 64 entries per claim: with larger chunks (nlist/(8*nwaves) = 2000 entries at 1e8 was tried first) the kernel
  // ended half a chunk's worth of work -- tens of milliseconds -- after the list ran out, most waves idle;
  // measured 429 -> 395 ms.  One atomic per 64 refills is still negligible.
  unsigned int cur = 0, end = 0;     // wave-uniform: the unread part of the wave's chunk
  unsigned int cbase = 0;            // wave-uniform: first entry of the chunk
  unsigned int ahead = 0;            // lane l: entry cbase + l of the list (the whole chunk, read with the claim)
  bool out_of_chunks = false;        // wave-uniform
  int wcls = 0;                      // wave-uniform: class (0..3) the wave draws its particles from
  unsigned int cls_out = 0u;         // wave-uniform: that class has no chunk left
  unsigned int cls_n = mine[0], cls_seg = mine[8];   // wave-uniform: length and first entry of that class's segment (read again when the wave moves on)
  // The wave-uniform values that are only read where a chunk is claimed -- once per 64 particles -- are kept in VECTOR
  // registers (the kernel has nine to spare; the empty asm makes them lane values for the compiler), not in scalar ones: the
  // scalar file is what the fine loop's polynomial constants live in, and every scalar held across the loop came back as a
  // v_readlane in each sub-step (+0.8 % VALU instructions at 1e8 with all of these in scalar registers).
#define FPX_IN_VGPR(x) asm volatile("" : "+v"(x))
  if (SUSP) { FPX_IN_VGPR(cbase); FPX_IN_VGPR(cls_out); FPX_IN_VGPR(cls_n); FPX_IN_VGPR(cls_seg); }
#ifdef FPX_LANE_STATS
  const unsigned long long t_s = wall_clock64();   // 100 MHz; timeline of the wave: start, list exhausted, end
  unsigned long long t_x = 0;
#endif

  bool have = false;
  unsigned int s = 0, pid = 0;
  double xt = 0, yt = 0;
  R zt = 0, wp = 0;
  int ldt = 0;
  short icbt = 1;
  LoopCtx<R> A;

  // the lanes of `who` (all of them hold a particle) hand their particles on: record written, slot appended to the next
  // launch's list in the segment of the wave's class
  auto write_record = [&](int state, int indz) {
    PblRecord<R> *rp = Q.rec + s;
    rp->v[0] = S.get(S_DX); rp->v[1] = S.get(S_DY); rp->v[2] = S.get(S_DAW); rp->v[3] = S.get(S_DCW);
    rp->v[4] = zt; rp->v[5] = S.get(S_UP); rp->v[6] = S.get(S_VP); rp->v[7] = wp;
    if (state == PBL_CONTINUE) {
      rp->v[8] = S.get(S_UST);      // hanna.f90:43 may have floored it; v[9..12] stay as k_prep wrote them
    } else {
      rp->v[8] = S.get(S_U); rp->v[9] = S.get(S_V); rp->v[10] = S.get(S_W);
    }
    rp->i[0] = A.nrand; rp->i[1] = ldt;
    rp->i[2] = pbl_pack(state, icbt, indz, A.ngrid, abs(A.itimec - itime));
    if (!LEAN && V.drydep) Q.tdep[s] = S.get(S_TDEP);
  };
  auto suspend = [&](bool mine_goes) {   // convergent: every lane of the wave calls it
    const unsigned long long sm = __ballot(mine_goes);
    if (sm == 0ull) return;
    if (mine_goes) { write_record(PBL_CONTINUE, 1); have = false; }
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(next_cnt + wcls, (unsigned int)__popcll(sm));
    base = __builtin_amdgcn_readfirstlane(base);
    const unsigned int seg = mine[8 + wcls];
    if (mine_goes) next_list[seg + base + (unsigned int)__popcll(sm & ((1ull << lane) - 1ull))] = s;
  };

  for (;;) {
    FPX_LANES(st, 10);
    unsigned long long need = __ballot(!have);
    if (need != 0ull && !out_of_chunks) {
      if (!SUSP) {
        // One launch, nothing to suspend: the class segments of the list are adjacent, so the wave walks the list as ONE
        // sequence with ONE cursor (chunks may straddle a class boundary, as in round 3)
        if (cur >= end) {
          unsigned int c = 0;
          if (lane == 0) c = atomicAdd(blk + 4, 1u);
          c = __builtin_amdgcn_readfirstlane(c);
          const unsigned long long c0 = (unsigned long long)c * 64u;
          if (c0 >= nlist_all) {
            out_of_chunks = true;
#ifdef FPX_LANE_STATS
            t_x = wall_clock64();
#endif
          } else {
            cur = (unsigned int)c0;
            cbase = cur;
            end = min(cur + 64u, nlist_all);
            ahead = pbl_list[min(cur + (unsigned int)lane, nlist_all - 1u)];
          }
        }
      } else
      while (cur >= end && !out_of_chunks) {   // wave-uniform: take the next chunk of the wave's class
        if (__builtin_amdgcn_readfirstlane(cls_out) == 0u) {
          const unsigned int nseg = __builtin_amdgcn_readfirstlane(cls_n), seg = __builtin_amdgcn_readfirstlane(cls_seg);
          const unsigned int kk = (nseg + 63u) >> 6;       // this class's length in chunks
          unsigned int c = 0;
          if (lane == 0) c = atomicAdd(blk + 4 + wcls, 1u);
          c = __builtin_amdgcn_readfirstlane(c);
          if (c < kk) {
            cur = seg + c * 64u;
            cbase = cur; FPX_IN_VGPR(cbase);
            end = min(cur + 64u, seg + nseg);
            ahead = pbl_list[min(cur + (unsigned int)lane, end - 1u)];
            break;
          }
          cls_out = 1u; FPX_IN_VGPR(cls_out);
        }
        // The wave's class has no chunk left.  A launch that can hand particles on keeps the wave on what it holds -- no
        // refill, so no second class in the wave -- until fewer than drain_lanes lanes are busy, then suspends those and
        // moves on to the next class with all lanes free; the last launch moves on at once (and mixes).
        if (can_suspend) {
          const int live = __popcll(__ballot(have));
          if (live > 0 && live >= drain_lanes) break;
          suspend(have);
          need = ~0ull;
        }
        wcls++;
        cls_out = 0u; FPX_IN_VGPR(cls_out);
        if (wcls < 4) { cls_n = mine[wcls]; cls_seg = mine[8 + wcls]; FPX_IN_VGPR(cls_n); FPX_IN_VGPR(cls_seg); }
        if (wcls == 4) {
          out_of_chunks = true;
#ifdef FPX_LANE_STATS
          t_x = wall_clock64();
#endif
        }
      }
      if (cur < end) {
        // A refill is ONE memory round trip: the list entry comes from the chunk read above (cross-lane), everything else
        // depends on the slot only and is requested together -- grid and mixing height travel in the record k_prep wrote.
        // (Before: list entry -> state -> mixing height -> six stash values one after the other = nine dependent round
        // trips, with two of the wave's lanes active, in 60 % of the passes.)
        const unsigned int rank = __popcll(need & ((1ull << lane) - 1ull));
        const unsigned int my = cur + rank;
        const unsigned int s_new = (unsigned int)__builtin_amdgcn_ds_bpermute((int)(min(my - cbase, 63u) << 2), (int)ahead);   // all lanes
        if (!have && my < end) {
          FPX_LANES(st, 8);
          s = s_new;
          const PblRecord<R> *rp = Q.rec + s;
          const double l_xt = P.xt[s], l_yt = P.yt[s];
          const unsigned int l_pid = P.pid[s];
          const int r_nrand = rp->i[0], r_ldt = SUSP ? rp->i[1] : 0, r_pk = rp->i[2];
          const R r_ust = rp->v[8], r_wst = rp->v[9], r_ol = rp->v[10], r_trans = rp->v[11], r_h = rp->v[12];
          // a fresh particle's state is in the particle arrays, a suspended one's in its record: both are requested
          // (one round trip either way; suspended particles are the few)
          R l_zt = P.zt[s], l_wp = P.wp[s], l_up = P.up[s], l_vp = P.vp[s];
          int l_idt = P.idt[s];
          short l_cbt = P.cbt[s];
          const R c_dx = SUSP ? rp->v[0] : (R)0, c_dy = SUSP ? rp->v[1] : (R)0, c_daw = SUSP ? rp->v[2] : (R)0, c_dcw = SUSP ? rp->v[3] : (R)0;
          const R c_zt = SUSP ? rp->v[4] : (R)0, c_up = SUSP ? rp->v[5] : (R)0, c_vp = SUSP ? rp->v[6] : (R)0, c_wp = SUSP ? rp->v[7] : (R)0;
          int l_npoint = 0;
          if (!LEAN && V.lsettling) l_npoint = P.npoint[s];
          R c_tdep = (R)0;
          if (SUSP && !LEAN && V.drydep) c_tdep = Q.tdep[s];
          __builtin_amdgcn_sched_barrier(0);   // every load above is issued before the first value is used
          const bool resumed = SUSP && pbl_state(r_pk) == PBL_CONTINUE;
          xt = l_xt; yt = l_yt; pid = l_pid;
          zt = resumed ? c_zt : l_zt; wp = resumed ? c_wp : l_wp;
          ldt = resumed ? r_ldt : l_idt;
          icbt = resumed ? (short)pbl_icbt(r_pk) : l_cbt;
          {
            R ddx, ddy;
            adv_begin_known(V, xt, yt, pbl_ngrid(r_pk), r_h, A, ddx, ddy);
            A.itimec = SUSP ? itime + pbl_elapsed(r_pk) * V.ldirect : itime;   // FRESH: elapsed = 0
            A.nrand = r_nrand;
            A.nsp = 0;
            if (!LEAN && V.lsettling) {
              const int nsp = settling_species(V, l_npoint);
              const SettleSpec<R> sp = settle_spec(V, nsp);
              S.put(S_SET_NUM, spec_row(V, nsp).density > (R)0 ? sp.num : (R)0);   // density(nsp) <= 0: no settling (advance.f90:525)
              S.put(S_SET_DQ6, sp.dq6); S.put(S_SET_V0, sp.vset);
              S.put(S_SETCELL, (R)settling_column(V, (R)xt, (R)yt));   // < nx*ny: exact in R (f32: grids up to 2^24 columns, checked at fpx_create)
              if (sizeof(R) == 4) S.put(S_RT_TAG, (R)0);
            }
            S.put(S_DDX, ddx); S.put(S_DDY, ddy);
          }
          S.put(S_DX, resumed ? c_dx : (R)0); S.put(S_DY, resumed ? c_dy : (R)0);
          S.put(S_DAW, resumed ? c_daw : (R)0); S.put(S_DCW, resumed ? c_dcw : (R)0);
          S.put(S_W, (R)0);
          S.put(S_UP, resumed ? c_up : l_up); S.put(S_VP, resumed ? c_vp : l_vp);
          S.put(S_UST, r_ust); S.put(S_WST, r_wst); S.put(S_OL, r_ol);
          S.put(S_TRANS, (r_wst * r_wst * r_wst) * r_trans);   // (wst**3)*transition, cbl.f90:103-104
          if (SUSP) S.put(S_NPASS, (R)0);
          if (!LEAN && V.drydep) S.put(S_TDEP, resumed ? c_tdep : (R)0);
          have = true;
        }
        cur = min(cur + (unsigned int)__popcll(need), end);
      }
    }
    if (!__any(have)) {
      if (out_of_chunks) break;   // no lane has work and the list is used up: the grid drains
      continue;                   // chunk ran dry mid-refill: take the next one
    }
    bool over_budget = false;
    if (have) {
      Rng<R, RNGM> G;
      make_rng(V, pid, step, G);
      int indz = 1;
      const int rc = pbl_pass<R, !LEAN, !LEAN, TSW, CBLF>(V, hgt, G, W, itime, xt, yt, zt, wp, ldt, icbt, A, S, indz, st);
      R npass = (R)0;
      if (SUSP) { npass = S.get(S_NPASS) + (R)1; S.put(S_NPASS, npass); }
      if (rc != PBL_CONTINUE) {
        FPX_LANES(st, 9);
        // the particle's state at the end of its last pass goes into its hand-over record: one contiguous line;
        // k_pbl_finish writes the particle arrays from it
        write_record(rc, indz);
        have = false;
      } else over_budget = SUSP && npass >= cap_r;
    }
    if (SUSP && cap_passes > 0) suspend(over_budget);
  }
#ifdef FPX_LANE_STATS
  if (lane == 0) {
    const unsigned long long t_e = wall_clock64();
    atomicAdd(&st->lanes[11][0], 1ull);
    atomicAdd(&st->lanes[11][1], t_x - t_s);            // sum over waves: start -> list exhausted
    atomicAdd(&st->lanes[12][0], t_e - t_s);            // sum over waves: start -> end
    atomicMax(&st->lanes[12][1], t_e - t_s);            // longest wave
    atomicMax(&st->lanes[13][0], ~(t_x - t_s));         // ~(first wave to see the list exhausted)
    atomicMax(&st->lanes[13][1], t_e - t_x);            // longest drain of one wave
  }
#endif
}

// diagnostics: the fp64 math helpers of fpx_device.hpp on plain arrays (fpx_math_probe)
__global__ void k_math_probe(int fn, const double *__restrict__ x, double *__restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = 0.0, c, ic2;
  switch (fn) {
    case 0: r = m_expp(v); break;
    case 1: r = m_logp(v); break;
    case 2: r = m_sqrtp(v); break;
    case 3: r = m_rcp(v); break;
    case 4: r = m_rsqrt(v); break;
    case 5: m_cuberoot_parts(v, c, ic2); r = c; break;
    case 7: r = m_erf_e(v, m_expp(-(v * v))); break;
    case 8: r = m_pow08(v); break;
    case 9: r = m_exp_tab(v, &kExpTab[0]); break;
    case 10: r = m_log_abs(v); break;
    case 11: r = m_rcbrt(v); break;
    default: m_cuberoot_parts(v, c, ic2); r = ic2; break;
  }
  y[i] = r;
}

// completion of ONE boundary-layer particle: label 700 if it left the PBL, sigmas for the mesoscale term, label 99 to the end
// of advance(), epilogue.  A function of its own for the same reason as prep_body: on a grid with polar caps the waves without
// a particle in a cap run the mother-grid instance (POLAR = false, CAPCHECK), the others the polar one.
template <typename R, bool DRYDEP, bool POLAR, bool NEST, bool CAPCHECK>
__device__ __forceinline__ void finish_body(const View<R> &V, const GridP<R> &Gp, Parts<R> &P, unsigned int s, const PblRecord<R> &rec, R tdep,
                                            int itime, unsigned int step, Stats *st, const R *hgt, DryDep<R> *dry) {
  constexpr bool MOTHER = !POLAR && !NEST;
  const int pk = rec.i[2];
  PState<R> ps;
  ps.xt = P.xt[s]; ps.yt = P.yt[s]; ps.zt = rec.v[4];
  ps.up = rec.v[5]; ps.vp = rec.v[6]; ps.wp = rec.v[7];
  ps.usigold = P.us[s]; ps.vsigold = P.vs[s]; ps.wsigold = P.ws[s];
  ps.ldt = rec.i[1]; ps.icbt = (short)pbl_icbt(pk);
  Rng<R> G;
  make_rng(V, P.pid[s], step, G);
  const TimeW<R> W = time_weights(V, itime);
  AdvCtx<R> A;
  {
    // same cell as at entry: the horizontal position does not change inside the loop
    AdvCtx<R> A0;
    adv_begin<R, MOTHER>(V, ps.xt, ps.yt, ps.zt, itime, 0, A0);
    A = A0;
  }
  if (V.lsettling) A.nsp = settling_species(V, P.npoint[s]);
  A.dxsave = rec.v[0]; A.dysave = rec.v[1]; A.dawsave = rec.v[2]; A.dcwsave = rec.v[3];
  A.u = rec.v[8]; A.v = rec.v[9]; A.w = rec.v[10];
  A.nrand = rec.i[0]; A.itimec = itime + pbl_elapsed(pk) * V.ldirect;
  const int rc = pbl_state(pk), indz = pbl_indz(pk);   // (every particle of the list is DONE or ESCAPED when the last time slice has run)
  R usig = (R)0, vsig = (R)0, wsig = (R)0;
  R prob[kMaxSpec];
  {
    Cell<R> C;
    cell_setup(C, A.ix, A.jy, A.ixp, A.jyp, A.xr, A.yr);
    if (rc == PBL_ESCAPED) {
      above_step(V, hgt, G, W, itime, ps.xt, ps.yt, ps.zt, ps.wp, ps.ldt, A, usig, vsig, wsig);
    } else {
      level_pair_sigma(V, fld_of(V, A.ngrid), C, W, indz, usig, vsig, wsig);   // advance.f90:604-606
    }
    // advance.f90:582-599 in closed form: prob(ks) = 1 - exp(-vdepo(ks) * T / (2*href)), T = the loop's sum of |dt| over the
    // passes that ended below 2*href (pbl_pass); the deposition velocity of the particle's cell, as every pass saw it
#pragma unroll
    for (int ks = 0; ks < kMaxSpec; ks++) {
      prob[ks] = (R)0;
      if (DRYDEP && ks < V.nspec && V.drydepspec[ks]) {
        const R vdepo = interp_vdep(V, fld_of(V, A.ngrid), C, W, ks);
        prob[ks] = (R)1 - m_expp(-vdepo * tdep / ((R)2. * (R)15.));   // href = 15, par_mod.f90:76
      }
    }
  }
  // what the epilogue reads of the particle travels with the last gather
  EpiPre<R> pre;
  int itramem = 0;
  unsigned int sl = s;
  auto late_epi = [&]() {
    asm volatile("" : "+v"(sl));
    pre.template load<DRYDEP>(V, Gp, P, sl);
    itramem = P.itramem[sl];
  };
  const int nstop = adv_finish<R, Rng<R>, POLAR, MOTHER, decltype(late_epi), CAPCHECK>(V, hgt, G, itime, ps, A, usig, vsig, wsig, late_epi);
  epilogue_store<R, DRYDEP>(V, Gp, P, s, itime, itramem, nstop, ps, prob, st, &pre, dry);
}

// completion of the PBL particles (finish_body).  One thread per list entry or slot.
template <typename R, bool DRYDEP, bool POLAR, bool NEST>
__global__ void __launch_bounds__(kBlock, FPX_FINISH_WAVES) k_pbl_finish(View<R> V_arg, GridP<R> Gp_arg, Parts<R> P_arg, PblRec<R> Q_arg, int itime, unsigned int step, Stats *st,
                                                       const unsigned char *__restrict__ pbl_key, long long numpart,
                                                       const unsigned int *__restrict__ pbl_list, const unsigned int *__restrict__ pbl_count) {
#ifndef FPX_VIEW_BY_VALUE
  struct KArgs { View<R> V; GridP<R> Gp; Parts<R> P; PblRec<R> Q; };
  const KArgs &ka = *(const KArgs *)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
  const View<R> &V = ka.V; const GridP<R> &Gp = ka.Gp; Parts<R> &P = const_cast<Parts<R> &>(ka.P); const PblRec<R> &Q = ka.Q;
  (void)V_arg; (void)Gp_arg; (void)P_arg; (void)Q_arg;
#else
  const View<R> &V = V_arg; const GridP<R> &Gp = Gp_arg; Parts<R> &P = P_arg; const PblRec<R> &Q = Q_arg;
#endif
  // Two orders.  pbl_list given: the work list (class by class, each class in slot = cell order): every lane busy.  With cost
  // buckets in the list (32 interleaved sub-sequences of the cell-sorted slots) following it costs this kernel half of its
  // time again in scattered record and state accesses (0.58 -> 0.87 ms at 1.25e7 particles); then pbl_list is null and the
  // kernel goes through the SLOTS, taking those whose key of this step says "boundary layer" (0.77 ms when every slot has a
  // lane and the others idle; with the per-wave queue below the waves are full).
  __shared__ R hgt[kMaxNz];
  for (int k = threadIdx.x; k < V.nz; k += blockDim.x) hgt[k] = V.height[k];
  __syncthreads();
  // Slot order: every wave walks tiles of 64 consecutive slots, collects the boundary-layer ones in a queue of its own in LDS
  // and works them off 64 at a time -- full waves in (nearly) slot order, whatever share of a tile is boundary layer.
  __shared__ unsigned int queue_mem[kBlock / 64][128];
  // (an LDS pointer by type: as a generic one it made the compiler emit an instruction its own verifier rejects in the polar instance)
  typedef volatile unsigned int __attribute__((address_space(3))) *lds_queue_ptr;
  const lds_queue_ptr queue = (lds_queue_ptr)&queue_mem[threadIdx.x >> 6][0];
  const int lane = threadIdx.x & 63;
  unsigned int qlen = 0;   // wave-uniform
  const long long nwork = pbl_list ? (long long)*pbl_count : (numpart + 63) >> 6;   // list entries | tiles of 64 slots
  const long long stride = pbl_list ? (long long)gridDim.x * blockDim.x : (long long)gridDim.x * (kBlock / 64);
  long long it = pbl_list ? (long long)blockIdx.x * blockDim.x : (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  for (;;) {   // ONE body for both orders (the body is large: a single call site keeps it inlined once)
    // one wave's worth of particles: slot s for the lanes with `mine` (every lane of the wave goes through the body: the
    // dry-deposition scatter at its end needs them all)
    unsigned int s = 0;
    bool mine = false;
    if (pbl_list) {
      if (it >= nwork) break;          // the same for every lane of the block
      const long long i = it + threadIdx.x;
      mine = i < nwork;
      if (mine) s = pbl_list[i];
      it += stride;
    } else {
      while (qlen < 64u && it < nwork) {
        const long long i = it * 64 + lane;
        const bool pbl = i < numpart && pbl_key[i] < kKeyDone;
        const unsigned long long m = __ballot(pbl);
        if (pbl) queue[qlen + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned int)i;
        qlen += (unsigned int)__popcll(m);
        it += stride;
      }
      if (qlen == 0u) break;           // the same for every lane of the wave
      const unsigned int take = min(qlen, 64u), rest = qlen - take;
      mine = (unsigned int)lane < take;
      if (mine) s = queue[lane];
      const unsigned int carry = (unsigned int)lane < rest ? queue[64 + lane] : 0u;
      if ((unsigned int)lane < rest) queue[lane] = carry;
      qlen = rest;
    }
    DryDep<R> dry;
#pragma unroll
    for (int ks = 0; ks < kMaxSpec; ks++) dry.dep[ks] = 0.f;
    dry.x = (R)0; dry.y = (R)0; dry.nage = 1; dry.kp = 1; dry.nclass = 1;
    if (mine) {
      const PblRecord<R> rec = Q.rec[s];
      const R tdep = DRYDEP ? Q.tdep[s] : (R)0;
      if (POLAR && !NEST && !__any(pbl_ngrid(rec.i[2]) < 0)) {   // wave-uniform: no particle of this wave sits in a polar cap
        finish_body<R, DRYDEP, false, false, true>(V, Gp, P, s, rec, tdep, itime, step, st, hgt, DRYDEP ? &dry : nullptr);
      } else {
        finish_body<R, DRYDEP, POLAR, NEST, false>(V, Gp, P, s, rec, tdep, itime, step, st, hgt, DRYDEP ? &dry : nullptr);
      }
    }
    if (DRYDEP && Gp.on && V.ldirect == 1) {
      // drydepokernel / drydepokernel_nest (timemanager.f90:690-696), with every lane of the wave: lanes of the same output
      // cell are summed into its neighbourhood before the atomics (wave_kernel_add); a lane without a deposit adds nothing
#pragma unroll
      for (int ks = 0; ks < kMaxSpec; ks++) {
        if (ks < V.nspec && V.drydepspec[ks]) {
          drydepo_particle<R, true>(V, Gp, dry.nclass, dry.dep[ks], ks, dry.x, dry.y, dry.nage, dry.kp);
          if (Gp.nested) drydepo_particle<R, true>(V, Gp, dry.nclass, dry.dep[ks], ks, dry.x, dry.y, dry.nage, dry.kp, true);
        }
      }
    }
  }
}

// conccalc.f90:50-295: every slot, whole waves stay convergent for the wave-level pre-reduction
template <typename R>
__global__ void __launch_bounds__(kBlock) k_conccalc(View<R> V_arg, GridP<R> Gp_arg, Parts<R> P_arg, long long numpart, int itime, R weight) {
#ifndef FPX_VIEW_BY_VALUE
  struct KArgs { View<R> V; GridP<R> Gp; Parts<R> P; };
  const KArgs &ka = *(const KArgs *)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
  const View<R> &V = ka.V; const GridP<R> &Gp = ka.Gp; const Parts<R> &P = ka.P;
  (void)V_arg; (void)Gp_arg; (void)P_arg;
#else
  const View<R> &V = V_arg; const GridP<R> &Gp = Gp_arg; const Parts<R> &P = P_arg;
#endif
  __shared__ R hgt[kMaxNz];
  __shared__ R outh[kMaxNz];
  for (int k = threadIdx.x; k < V.nz; k += blockDim.x) hgt[k] = V.height[k];
  for (int k = threadIdx.x; k < Gp.numzgrid; k += blockDim.x) outh[k] = Gp.outheight[k];
  __syncthreads();
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool active = s < numpart;
  double xt = 0, yt = 0;
  R zt = 0, xm[kMaxSpec];
  int itage = 0, npoint = 1, nclass = 1;
#pragma unroll
  for (int ks = 0; ks < kMaxSpec; ks++) xm[ks] = (R)0;
  if (active) {
    // the particle's state in ONE memory round trip (nearly every particle is due; the asm keeps the compiler from sinking
    // each load behind the branch that first needs it)
    int itra1 = P.itra1[s], itramem = P.itramem[s];
    xt = P.xt[s]; yt = P.yt[s]; zt = P.zt[s];
    npoint = P.npoint[s]; nclass = P.nclass[s];
#pragma unroll
    for (int ks = 0; ks < kMaxSpec; ks++)
      if (ks < V.nspec) xm[ks] = P.xmass1[(size_t)ks * P.cap + s];
    static_assert(kMaxSpec == 5, "the tie below names five species");
    asm volatile("" : "+v"(itra1), "+v"(itramem), "+v"(xt), "+v"(yt), "+v"(zt), "+v"(npoint), "+v"(nclass),
                      "+v"(xm[0]), "+v"(xm[1]), "+v"(xm[2]), "+v"(xm[3]), "+v"(xm[4]));
    itage = abs(itime - itramem);
    active = itra1 == itime;
    if (!(xt >= 0. && xt <= (double)V.nxmin1 && yt >= 0. && yt <= (double)V.nymin1) || !(zt == zt)) active = false;
  }
  if (P.xscav) {   // wave-uniform: DRYBKDEP / WETBKDEP
    R sc[kMaxSpec];
#pragma unroll
    for (int ks = 0; ks < kMaxSpec; ks++) sc[ks] = (active && ks < V.nspec) ? m_max(P.xscav[(size_t)ks * P.cap + s], (R)0) : (R)0;
    conccalc_particle(V, Gp, hgt, active, xt, yt, zt, itage, npoint, nclass, xm, weight, sc, outh);
  } else {
    conccalc_particle(V, Gp, hgt, active, xt, yt, zt, itage, npoint, nclass, xm, weight, (const R *)nullptr, outh);
  }
}

// The receptor block of timemanager.f90:564-598 (backward runs with DRYBKDEP / WETBKDEP): once per particle, before it is
// moved for the first time -- xscav_frac1 < 0 marks a particle that has not been through it (releaseparticles.f90:167-171).
// get_vdep_prob.f90 sets the cell (the nest's inside a nest) but not the weights p1..p4, dt1, dt2, dtt that
// interpol_vdep[_nests] reads: those are the ones initialize() of the same particle left in interpol_mod, i.e. the
// mother grid's weights at the particle's position and the step's time weights -- computed here from the position.
template <typename R>
__global__ void __launch_bounds__(kBlock) k_bkdep(View<R> V_arg, WetP<R> Wp, Parts<R> P, long long numpart, int itime, int drybkdep, int wetbkdep,
                                                  const R *__restrict__ zspan /* [numpoint] zpoint2 - zpoint1 */) {
  FPX_VIEW_FROM_KERNARG(V, V_arg);
  __shared__ R hgt[kMaxNz];
  for (int k = threadIdx.x; k < V.nz; k += blockDim.x) hgt[k] = V.height[k];
  __syncthreads();
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= numpart) return;
  if (P.itra1[s] != itime) return;
  bool todo = false;
  for (int ks = 0; ks < V.nspec; ks++) todo = todo || P.xscav[(size_t)ks * P.cap + s] < (R)0;
  if (!todo) return;
  const double xt = P.xt[s], yt = P.yt[s];
  const R zt = P.zt[s];
  if (!(xt >= 0. && xt <= (double)V.nxmin1 && yt >= 0. && yt <= (double)V.nymin1) || !(zt == zt)) return;   // k_prep terminates it
  if (drybkdep) {
    const R href = (R)15.;
    // get_vdep_prob.f90:52-99
    const int ngrid = pick_grid<R, false>(V, xt, yt);
    int ix, jy;
    if (ngrid > 0) {
      const NestDesc<R> &N = V.nest[ngrid - 1];
      ix = (int)(R)((xt - (double)N.xl) * (double)N.xres);
      jy = (int)(R)((yt - (double)N.yl) * (double)N.yres);
    } else {
      ix = (int)xt; jy = (int)yt;
    }
    Cell<R> C;
    cell_setup(C, ix, jy, ix + 1, jy + 1, (R)xt, (R)yt);
    {   // the weights of initialize(): ddx = real(xt) - real(int(xt)) on the mother grid
      const int ixm = (int)xt, jym = (int)yt;
      const R ddx = (R)xt - (R)ixm, ddy = (R)yt - (R)jym, rddx = (R)1 - ddx, rddy = (R)1 - ddy;
      C.p1 = rddx * rddy; C.p2 = ddx * rddy; C.p3 = rddx * ddy; C.p4 = ddx * ddy;
    }
    const Fld<R> F = fld_of(V, ngrid);
    const TimeW<R> W = time_weights(V, itime);
    for (int ks = 0; ks < V.nspec; ks++) {
      if (!(P.xscav[(size_t)ks * P.cap + s] < (R)0)) continue;
      if (V.drydepspec[ks]) {
        R prob = (R)0;
        if (V.drydep && zt < (R)2. * href) prob = interp_vdep(V, F, C, W, ks);   // :105-127: the deposition velocity itself
        P.xscav[(size_t)ks * P.cap + s] = prob;
      } else {
        P.xmass1[(size_t)ks * P.cap + s] = (R)0;
        P.xscav[(size_t)ks * P.cap + s] = (R)0;
      }
    }
  }
  if (wetbkdep) {
    const int np = release_index(V, P.npoint[s]);
    for (int ks = 0; ks < V.nspec; ks++) {
      if (!(P.xscav[(size_t)ks * P.cap + s] < (R)0)) continue;
      R grfr = (R)0;
      const R wetscav = get_wetscav(V, Wp, hgt, itime, V.lsynctime, xt, yt, zt, ks, grfr);
      if (wetscav > (R)0) {
        P.xscav[(size_t)ks * P.cap + s] = wetscav * zspan[np] * grfr;
      } else {
        P.xmass1[(size_t)ks * P.cap + s] = (R)0;
        P.xscav[(size_t)ks * P.cap + s] = (R)0;
      }
    }
  }
}

// wetdepo.f90:58-151: every live particle that is due or overdue
template <typename R>
__global__ void __launch_bounds__(kBlock) k_wetdepo(View<R> V_arg, GridP<R> Gp_arg, WetP<R> Wp_arg, Parts<R> P_arg, long long numpart, int itime,
                                                    int ltsample, int loutnext) {
#ifndef FPX_VIEW_BY_VALUE
  struct KArgs { View<R> V; GridP<R> Gp; WetP<R> Wp; Parts<R> P; };
  const KArgs &ka = *(const KArgs *)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
  const View<R> &V = ka.V; const GridP<R> &Gp = ka.Gp; const WetP<R> &Wp = ka.Wp; const Parts<R> &P = ka.P;
  (void)V_arg; (void)Gp_arg; (void)Wp_arg; (void)P_arg;
#else
  const View<R> &V = V_arg; const GridP<R> &Gp = Gp_arg; const WetP<R> &Wp = Wp_arg; const Parts<R> &P = P_arg;
#endif
  __shared__ R hgt[kMaxNz];
  for (int k = threadIdx.x; k < V.nz; k += blockDim.x) hgt[k] = V.height[k];
  __syncthreads();
  // every lane stays to the end (the wave-level sums of wetdepo_scatter need convergent waves): `live` says whether the lane
  // holds a particle that is scavenged at all
  const long long s0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool live = s0 < numpart;
  const long long s = live ? s0 : 0;
  // the particle's state in ONE memory round trip (as in k_conccalc)
  int itra1 = P.itra1[s], itramem = P.itramem[s], nunc = P.nclass[s];
  int kp = Gp.ioutputforeachrelease == 1 ? P.npoint[s] : 1;
  double xt = P.xt[s], yt = P.yt[s];
  R zt = P.zt[s];
  R xm_all[kMaxSpec];
#pragma unroll
  for (int ks = 0; ks < kMaxSpec; ks++) xm_all[ks] = (ks < V.nspec && Wp.wetdepspec[ks]) ? P.xmass1[(size_t)ks * P.cap + s] : (R)0;
  static_assert(kMaxSpec == 5, "the tie below names five species");
  asm volatile("" : "+v"(itra1), "+v"(itramem), "+v"(nunc), "+v"(kp), "+v"(xt), "+v"(yt), "+v"(zt),
                    "+v"(xm_all[0]), "+v"(xm_all[1]), "+v"(xm_all[2]), "+v"(xm_all[3]), "+v"(xm_all[4]));
  if (itra1 == kDead) live = false;
  if (V.ldirect == 1) { if (itra1 > itime) live = false; } else { if (itra1 < itime) live = false; }
  if (!(xt >= 0. && xt <= (double)V.nxmin1 && yt >= 0. && yt <= (double)V.nymin1) || !(zt == zt)) live = false;
  if (!live) { xt = 0.; yt = 0.; zt = (R)0; }      // a harmless position for the lanes that only take part in the sums
  const int ldeltat = itime <= loutnext ? itime - (loutnext - Gp.loutstep) : itime - loutnext;   // wetdepo.f90:58-62
  const int nage = ageclass(Gp, abs(itra1 - itramem));
  const R smallnum = sizeof(R) == 4 ? (R)1.17549435e-38f : (R)2.2250738585072014e-308;
  for (int ks = 0; ks < V.nspec; ks++) {
    if (!Wp.wetdepspec[ks]) continue;
    R grfr = (R)0;
    R wetdeposit = (R)0;
    if (live) {
      const R wetscav = get_wetscav(V, Wp, hgt, itime, ltsample, xt, yt, zt, ks, grfr);
      const R xm = pick(xm_all, ks);
      if (wetscav > (R)0) wetdeposit = xm * ((R)1 - m_exp(-wetscav * (R)abs(ltsample))) * grfr;
      const R restmass = xm - wetdeposit;
      P.xmass1[(size_t)ks * P.cap + s] = restmass > smallnum ? restmass : (R)0;
      if (V.decay[ks] > (R)0) wetdeposit = wetdeposit * m_exp((R)abs(ldeltat) * V.decay[ks]);
    }
    if (V.ldirect == 1 && Gp.on) wetdepo_scatter(V, Gp, nunc, wetdeposit, ks, (R)xt, (R)yt, nage, kp);
    if (V.ldirect == 1 && Gp.on && Gp.nested) wetdepo_scatter(V, Gp, nunc, wetdeposit, ks, (R)xt, (R)yt, nage, kp, true);   // wetdepo.f90:142
  }
}

template <typename T>
__global__ void k_convert(const T *__restrict__ src, double *__restrict__ dst64, float *__restrict__ dst32, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dst64) dst64[i] = (double)src[i];
  if (dst32) dst32[i] = (float)src[i];
}

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
FPX_TU_CLOSE   // one type for all translation units
struct EngineBase {
  virtual ~EngineBase() {}
  virtual int set_height(const void *h, int n) = 0;
  virtual int upload_fields(int slot, const fpx_fields *f) = 0;
  virtual int verttransform(int slot, const fpx_model_levels *m, const fpx_fields *sfc, const fpx_fields_out *out) = 0;
  virtual int verttransform_nest(int nest, int slot, const fpx_model_levels *m, const fpx_fields *sfc, const fpx_fields_out *out) = 0;
  virtual int calcpar(int slot, const fpx_calcpar_in *c, const fpx_calcpar_out *out) = 0;
  virtual double cp_ms() = 0;
  virtual int set_windtime(const int32_t mt[2], const int32_t mi[2]) = 0;
  virtual int rng_fill_table() = 0;
  virtual int rng_set_table(const void *t, int n) = 0;
  virtual int rng_get_table(void *t, int n) = 0;
  virtual int upload_particles(long long first, long long count, const fpx_particles *p) = 0;
  virtual int download_particles(long long first, long long count, const fpx_particles *p) = 0;
  virtual int set_numpart(long long n) = 0;
  virtual int set_release_points(int numpoint, const void *xmass, const int32_t *npart) = 0;
  virtual int set_release_heights(int numpoint, const void *zpoint1, const void *zpoint2) = 0;
  virtual int release_init(const fpx_release *r) = 0;
  virtual int releaseparticles(int itime, int64_t *numpart, int32_t *numparticlecount, void *xmasssave, void *rho_rel, int64_t *nreleased) = 0;
  virtual int split_particles(int itime, int64_t *numpart) = 0;
  virtual size_t redist_bytes(int64_t num_trans) = 0;
  virtual int redist_pack(int itime, int64_t num_trans, void *buf, size_t buf_bytes, int64_t *numpart) = 0;
  virtual int redist_unpack(int itime, int64_t num_trans, const void *buf, size_t buf_bytes, int64_t *numpart) = 0;
  virtual int step(int itime, fpx_step_stats *st, bool async) = 0;
  virtual int sync() = 0;
  virtual int counters(fpx_step_stats *out, int reset) = 0;
  virtual int kernel_time(double *ms, long long *launches, int reset, double *parts) = 0;
  virtual int sort_particles() = 0;
  virtual int seed_particles(long long n, unsigned long long seed, double frac_pbl, double zmax, double lat_margin,
                             int itime0) = 0;
  virtual int outgrid_init(const fpx_outgrid *g, const void *outheight) = 0;
  virtual int set_output_times(int loutnext, int loutstep) = 0;
  virtual int conccalc(int itime, double weight) = 0;
  virtual int get_grids(void *gridunc, void *drygridunc, int allreduce, int clear) = 0;
  virtual int count_particles(int64_t *local, int64_t *total, int allreduce) = 0;
  virtual int lane_stats(uint64_t *out, int n, int reset) = 0;
  virtual int set_option(const char *name, const char *value) = 0;
  virtual int get_info(const char *name, int64_t *value) = 0;
  virtual int comm_init(const void *id, int nbytes, int nranks, int rank) = 0;
  virtual int comm_init_host(int nranks, int rank, fpx_allreduce_fn fn, void *user) = 0;
  virtual int nests_init(const fpx_nests *n) = 0;
  virtual int upload_nest_fields(int nest, int slot, const fpx_fields *f) = 0;
  virtual int wet_init(const fpx_wet_config *w) = 0;
  virtual int upload_wet_fields(int slot, const fpx_wet_fields *f) = 0;
  virtual int upload_wet_nest_fields(int nest, int slot, const fpx_wet_fields *f, int readclouds_nest) = 0;
  virtual int wetdepo(int itime, int ltsample, int loutnext) = 0;
  virtual int get_wetgrid(void *wetgridunc, int allreduce) = 0;
  virtual int outgrid_nest_init(const fpx_outgrid_nest *g) = 0;
  virtual int get_grids_nest(void *griduncn, void *drygriduncn, void *wetgriduncn, int allreduce, int clear) = 0;
  virtual int receptors_init(int n, const void *x, const void *y, const void *area) = 0;
  virtual int get_receptors(void *creceptor, int ld, int allreduce, int clear) = 0;
  virtual void *stream_ptr() = 0;
  virtual double vt_ms() = 0;
  virtual double po_ms() = 0;
  virtual int upload_diag_fields(int slot, const fpx_diag_fields *f) = 0;
  virtual int upload_diag_nest_fields(int nest, int slot, const fpx_diag_fields *f) = 0;
  virtual int partoutput(int itime, const char *path, int64_t *nrec) = 0;
  virtual int concoutput(int itime, const fpx_concout *c, const char *prefix, int clear) = 0;
  virtual int readpartpositions(const char *path, const fpx_restart *r, int64_t *numpart_out, int32_t *numparticlecount, int32_t *itimein) = 0;
  virtual int conv_init(const fpx_conv_config *c) = 0;
  virtual int upload_conv_fields(int slot, const fpx_conv_fields *f) = 0;
  virtual int convmix(int itime, int64_t *nmoved) = 0;
  virtual int cbaseflux_io(void *host, bool set) = 0;
  virtual int upload_conv_nest_fields(int nest, int slot, const fpx_conv_fields *f) = 0;
  virtual int cbaseflux_nest_io(int nest, void *host, bool set) = 0;
  virtual double conv_ms() = 0;
  virtual int checkpoint_write(const char *path, int itime, int numparticlecount) = 0;
  virtual int checkpoint_read(const char *path, int32_t *itime, int64_t *numpart_out, int32_t *numparticlecount) = 0;
};
EngineBase *make_engine_f32(const fpx_config *cfg, int *rc);   // defined by the unit that holds Engine<float>
// The three kernels of the step are compiled in units of their own (fpx_tu.hpp: FPX_TU_PART 1 and 2) and reach the
// engine as untyped host-stub addresses; the engine casts them to the kernel's signature (same headers, same layout).
// real_bytes 8|4 picks the arithmetic type.
const void *step_kernel_prep(int real_bytes, bool drydep, bool init, bool polar, bool nest);
const void *step_kernel_loop(int real_bytes, bool lean, int turbswitch, int cblflag, int rng_mode, bool susp, bool *is_lean);
const void *step_kernel_finish(int real_bytes, bool drydep, bool polar, bool nest);
FPX_TU_OPEN

template <typename R>
struct Engine : EngineBase {
  fpx_config cfg;
  hipStream_t stream = nullptr;
  View<R> V;
  Parts<R> P;
  bool maybe_new = true;   // particles may have been released since the last step
  // A particle is initialised in the step whose time equals its itramem (timemanager.f90:553) -- normally the step after it was
  // released or uploaded (maybe_new).  A host may also hand over particles that are born LATER (itra1 = itramem in the future):
  // the instance of k_prep that can initialize() runs at every step up to the latest birth among the particles the host placed
  // (found by one reduction over the particle arrays at the first step after an upload, a warm start, a restore or a receive).
  bool births_unknown = true;
  long long birth_horizon = LLONG_MIN;   // ldirect * itramem, the latest among the live particles at the last such event
  int find_birth_horizon() {
    births_unknown = false;
    birth_horizon = LLONG_MIN;
    if (numpart == 0) return 0;
    int rc;
    if (!d_count && (rc = dalloc(&d_count, 4))) return rc;
    const long long init = LLONG_MIN;
    HIPCHK(hipMemcpyAsync(d_count, &init, sizeof(long long), hipMemcpyHostToDevice, stream));
    const int nb = (int)std::min<long long>((numpart + 255) / 256, 256 * 16);
    k_birth_max<<<nb, 256, 0, stream>>>(P.itra1, P.itramem, numpart, cfg.ldirect, d_count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&birth_horizon, d_count, sizeof(long long), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  bool height_set = false, window_set = false, table_set = false, slot_loaded[2] = {false, false};
  long long numpart = 0;
  // owned device memory
  std::vector<void *> owned;
  void *staging = nullptr;
  size_t staging_bytes = 0;
  unsigned int *slot_of_pid = nullptr;   // only after a locality sort; read it through slot_map()
  bool slot_map_dirty = false;
  // Time-blended wind packs of the step in flight (View::w3t0 / w3t1).  Worth their 0.65 GB of extra traffic per step only
  // for a large cloud.  Blended and unblended gathers round differently, so the switch is a function of the configuration
  // alone -- fpx_config.blend_mode, or in its automatic mode the run's particle count over ALL ranks (global_particles) --
  // never of what this rank or this step happens to hold: runs with different rank counts, restarted runs and runs under
  // fpx_step_async stay bitwise equal (README_PARALLEL.md:189-192).
  R *d_w3t[2] = {nullptr, nullptr}, *d_r2t = nullptr;
  static constexpr long long kBlendMinGlobal = 30000000ll;
  bool blend_on() const {
    if (cfg.blend_mode == 1) return true;
    if (cfg.blend_mode == 2) return false;
    return (cfg.global_particles > 0 ? cfg.global_particles : cfg.max_particles) >= kBlendMinGlobal;
  }
  unsigned long long blended_steps = 0;
  int blend_winds(int itime) {
    V.w3t0 = nullptr; V.w3t1 = nullptr; V.r2t0 = nullptr;
    if (!blend_on()) return 0;
    blended_steps++;
    const long long npoint = (long long)cfg.nx * cfg.ny * cfg.nz;
    int rc;
    for (int k = 0; k < 2; k++)
      if (!d_w3t[k]) { if ((rc = dalloc(&d_w3t[k], (size_t)npoint * 3))) return rc; }
    if (!d_r2t) { if ((rc = dalloc(&d_r2t, (size_t)npoint * 2))) return rc; }
    const int t1 = itime + std::abs(cfg.lsynctime) * cfg.ldirect;   // the time of advance.f90:836-841 (ldt = |lsynctime| there)
    const bool have1 = std::abs((long long)t1) <= std::abs((long long)V.memtime1);      // advance.f90:836: no Petterssen step beyond the window
    const R dt1a = (R)(itime - V.memtime0), dt2a = (R)(V.memtime1 - itime), dtta = (R)1 / (dt1a + dt2a);
    const R dt1b = (R)(t1 - V.memtime0), dt2b = (R)(V.memtime1 - t1), dttb = have1 ? (R)1 / (dt1b + dt2b) : (R)0;
    k_blend_w3<R><<<(int)((npoint + 255) / 256), 256, 0, stream>>>(V.w3, V.r2, npoint, V.m1, V.m2, dt1a, dt2a, dtta, dt1b, dt2b, dttb,
                                                                     d_w3t[0], have1 ? d_w3t[1] : (R *)nullptr, d_r2t);
    HIPCHK(hipGetLastError());
    V.w3t0 = d_w3t[0];
    V.r2t0 = d_r2t;
    V.w3t1 = have1 ? d_w3t[1] : nullptr;
    return 0;
  }
  const unsigned int *slot_map() {
    if (slot_of_pid && slot_map_dirty) {
      k_slot_map<<<(int)((P.cap + kBlock - 1) / kBlock), kBlock, 0, stream>>>(P.pid, P.cap, d_slot_of_pid);
      slot_map_dirty = false;
    }
    return slot_of_pid;
  }
  unsigned int *d_slot_of_pid = nullptr;
  Parts<R> P2;                           // second particle buffer set (sort ping-pong), allocated lazily
  bool have_p2 = false;
  unsigned int *d_keys = nullptr, *d_keys2 = nullptr, *d_vals = nullptr, *d_vals2 = nullptr;
  void *d_sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  Stats *d_stats = nullptr;
  unsigned int *d_pbl_list = nullptr, *d_pbl_ctr = nullptr;   // the counters of the work list and of the launches that go through it (layout: k_list_counts)
  unsigned int *d_surv[2] = {nullptr, nullptr};               // lists of the suspended particles (ping-pong between launches), allocated with the first sliced step
  std::vector<int> slice_caps;                                // pass budget of each launch of the Langevin kernel, the last one 0 (none)
  struct Options {                                            // fpx_set_option
    int verbose = 0, pbl_blocks_per_cu = 0, prep_lds_pad = 0, permute = 0 /* 0 auto, 1 direct, 2 staged */, vt_unfused = 0;
    long conv_scratch_mb = 0;
    int conv_one_lane = 0, conv_no_walk = 0, conv_rows_plain = 0;
    int pbl_drain_lanes = -1;                     // -1: the engine's default (FPX_DRAIN_LANES)
    int prep_init_always = 0;                     // measurements: every step runs the instance of k_prep that can initialize() new particles
    int pbl_cost_buckets = FPX_COST_BUCKETS;      // -1: by the size of this rank's cloud
    std::vector<int> pbl_slices;
  } opt;
  int drain_lanes() const { return opt.pbl_drain_lanes >= 0 ? opt.pbl_drain_lanes : FPX_DRAIN_LANES; }
  unsigned char *d_pbl_flag = nullptr, *d_pbl_flag2 = nullptr;
  unsigned int *d_iota = nullptr;
  PblRec<R> Q;
  GridP<R> Gp;
  WetP<R> Wp;
  bool wet_on = false, wet_slot[2] = {false, false};
  WetNest<R> h_wnest[kMaxNests] = {};     // host copy of the device table Wp.nest
  WetNest<R> *d_wnest = nullptr;
  bool wet_nest_slot[kMaxNests][2] = {};
  size_t n_grid3 = 0, n_grid2 = 0, n_grid3n = 0, n_grid2n = 0;
  ncclComm_t comm = nullptr;
  int comm_ranks = 1, comm_rank = 0;
  long long rel_global_count = 0;        // particles released so far by all ranks (key of the release's counter RNG)
  // host-supplied all-reduce (fpx_comm_init_host): the transport of an MPI host, or gloo in the tests
  fpx_allreduce_fn host_allreduce = nullptr;
  void *host_allreduce_user = nullptr;
  void *red_pin = nullptr;               // pinned bounce buffer of the host transport: [send | recv]
  size_t red_pin_bytes = 0;
  // Receive buffers of the grid reduction (the reference's gridunc0, drygridunc0, wetgridunc0, griduncn0, ... and
  // creceptor0: mpi_mod.f90:2451-2492,2543-2569; allocated in outgrid_init.f90:220-225).  The per-rank partial sums
  // stay untouched in Gp.*: drygridunc / wetgridunc accumulate over the whole run and are reduced again at every
  // output time.  Allocated on the first reduction.
  R *gridunc0 = nullptr, *griduncn0 = nullptr, *creceptor0 = nullptr;
  float *drygridunc0 = nullptr, *wetgridunc0 = nullptr, *drygriduncn0 = nullptr, *wetgriduncn0 = nullptr;
  bool red_valid[7] = {};                // which receive buffers hold the sums of the last reduction
  enum { RG_GRID = 0, RG_DRY, RG_WET, RG_GRIDN, RG_DRYN, RG_WETN, RG_REC };
  void *d_sel_tmp = nullptr;
  size_t sel_tmp_bytes = 0;
  int pbl_grid = 0, pbl_per_cu = 0;
  // TABLE_SEQ state
  HostRng<float> rng4;
  HostRng<double> rng8;
  std::vector<float> tab4;
  std::vector<double> tab8;
  int *d_nrand_adv = nullptr, *d_nrand_init = nullptr;
  float *d_dcas4 = nullptr, *d_dcas14 = nullptr;
  double *d_dcas8 = nullptr, *d_dcas18 = nullptr;
  unsigned char *d_flags = nullptr;
  std::vector<unsigned char> h_flags;
  std::vector<int> h_nrand_adv, h_nrand_init;
  std::vector<float> h_dcas4, h_dcas14;
  std::vector<double> h_dcas8, h_dcas18;
  // timing
  struct StepEvents { hipEvent_t e[5]; };   // before k_prep | before k_pbl_loop | after it | after k_pbl_finish | after k_prep (before the work-list sort)
  std::vector<StepEvents> ev_pool;
  size_t ev_used = 0;
  double acc_ms = 0, acc_part_ms[4] = {0, 0, 0, 0};   // k_prep + work list | k_pbl_loop | k_pbl_finish | k_prep alone
  long long acc_launches = 0;
  unsigned int step_counter = 0;

  template <typename T>
  int dalloc(T **p, size_t n) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return fail(FPX_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    owned.push_back(q);
    *p = (T *)q;
    return 0;
  }
  int ensure_staging(size_t bytes) {
    if (bytes <= staging_bytes) return 0;
    if (staging) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(staging)); staging = nullptr; staging_bytes = 0; }
    HIPCHK(hipMalloc(&staging, bytes));
    staging_bytes = bytes;
    return 0;
  }

  int init(const fpx_config *c) {
    cfg = *c;
    if (cfg.nz > kMaxNz) return fail(FPX_ERR_ARG, "nz exceeds the engine's level limit (512)");
    if (cfg.nx < 2 || cfg.ny < 2 || cfg.nz < 2) return fail(FPX_ERR_ARG, "grid too small");
    if ((long long)cfg.nx * cfg.ny * cfg.nz > 0xFFFFFFF0ll) return fail(FPX_ERR_ARG, "grid too large: nx*ny*nz must fit 32 bits (the kernels index grid cells with 32-bit integers)");
    if (cfg.nxmax < cfg.nx || cfg.nymax < cfg.ny || cfg.nzmax < cfg.nz) return fail(FPX_ERR_ARG, "allocated extents smaller than used extents");
    if (cfg.nspec < 1 || cfg.nspec > FPX_MAXSPEC || cfg.maxspec < cfg.nspec) return fail(FPX_ERR_ARG, "bad nspec/maxspec");
    if (cfg.max_particles < 1 || cfg.max_particles > 0xFFFFFFF0ll) return fail(FPX_ERR_ARG, "bad max_particles");
    if (cfg.ifine < 1) return fail(FPX_ERR_ARG, "ifine must be >= 1");
    if (cfg.lsynctime == 0 || std::abs((long long)cfg.lsynctime) > 65535) return fail(FPX_ERR_ARG, "lsynctime must be non-zero and at most 65535 s in magnitude (the hand-over record of the Langevin kernel keeps |itimec - itime| in 16 bits)");
    if (cfg.ipout == 3) return fail(FPX_ERR_UNSUPPORTED, "ipout = 3: the particle loop's partpos_average (timemanager.f90:617) is not computed by this engine");
    if (cfg.iflux == 1) return fail(FPX_ERR_UNSUPPORTED, "iflux = 1: the particle loop's calcfluxes (timemanager.f90:623) is not computed by this engine");
    if (cfg.linit_cond >= 1) return fail(FPX_ERR_UNSUPPORTED, "linit_cond >= 1: the particle loop's initial_cond_calc (timemanager.f90:631,702) is not computed by this engine");
    if (cfg.lsettling && cfg.compute_real_bytes == 4 && (long long)cfg.nx * cfg.ny > (1ll << 24)) return fail(FPX_ERR_ARG, "lsettling with the f32 engine: nx*ny must not exceed 2^24 (the Langevin kernel keeps the column index of get_settling in an f32 stash slot)");
    if (cfg.blend_mode < 0 || cfg.blend_mode > 2) return fail(FPX_ERR_ARG, "blend_mode must be 0 (automatic), 1 (on) or 2 (off)");
    if (cfg.global_particles < 0) return fail(FPX_ERR_ARG, "global_particles must not be negative");
    if (cfg.pbl_slice_passes < -1) return fail(FPX_ERR_ARG, "pbl_slice_passes must be -1 (one launch), 0 (the engine's schedule) or a pass budget");
    HIPCHK(hipSetDevice(cfg.device));
    HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    memset(&V, 0, sizeof(V));
    memset(&P, 0, sizeof(P));
    memset(&Gp, 0, sizeof(Gp));
    // scalars: typed exactly as the reference's default-real variables
    V.nx = cfg.nx; V.ny = cfg.ny; V.nz = cfg.nz; V.nxmin1 = cfg.nx - 1; V.nymin1 = cfg.ny - 1; V.nmixz = cfg.nmixz;
    V.dx = (R)cfg.dx; V.dy = (R)cfg.dy; V.xlon0 = (R)cfg.xlon0; V.ylat0 = (R)cfg.ylat0;
    {  // gridcheck_ecmwf.f90:311-312 with par_mod.f90:59 r_earth, pi
      const R r_earth = (R)6.371e6, pi = (R)3.14159265;
      V.dxconst = (R)180. / (V.dx * r_earth * pi);
      V.dyconst = (R)180. / (V.dy * r_earth * pi);
    }
    V.xglobal = cfg.xglobal; V.nglobal = cfg.nglobal; V.sglobal = cfg.sglobal;
    V.switchnorthg = (R)cfg.switchnorthg; V.switchsouthg = (R)cfg.switchsouthg;
    for (int i = 0; i < 9; i++) { V.northpolemap[i] = (R)cfg.northpolemap[i]; V.southpolemap[i] = (R)cfg.southpolemap[i]; }
    V.polemaps = nullptr;   // allocated below, with the other device arrays
    V.ldirect = cfg.ldirect; V.lsynctime = cfg.lsynctime; V.method = cfg.method; V.mintime = cfg.mintime;
    V.ifine = cfg.ifine; V.turbswitch = cfg.turbswitch; V.cblflag = cfg.cblflag; V.mdomainfill = cfg.mdomainfill;
    V.lsettling = cfg.lsettling; V.nspec = cfg.nspec; V.drydep = cfg.drydep;
    V.turboff = cfg.turboff != 0; V.interpolhmix = cfg.interpolhmix != 0;
    V.pbl_cost_buckets = 0;   // set per step (Engine::step)
    V.ctl = (R)cfg.ctl; V.fine = (R)1. / (R)cfg.ifine;   // readcommand.f90:271
    V.d_trop = (R)cfg.d_trop; V.d_strat = (R)cfg.d_strat; V.turbmesoscale = (R)cfg.turbmesoscale;
    for (int i = 0; i < FPX_MAXSPEC; i++) {
      V.drydepspec[i] = cfg.drydepspec[i];
      V.density[i] = (R)cfg.density[i]; V.dquer[i] = (R)cfg.dquer[i]; V.vsetaver[i] = (R)cfg.vsetaver[i];
      V.cunningham[i] = (R)cfg.cunningham[i]; V.decay[i] = (R)cfg.decay[i];
    }
    V.lage_last = cfg.lage_last; V.mquasilag = cfg.mquasilag;
    V.numpoint = 0; V.rel_xmass = nullptr; V.rel_npart = nullptr; V.rel_nsp = nullptr;
    V.rng_mode = cfg.rng_mode; V.seed = cfg.seed; V.maxrand = 1000000;
    if (cfg.particle_base < 0 || cfg.particle_base + cfg.max_particles > 0xFFFFFFF0ll) return fail(FPX_ERR_ARG, "particle_base + max_particles must fit 32 bits");
    if (cfg.particle_base != 0 && cfg.rng_mode == FPX_RNG_TABLE_SEQ) return fail(FPX_ERR_ARG, "particle_base: the serial-stream RNG mode (TABLE_SEQ) is a single-rank mode");
    V.pid_base = (unsigned int)cfg.particle_base;
    V.eps = (R)(cfg.par_nxmax > 0 ? cfg.par_nxmax : cfg.nxmax) / (R)3.e5;   // advance.f90:107
    V.numbnests = 0;
    V.nest = nullptr;
    g_nx = cfg.nx; g_ny = cfg.ny; g_nxmax = cfg.nxmax; g_nymax = cfg.nymax;

    const size_t ncol = (size_t)cfg.nx * cfg.ny, nlev = ncol * cfg.nz;
    R *p;
    int rc;
    if ((rc = dalloc(&p, cfg.nz))) return rc; V.height = p;
    {
      R maps[18];
      for (int i = 0; i < 9; i++) { maps[i] = V.northpolemap[i]; maps[9 + i] = V.southpolemap[i]; }
      if ((rc = dalloc(&p, 18))) return rc;
      HIPCHK(hipMemcpy(p, maps, sizeof(maps), hipMemcpyHostToDevice));
      V.polemaps = p;
    }
    {
      R rows[kMaxSpec][4];
      for (int i = 0; i < kMaxSpec; i++) { rows[i][0] = V.density[i]; rows[i][1] = V.dquer[i]; rows[i][2] = V.vsetaver[i]; rows[i][3] = V.cunningham[i]; }
      if ((rc = dalloc(&p, kMaxSpec * 4))) return rc;
      HIPCHK(hipMemcpy(p, rows, sizeof(rows), hipMemcpyHostToDevice));
      V.spec = p;
    }
    if ((rc = dalloc(&p, nlev * 6))) return rc; V.w3 = p;
    HIPCHK(hipMemsetAsync(p, 0, nlev * 6 * sizeof(R), stream));
    if (cfg.nglobal || cfg.sglobal) {
      if ((rc = dalloc(&p, nlev * 6))) return rc; V.w3pol = p;
      HIPCHK(hipMemsetAsync(p, 0, nlev * 6 * sizeof(R), stream));
    }
    if ((rc = dalloc(&p, nlev * 4))) return rc; V.r2 = p;
    HIPCHK(hipMemsetAsync(p, 0, nlev * 4 * sizeof(R), stream));
    if ((rc = dalloc(&p, ncol * 8))) return rc; V.sfc = p;
    HIPCHK(hipMemsetAsync(p, 0, ncol * 8 * sizeof(R), stream));
    if ((rc = dalloc(&p, ncol))) return rc; V.hcell = p;
    if ((rc = dalloc(&p, ncol))) return rc; V.tropo = p;
    if (cfg.drydep) { if ((rc = dalloc(&p, ncol * 2 * cfg.nspec))) return rc; V.vdep = p; }
    if (cfg.lsettling) { if ((rc = dalloc(&p, nlev * 2))) return rc; V.rhott = p; }
    if ((rc = dalloc(&p, V.maxrand))) return rc; V.rannumb = p;
    if ((rc = dalloc(&d_stats, 1))) return rc;
    HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(Stats), stream));

    const size_t cap = (size_t)cfg.max_particles;
    P.cap = (long long)cap;
    if ((rc = dalloc(&P.xt, cap))) return rc;
    if ((rc = dalloc(&P.yt, cap))) return rc;
    R **rs[] = {&P.zt, &P.up, &P.vp, &P.wp, &P.us, &P.vs, &P.ws};
    for (auto q : rs) if ((rc = dalloc(q, cap))) return rc;
    int **is[] = {&P.idt, &P.itra1, &P.itramem, &P.npoint, &P.nclass, &P.itrasplit};
    for (auto q : is) if ((rc = dalloc(q, cap))) return rc;
    if ((rc = dalloc(&P.cbt, cap))) return rc;
    if ((rc = dalloc(&P.xmass1, cap * cfg.nspec))) return rc;
    if (cfg.drybkdep || cfg.wetbkdep) {
      // backward runs with receptor scavenging (readcommand.f90:320-340): xscav_frac1(maxpart,maxspec), -1 = not yet scavenged
      if (cfg.ldirect != -1) return fail(FPX_ERR_ARG, "drybkdep / wetbkdep are options of backward runs (ldirect = -1)");
      if (cfg.drybkdep && cfg.wetbkdep) return fail(FPX_ERR_ARG, "drybkdep and wetbkdep exclude each other (COMMAND ind_receptor 3 or 4)");
      if ((rc = dalloc(&P.xscav, cap * cfg.nspec))) return rc;
      k_fill<R><<<(int)((cap * cfg.nspec + kBlock - 1) / kBlock), kBlock, 0, stream>>>(P.xscav, (R)-1, 0, (long long)(cap * cfg.nspec), nullptr);
    }
    if ((rc = dalloc(&P.pid, cap))) return rc;
    if ((rc = dalloc(&d_pbl_list, cap))) return rc;
    if ((rc = dalloc(&d_pbl_ctr, kCtrWords))) return rc;
    if ((rc = dalloc(&d_pbl_flag, cap))) return rc;
    if ((rc = dalloc(&d_pbl_flag2, cap))) return rc;
    if ((rc = dalloc(&d_iota, cap))) return rc;
    memset(&Q, 0, sizeof(Q));
    {
      if ((rc = dalloc(&Q.rec, cap))) return rc;
      if (cfg.drydep && (rc = dalloc(&Q.tdep, cap))) return rc;
    }
    // every storage space starts zeroed (releaseparticles leaves the turbulent state of a space it takes as it finds it;
    // the host's arrays start from zero too), dead (FLEXPART.f90:315-317) and identity-numbered
    {
      R *rz[] = {P.zt, P.up, P.vp, P.wp, P.us, P.vs, P.ws};
      for (R *q : rz) HIPCHK(hipMemsetAsync(q, 0, cap * sizeof(R), stream));
      HIPCHK(hipMemsetAsync(P.xt, 0, cap * sizeof(double), stream));
      HIPCHK(hipMemsetAsync(P.yt, 0, cap * sizeof(double), stream));
      HIPCHK(hipMemsetAsync(P.xmass1, 0, cap * cfg.nspec * sizeof(R), stream));
      int *iz[] = {P.idt, P.itramem, P.npoint, P.nclass};
      for (int *q : iz) HIPCHK(hipMemsetAsync(q, 0, cap * sizeof(int), stream));
      HIPCHK(hipMemsetAsync(P.cbt, 0, cap * sizeof(short), stream));
    }

    const int nb = (int)((cap + kBlock - 1) / kBlock);
    k_fill<int><<<nb, kBlock, 0, stream>>>(P.itra1, kDead, 0, (long long)cap, nullptr);
    // "never split" carries the sign of the run's direction: the test is ldirect*itime >= ldirect*itrasplit (timemanager.f90:478)
    k_fill<int><<<nb, kBlock, 0, stream>>>(P.itrasplit, (cfg.ldirect < 0 ? -1 : 1) * 999999999, 0, (long long)cap, nullptr);
    k_iota_pid<<<nb, kBlock, 0, stream>>>(P.pid, 0, (long long)cap);
    k_iota_pid<<<nb, kBlock, 0, stream>>>(d_iota, 0, (long long)cap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  ~Engine() override {
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto &e : pinned_host) (void)hipHostUnregister(const_cast<void *>(e.first));
    for (auto &e : ev_pool) for (int i = 0; i < 5; i++) (void)hipEventDestroy(e.e[i]);
    for (void *q : owned) (void)hipFree(q);
    if (staging) (void)hipFree(staging);
    if (d_sort_tmp) (void)hipFree(d_sort_tmp);
    if (d_sel_tmp) (void)hipFree(d_sel_tmp);
    if (conv_scr) (void)hipFree(conv_scr);
    if (conv_scan_tmp) (void)hipFree(conv_scan_tmp);
    if (conv_alive) (void)hipFree(conv_alive);
    if (rel_flags_buf) (void)hipFree(rel_flags_buf);
    if (rel_rank_buf) (void)hipFree(rel_rank_buf);
    if (rel_tmp_buf) (void)hipFree(rel_tmp_buf);
    if (redist_dev) (void)hipFree(redist_dev);      // (d_w3t: dalloc'ed, freed with the engine's other owned buffers)
    if (d_sort_rec) (void)hipFree(d_sort_rec);
    if (red_pin) (void)hipHostFree(red_pin);
    if (comm) (void)ncclCommDestroy(comm);
    if (stream) (void)hipStreamDestroy(stream);
  }

  int set_height(const void *h, int n) override {
    if (!h || n != cfg.nz) return fail(FPX_ERR_ARG, "set_height: need nz values");
    std::vector<R> tmp(n);
    for (int k = 0; k < n; k++) tmp[k] = cfg.host_real_bytes == 4 ? (R)((const float *)h)[k] : (R)((const double *)h)[k];
    for (int k = 1; k < n; k++)
      if (!(tmp[k] > tmp[k - 1])) return fail(FPX_ERR_ARG, "set_height: heights must increase strictly");
    HIPCHK(hipMemcpyAsync((void *)V.height, tmp.data(), n * sizeof(R), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    height_host.resize(n);
    for (int k = 0; k < n; k++) height_host[k] = cfg.host_real_bytes == 4 ? (double)((const float *)h)[k] : ((const double *)h)[k];
    height_set = true;
    return 0;
  }

  // stage one host array and repack it
  // geometry of the host array being repacked (mother grid by default, a nest during nest uploads)
  int g_nx = 0, g_ny = 0, g_nxmax = 0, g_nymax = 0;
  template <typename H>
  int pack3(const void *host, R *out, int stride, int off) {
    const size_t n = (size_t)g_nxmax * g_nymax * cfg.nz;   // levels beyond nz are never read
    int rc = ensure_staging(n * sizeof(H));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(staging, host, n * sizeof(H), hipMemcpyHostToDevice, stream));
    dim3 grid((g_nx + 31) / 32, (cfg.nz + 31) / 32, g_ny), block(32, 8);
    k_pack3<H, R><<<grid, block, 0, stream>>>((const H *)staging, out, g_nx, g_ny, cfg.nz, g_nxmax, g_nymax, stride, off);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));   // staging is reused by the next field
    return 0;
  }
  template <typename H>
  int pack2(const void *host, R *out, int stride, int off) {
    const size_t n = (size_t)g_nxmax * g_nymax;
    int rc = ensure_staging(n * sizeof(H));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(staging, host, n * sizeof(H), hipMemcpyHostToDevice, stream));
    int tot = g_nx * g_ny;
    k_pack2<H, R><<<(tot + kBlock - 1) / kBlock, kBlock, 0, stream>>>((const H *)staging, out, g_nx, g_ny, g_nxmax, stride, off);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  int p3(const void *host, const R *out, int stride, int off) {
    return cfg.host_real_bytes == 4 ? pack3<float>(host, (R *)out, stride, off) : pack3<double>(host, (R *)out, stride, off);
  }
  int p2(const void *host, const R *out, int stride, int off) {
    return cfg.host_real_bytes == 4 ? pack2<float>(host, (R *)out, stride, off) : pack2<double>(host, (R *)out, stride, off);
  }

  int upload_fields(int slot, const fpx_fields *f) override {
    if (slot != 1 && slot != 2) return fail(FPX_ERR_ARG, "upload_fields: slot must be 1 or 2");
    if (!f || !f->uu || !f->vv || !f->ww || !f->rho || !f->drhodz || !f->hmix || !f->ustar || !f->wstar || !f->oli || !f->tropopause)
      return fail(FPX_ERR_ARG, "upload_fields: uu, vv, ww, rho, drhodz, hmix, ustar, wstar, oli, tropopause are required");
    if ((cfg.nglobal || cfg.sglobal) && (!f->uupol || !f->vvpol)) return fail(FPX_ERR_ARG, "upload_fields: uupol/vvpol required on a grid with poles");
    if (cfg.drydep && !f->vdep) return fail(FPX_ERR_ARG, "upload_fields: vdep required with DRYDEP");
    if (cfg.lsettling && !f->tt) return fail(FPX_ERR_ARG, "upload_fields: tt required with lsettling");
    const int s = slot - 1;
    int rc;
    if ((rc = p3(f->uu, V.w3, 6, s * 3 + 0))) return rc;
    if ((rc = p3(f->vv, V.w3, 6, s * 3 + 1))) return rc;
    if ((rc = p3(f->ww, V.w3, 6, s * 3 + 2))) return rc;
    if (V.w3pol) {
      if ((rc = p3(f->uupol, V.w3pol, 6, s * 3 + 0))) return rc;
      if ((rc = p3(f->vvpol, V.w3pol, 6, s * 3 + 1))) return rc;
      if ((rc = p3(f->ww, V.w3pol, 6, s * 3 + 2))) return rc;
    }
    if ((rc = p3(f->rho, V.r2, 4, s * 2 + 0))) return rc;
    if ((rc = p3(f->drhodz, V.r2, 4, s * 2 + 1))) return rc;
    if ((rc = p2(f->ustar, V.sfc, 8, s * 4 + 0))) return rc;
    if ((rc = p2(f->wstar, V.sfc, 8, s * 4 + 1))) return rc;
    if ((rc = p2(f->oli, V.sfc, 8, s * 4 + 2))) return rc;
    if ((rc = p2(f->hmix, V.sfc, 8, s * 4 + 3))) return rc;
    if ((rc = diag_alloc()) || (rc = diag2_from_host(diag_tropo[s], f->tropopause, DG_TROPO + s))) return rc;   // partoutput interpolates it in time
    if (slot == 1) {   // literal time index 1 uses: advance.f90:253, get_settling.f90:83-84
      if ((rc = p2(f->tropopause, V.tropo, 1, 0))) return rc;
      if (V.rhott) {
        if ((rc = p3(f->rho, V.rhott, 2, 0))) return rc;
        if ((rc = p3(f->tt, V.rhott, 2, 1))) return rc;
      }
    }
    if (V.vdep) {
      const size_t plane = (size_t)cfg.nxmax * cfg.nymax * cfg.host_real_bytes;
      for (int ks = 0; ks < cfg.nspec; ks++)
        if ((rc = p2((const char *)f->vdep + plane * ks, V.vdep, 2 * cfg.nspec, s * cfg.nspec + ks))) return rc;
    }
    int tot = cfg.nx * cfg.ny;
    k_hcell<R><<<(tot + kBlock - 1) / kBlock, kBlock, 0, stream>>>(V.sfc, (R *)V.hcell, cfg.nx, cfg.ny);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    slot_loaded[s] = true;
    return 0;
  }


  // ---- verttransform_ecmwf on the device --------------------------------
  // Model-level input -> z-level fields, computed in the host's real kind H in the host's own
  // array layout on the device, then repacked like an upload (the 3-D fields never cross PCIe
  // as z-level arrays unless the host asks for them back through `out`).
  std::vector<double> height_host;     // height(nz) in the host's real kind, widened
  // host arrays registered for DMA on request (fpx_model_levels.pin_host): address -> bytes; released in the destructor
  std::vector<std::pair<const void *, size_t>> pinned_host;
  void pin_host_range(const void *p, size_t bytes) {
    for (auto &e : pinned_host) if (e.first == p && e.second >= bytes) return;
    if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) == hipSuccess) pinned_host.emplace_back(p, bytes);
    else (void)hipGetLastError();      // not fatal: the copy falls back to the pageable path
  }
  void *vt_sets[1 + kMaxNests][32] = {};   // device arrays of the transform per grid (0 = mother, l = nest l), allocated on first use
  bool vt_set_ready[1 + kMaxNests] = {};

  template <typename H>
  static H vt_ew_host(H x) {           // ew.f90:4-29
    H y = (H)373.16 / x;
    H a = (H)-7.90298 * (y - (H)1.);
    a = a + ((H)5.02808 * (H)0.43429 * std::log(y));
    H c = ((H)1. - ((H)1. / y)) * (H)11.344;
    c = (H)-1. + std::pow((H)10., c);
    c = (H)-1.3816 * c / (H)1.e7;
    H d = ((H)1. - y) * (H)3.49149;
    d = (H)-1. + std::pow((H)10., d);
    d = (H)8.1328 * d / (H)1.e3;
    y = a + c + d;
    return (H)101324.6 * std::pow((H)10., y);
  }

  // first call: z levels from the first column with ps > 1000 hPa, verttransform_ecmwf.f90:134-187
  template <typename H>
  int vt_init_height(const fpx_model_levels *m, std::vector<H> &hgt, int &nmixz) {
    const H *ps = (const H *)m->ps, *tt2 = (const H *)m->tt2, *td2 = (const H *)m->td2;
    const H *tth = (const H *)m->tth, *qvh = (const H *)m->qvh, *akz = (const H *)m->akz, *bkz = (const H *)m->bkz;
    const size_t sx = (size_t)cfg.nxmax, sxy = (size_t)cfg.nxmax * cfg.nymax;
    long long col = -1;
    for (int jy = 0; jy < cfg.ny && col < 0; jy++)
      for (int ix = 0; ix < cfg.nx; ix++)
        if (ps[ix + sx * jy] > (H)100000.) { col = (long long)(ix + sx * jy); break; }
    if (col < 0) return fail(FPX_ERR_ARG, "verttransform: no column with ps > 100000 Pa to build the z levels from");
    const H konst = (H)287.05 / (H)9.81;
    H tvold = tt2[col] * ((H)1. + (H)0.378 * vt_ew_host<H>(td2[col]) / ps[col]);
    H pold = ps[col];
    hgt.assign(cfg.nz, (H)0);
    for (int kz = 2; kz <= m->nuvz; kz++) {
      const H pint = akz[kz - 1] + bkz[kz - 1] * ps[col];
      const H tv = tth[col + sxy * (kz - 1)] * ((H)1. + (H)0.608 * qvh[col + sxy * (kz - 1)]);
      if (std::abs(tv - tvold) > (H)0.2) hgt[kz - 1] = hgt[kz - 2] + konst * std::log(pold / pint) * (tv - tvold) / std::log(tv / tvold);
      else hgt[kz - 1] = hgt[kz - 2] + konst * std::log(pold / pint) * tv;
      tvold = tv;
      pold = pint;
    }
    nmixz = cfg.nz;
    for (int kz = 1; kz <= cfg.nz; kz++)
      if (hgt[kz - 1] > (H)4500.) { nmixz = kz; break; }   // hmixmax, par_mod.f90:77
    return 0;
  }

  template <typename H>
  int verttransform_t(int nest, int slot, const fpx_model_levels *m, const fpx_fields *sfc, const fpx_fields_out *out) {
    // nest = 0: the mother grid (verttransform_ecmwf.f90); nest = l >= 1: nested grid l (verttransform_nests.f90: the same
    // algorithm on the nest's arrays -- no height initialisation, no polar caps, the nest's own dyn/ylat0n in cosf and
    // the mother's dxconst, dyconst times xresoln, yresoln in the slope term, :346,384-385)
    const int nz = cfg.nz;
    const int gnx = nest ? h_nest[nest - 1].nx : cfg.nx, gny = nest ? h_nest[nest - 1].ny : cfg.ny;
    const int gnxmax = nest ? nest_nxmaxn : cfg.nxmax, gnymax = nest ? nest_nymaxn : cfg.nymax;
    void **vt_dev = vt_sets[nest];
    bool &vt_ready = vt_set_ready[nest];
    const size_t n2 = (size_t)gnxmax * gnymax, n3 = n2 * nz;
    int rc;
    enum { UUH, VVH, PVH, WWH, TTH, QVH, PS, TT2, TD2, AKZ, BKZ, AKN, BKN, HGT,
           UU, VV, WW, TT, QV, PV, RHO, DRHO, UPOL, VPOL, UVZ, WZ, RHOH, PINM, KUV, KW, NBUF };
    if (!vt_ready) {
      for (int i = 0; i < NBUF; i++) {
        const size_t n = (i >= PS && i <= TD2) ? n2 : (i >= AKZ && i <= HGT) ? (size_t)nz : n3;
        H *q = nullptr;
        if ((rc = dalloc(&q, n))) return rc;
        HIPCHK(hipMemsetAsync(q, 0, n * sizeof(H), stream));
        vt_dev[i] = q;
      }
      vt_ready = true;
    }
    auto D = [&](int i) { return (H *)vt_dev[i]; };
    // z levels: derived on the first call (or on request), else the ones the host has set
    if (!nest && (m->init || !height_set)) {
      std::vector<H> hgt;
      int nmixz = 0;
      if ((rc = vt_init_height<H>(m, hgt, nmixz))) return rc;
      if ((rc = set_height(hgt.data(), nz))) return rc;
      V.nmixz = nmixz;
      cfg.nmixz = nmixz;
    }
    {
      std::vector<H> hh(nz);
      for (int k = 0; k < nz; k++) hh[k] = (H)height_host[k];
      HIPCHK(hipMemcpyAsync(D(HGT), hh.data(), nz * sizeof(H), hipMemcpyHostToDevice, stream));
      HIPCHK(hipStreamSynchronize(stream));
    }
    const void *src[13] = {m->uuh, m->vvh, m->pvh, m->wwh, m->tth, m->qvh, m->ps, m->tt2, m->td2, m->akz, m->bkz, m->aknew, m->bknew};
    for (int i = 0; i < 13; i++) {
      const size_t n = (i >= PS && i <= TD2) ? n2 : (i >= AKZ) ? (size_t)nz : n3;
      if (m->pin_host && n >= n2) pin_host_range(src[i], n * sizeof(H));
      HIPCHK(hipMemcpyAsync(D(i), src[i], n * sizeof(H), hipMemcpyHostToDevice, stream));
    }
    vt::Geo<H> G;
    G.nx = gnx; G.ny = gny; G.nz = nz; G.nuvz = m->nuvz; G.nwz = m->nwz; G.nxmax = gnxmax; G.nymax = gnymax;
    G.dx = (H)cfg.dx; G.dy = (H)cfg.dy; G.xlon0 = (H)cfg.xlon0; G.ylat0 = (H)cfg.ylat0;
    {   // gridcheck_ecmwf.f90:311-312, in the host's real kind
      const H pi = (H)3.14159265, r_earth = (H)6.371e6;
      G.dxconst = (H)180. / (G.dx * r_earth * pi);
      G.dyconst = (H)180. / (G.dy * r_earth * pi);
    }
    G.xres = (H)1.; G.yres = (H)1.;
    if (nest) {   // after dxconst, dyconst (the mother's): the nest's own spacing and origin for cosf
      G.dy = (H)m->nest_dy; G.ylat0 = (H)m->nest_ylat0;
      G.xres = (H)h_nest[nest - 1].xres; G.yres = (H)h_nest[nest - 1].yres;
    }
    G.nglobal = nest ? 0 : cfg.nglobal; G.sglobal = nest ? 0 : cfg.sglobal;
    G.switchnorthg = (H)cfg.switchnorthg; G.switchsouthg = (H)cfg.switchsouthg;
    for (int i = 0; i < 9; i++) { G.northpolemap[i] = (H)cfg.northpolemap[i]; G.southpolemap[i] = (H)cfg.southpolemap[i]; }
    vt::In<H> I{D(UUH), D(VVH), D(PVH), D(WWH), D(TTH), D(QVH), D(PS), D(TT2), D(TD2), D(AKZ), D(BKZ), D(AKN), D(BKN), D(HGT)};
    vt::Out<H> O{D(UU), D(VV), D(WW), D(TT), D(QV), D(PV), D(RHO), D(DRHO), D(UPOL), D(VPOL), D(UVZ), D(WZ), D(RHOH), D(PINM)};
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, stream));
    const int ncol = gnx * gny, nb = (ncol + 255) / 256;
    const size_t sm = (size_t)nz * sizeof(H);
    const dim3 g3(nb, nz);
    unsigned short *kuv = (unsigned short *)vt_dev[KUV], *kw = (unsigned short *)vt_dev[KW];
    // fused path: the ECMWF level structure (nuvz = nwz = nz) and a tile's uvzlev fits the LDS of a CU twice
    constexpr int kVtWaves = FPX_VT_WAVES;
    const size_t sm_lev = (size_t)nz * vt::kVtCols * sizeof(H), sm_fused = sm_lev + (size_t)5 * nz * sizeof(H);
    const bool fused = m->nuvz == nz && m->nwz == nz && nz >= 3 && sm_fused <= (size_t)80 * 1024 && n3 < ((size_t)1 << 31) && !opt.vt_unfused;
    if (fused) {
      vt::Tiles T;
      T.tiles_y = (gny + vt::kVtTy - 1) / vt::kVtTy;
      T.ntiles = T.tiles_y * ((gnx + vt::kVtTx - 1) / vt::kVtTx);
      T.tiles_per_xcd = (T.ntiles + 7) / 8;
      static bool attr_set = false;
      if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void *)vt::k_vt_levels<H>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)vt::k_vt_fused<H, kVtWaves>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr_set = true;
      }
      vt::k_vt_levels<H><<<8 * T.tiles_per_xcd, 512, sm_lev, stream>>>(G, I, O, T);
      vt::k_vt_fused<H, kVtWaves><<<8 * T.tiles_per_xcd, kVtWaves * 64, sm_fused, stream>>>(G, I, O, T);
    } else {
      vt::k_vt_inc<H><<<g3, 256, 0, stream>>>(G, I, O);
      vt::k_vt_column<H><<<nb, 256, 0, stream>>>(G, I, O);
      vt::k_vt_search<H><<<dim3(nb, 2), 256, sm, stream>>>(G, I, O, kuv, kw);
      vt::k_vt_fill<H><<<g3, 256, 0, stream>>>(G, I, O, kuv, kw);
      vt::k_vt_post<H><<<g3, 256, 0, stream>>>(G, I, O, kuv);
    }
    const size_t smrow = (size_t)(gnx + 3) * sizeof(H);
    if (G.nglobal) {
      const int jy0 = std::max(0, (int)G.switchnorthg - 2), jy1 = cfg.ny - 1;
      if (jy1 >= jy0) vt::k_vt_polar<H><<<dim3((cfg.nx + 255) / 256, jy1 - jy0 + 1, nz), 256, 0, stream>>>(G, O, jy0, jy1, 0);
      vt::k_vt_polerow<H><<<nz, 64, smrow, stream>>>(G, O, 0);
    }
    if (G.sglobal) {
      const int jy0 = 0, jy1 = std::min(cfg.ny - 1, (int)G.switchsouthg + 3);
      if (jy1 >= jy0) vt::k_vt_polar<H><<<dim3((cfg.nx + 255) / 256, jy1 - jy0 + 1, nz), 256, 0, stream>>>(G, O, jy0, jy1, 1);
      vt::k_vt_polerow<H><<<nz, 64, smrow, stream>>>(G, O, 1);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, stream));
    // repack into the gather layout, as upload_fields / upload_nest_fields do from the staged host arrays
    const int s = slot - 1;
    auto pk = [&](const H *in, const R *outp, int stride, int off) -> int {
      dim3 grid((gnx + 31) / 32, (nz + 31) / 32, gny), block(32, 8);
      k_pack3<H, R><<<grid, block, 0, stream>>>(in, (R *)outp, gnx, gny, nz, gnxmax, gnymax, stride, off);
      HIPCHK(hipGetLastError());
      return 0;
    };
    const R *t_w3 = nest ? h_nest[nest - 1].w3 : V.w3, *t_r2 = nest ? h_nest[nest - 1].r2 : V.r2;
    const R *t_sfc = nest ? h_nest[nest - 1].sfc : V.sfc, *t_tropo = nest ? h_nest[nest - 1].tropo : V.tropo;
    const R *t_vdep = nest ? h_nest[nest - 1].vdep : V.vdep, *t_hcell = nest ? h_nest[nest - 1].hcell : V.hcell;
    if ((rc = pk(D(UU), t_w3, 6, s * 3 + 0)) || (rc = pk(D(VV), t_w3, 6, s * 3 + 1)) || (rc = pk(D(WW), t_w3, 6, s * 3 + 2))) return rc;
    if (!nest && V.w3pol)
      if ((rc = pk(D(UPOL), V.w3pol, 6, s * 3 + 0)) || (rc = pk(D(VPOL), V.w3pol, 6, s * 3 + 1)) || (rc = pk(D(WW), V.w3pol, 6, s * 3 + 2))) return rc;
    if ((rc = pk(D(RHO), t_r2, 4, s * 2 + 0)) || (rc = pk(D(DRHO), t_r2, 4, s * 2 + 1))) return rc;
    if (!nest && slot == 1 && V.rhott)
      if ((rc = pk(D(RHO), V.rhott, 2, 0)) || (rc = pk(D(TT), V.rhott, 2, 1))) return rc;
    if (!nest && wet_on && Wp.ttw) { if ((rc = pk(D(TT), Wp.ttw, 2, s))) return rc; }
    // the 2-D fields calcpar / calcpar_nests leave on the host
    g_nx = gnx; g_ny = gny; g_nxmax = gnxmax; g_nymax = gnymax;
    rc = 0;
    if (sfc) do {
      if ((rc = p2(sfc->ustar, t_sfc, 8, s * 4 + 0)) || (rc = p2(sfc->wstar, t_sfc, 8, s * 4 + 1)) ||
          (rc = p2(sfc->oli, t_sfc, 8, s * 4 + 2)) || (rc = p2(sfc->hmix, t_sfc, 8, s * 4 + 3))) break;
      if (slot == 1 && (rc = p2(sfc->tropopause, t_tropo, 1, 0))) break;
      if (t_vdep) {
        const size_t plane = n2 * cfg.host_real_bytes;
        for (int ks = 0; ks < cfg.nspec && !rc; ks++) rc = p2((const char *)sfc->vdep + plane * ks, t_vdep, 2 * cfg.nspec, s * cfg.nspec + ks);
      }
    } while (0);
    g_nx = cfg.nx; g_ny = cfg.ny; g_nxmax = cfg.nxmax; g_nymax = cfg.nymax;
    if (rc) return rc;
    if (!nest) {
      if ((rc = diag_alloc())) return rc;
      if (sfc && (rc = diag2_from_host(diag_tropo[s], sfc->tropopause, DG_TROPO + s))) return rc;
      // pv, qv, tt of this slot stay on the device for partoutput
      if ((rc = diag3(D(PV), true, 0, s)) || (rc = diag3(D(QV), true, 1, s)) || (rc = diag3(D(TT), true, 2, s))) return rc;
    }
    if (sfc) k_hcell<R><<<(gnx * gny + kBlock - 1) / kBlock, kBlock, 0, stream>>>(t_sfc, (R *)t_hcell, gnx, gny);
    HIPCHK(hipGetLastError());
    if (out) {   // z-level arrays the host still wants (partoutput, convection, cloud diagnostics ...)
      void *dst[10] = {out->uu, out->vv, out->ww, out->tt, out->qv, out->pv, out->rho, out->drhodz, out->uupol, out->vvpol};
      for (int i = 0; i < 10; i++)
        if (dst[i]) HIPCHK(hipMemcpyAsync(dst[i], D(UU + i), n3 * sizeof(H), hipMemcpyDeviceToHost, stream));
      if (out->height) for (int k = 0; k < nz; k++) ((H *)out->height)[k] = (H)height_host[k];
      if (out->nmixz) *out->nmixz = V.nmixz;
    }
    HIPCHK(hipStreamSynchronize(stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    vt_last_ms = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    // without sfc the slot's 2-D fields are still those of the previous wind field until fpx_calcpar has run: not loaded
    if (nest) nest_loaded[nest - 1][s] = true; else slot_loaded[s] = sfc != nullptr;
    if (!nest) vt_slot_on_device = slot;      // whose model-level arrays the buffers hold (fpx_calcpar)
    return 0;
  }
  int vt_slot_on_device = 0;

  // ---- calcpar on the device ----------------------------------------------------------
  template <typename H>
  int calcpar_t(int slot, const fpx_calcpar_in *c, const fpx_calcpar_out *out) {
    enum { UUH, VVH, PVH, WWH, TTH, QVH, PS, TT2, TD2, AKZ, BKZ, AKN, BKN, HGT,
           UU, VV, WW, TT, QV, PV, RHO, DRHO, UPOL, VPOL, UVZ, WZ, RHOH, PINM, KUV, KW, NBUF };
    void **vt_dev = vt_sets[0];
    auto D = [&](int i) { return (H *)vt_dev[i]; };
    const int nz = cfg.nz, s = slot - 1;
    const size_t n2 = (size_t)cfg.nxmax * cfg.nymax;
    int rc;
    if (!cp_buf) {
      H *q = nullptr;
      if ((rc = dalloc(&q, 8 * n2 + 2 * (size_t)nz))) return rc;       // surfstr, sshf, excessoro, ustar, wstar, oli, hmix, tropopause | akm, bkm
      HIPCHK(hipMemsetAsync(q, 0, (8 * n2 + 2 * (size_t)nz) * sizeof(H), stream));
      cp_buf = q;
    }
    H *B = (H *)cp_buf;
    H *d_str = B, *d_shf = B + n2, *d_exc = B + 2 * n2, *d_ust = B + 3 * n2, *d_wst = B + 4 * n2, *d_oli = B + 5 * n2, *d_hmix = B + 6 * n2, *d_akm = B + 8 * n2, *d_bkm = d_akm + nz;
    H *d_trop = (H *)diag_tropo[s];            // the slot's tropopause in the host's layout (kept for partoutput); stale values persist as in the reference
    if ((rc = diag_alloc())) return rc;
    d_trop = (H *)diag_tropo[s];
    HIPCHK(hipMemcpyAsync(d_str, c->surfstr, n2 * sizeof(H), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_shf, c->sshf, n2 * sizeof(H), hipMemcpyHostToDevice, stream));
    if (c->lsubgrid == 1) HIPCHK(hipMemcpyAsync(d_exc, c->excessoro, n2 * sizeof(H), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_akm, c->akm, (size_t)nz * sizeof(H), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_bkm, c->bkm, (size_t)nz * sizeof(H), hipMemcpyHostToDevice, stream));
    cp::Args<H> A;
    A.G.nx = cfg.nx; A.G.ny = cfg.ny; A.G.nz = nz; A.G.nuvz = nz; A.G.nwz = nz; A.G.nxmax = cfg.nxmax; A.G.nymax = cfg.nymax;
    A.G.dx = (H)cfg.dx; A.G.dy = (H)cfg.dy; A.G.xlon0 = (H)cfg.xlon0; A.G.ylat0 = (H)cfg.ylat0;
    A.I = vt::In<H>{D(UUH), D(VVH), D(PVH), D(WWH), D(TTH), D(QVH), D(PS), D(TT2), D(TD2), D(AKZ), D(BKZ), D(AKN), D(BKN), D(HGT)};
    A.surfstr = d_str; A.sshf = d_shf; A.excessoro = d_exc; A.akm = d_akm; A.bkm = d_bkm; A.lsubgrid = c->lsubgrid;
    A.zlev = D(RHOH);                          // scratch of the transform, free between calls
    A.ustar = d_ust; A.wstar = d_wst; A.oli = d_oli; A.hmix = d_hmix; A.tropopause = d_trop;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, stream));
    cp::k_calcpar<H><<<(cfg.nx * cfg.ny + 255) / 256, 256, 0, stream>>>(A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, stream));
    // into the gather packs, as upload_fields does from the host's arrays
    auto pk2 = [&](const H *in, const R *outp, int stride, int off) -> int {
      const int tot = cfg.nx * cfg.ny;
      k_pack2<H, R><<<(tot + kBlock - 1) / kBlock, kBlock, 0, stream>>>(in, (R *)outp, cfg.nx, cfg.ny, cfg.nxmax, stride, off);
      HIPCHK(hipGetLastError());
      return 0;
    };
    if ((rc = pk2(d_ust, V.sfc, 8, s * 4 + 0)) || (rc = pk2(d_wst, V.sfc, 8, s * 4 + 1)) || (rc = pk2(d_oli, V.sfc, 8, s * 4 + 2)) || (rc = pk2(d_hmix, V.sfc, 8, s * 4 + 3))) return rc;
    if (slot == 1 && (rc = pk2(d_trop, V.tropo, 1, 0))) return rc;        // literal time index 1, advance.f90:253
    diag_have[DG_TROPO + s] = true;
    if (V.vdep) {
      const size_t plane = n2 * cfg.host_real_bytes;
      for (int ks = 0; ks < cfg.nspec; ks++)
        if ((rc = p2((const char *)c->vdep + plane * ks, V.vdep, 2 * cfg.nspec, s * cfg.nspec + ks))) return rc;
    }
    k_hcell<R><<<(cfg.nx * cfg.ny + kBlock - 1) / kBlock, kBlock, 0, stream>>>(V.sfc, (R *)V.hcell, cfg.nx, cfg.ny);
    HIPCHK(hipGetLastError());
    if (out) {
      void *dst[5] = {out->ustar, out->wstar, out->oli, out->hmix, out->tropopause};
      const H *src[5] = {d_ust, d_wst, d_oli, d_hmix, d_trop};
      for (int i = 0; i < 5; i++)
        if (dst[i]) HIPCHK(hipMemcpyAsync(dst[i], src[i], n2 * sizeof(H), hipMemcpyDeviceToHost, stream));
    }
    HIPCHK(hipStreamSynchronize(stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    cp_last_ms = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    slot_loaded[s] = true;
    return 0;
  }
  void *cp_buf = nullptr;
  double cp_last_ms = 0;
  int calcpar(int slot, const fpx_calcpar_in *c, const fpx_calcpar_out *out) override {
    if (slot != 1 && slot != 2) return fail(FPX_ERR_ARG, "calcpar: slot must be 1 or 2");
    if (!c || !c->surfstr || !c->sshf || !c->akm || !c->bkm) return fail(FPX_ERR_ARG, "calcpar: surfstr, sshf, akm, bkm are required");
    if (c->lsubgrid == 1 && !c->excessoro) return fail(FPX_ERR_ARG, "calcpar: excessoro required with lsubgrid = 1");
    if (cfg.drydep && !c->vdep) return fail(FPX_ERR_ARG, "calcpar: vdep (the host's getvdep) required with DRYDEP");
    if (!vt_set_ready[0] || vt_slot_on_device != slot) return fail(FPX_ERR_STATE, "calcpar: fpx_verttransform_ecmwf of this slot first (its model-level arrays are read on the device)");
    return cfg.host_real_bytes == 4 ? calcpar_t<float>(slot, c, out) : calcpar_t<double>(slot, c, out);
  }
  double vt_last_ms = 0, po_last_ms = 0;
  double po_ms() override { return po_last_ms; }
  double vt_ms() override { return vt_last_ms; }
  double cp_ms() override { return cp_last_ms; }

  int verttransform(int slot, const fpx_model_levels *m, const fpx_fields *sfc, const fpx_fields_out *out) override {
    if (slot != 1 && slot != 2) return fail(FPX_ERR_ARG, "verttransform: slot must be 1 or 2");
    if (!m || !m->uuh || !m->vvh || !m->pvh || !m->wwh || !m->tth || !m->qvh || !m->ps || !m->tt2 || !m->td2 || !m->akz || !m->bkz || !m->aknew || !m->bknew)
      return fail(FPX_ERR_ARG, "verttransform: uuh, vvh, pvh, wwh, tth, qvh, ps, tt2, td2, akz, bkz, aknew, bknew are required");
    if (m->nuvz != cfg.nz || m->nwz != cfg.nz) return fail(FPX_ERR_ARG, "verttransform: nuvz = nwz = nz expected (gridcheck_ecmwf.f90 sets them equal)");
    if (cfg.nz < 3 || cfg.nz > 65535 || cfg.ny > 65535) return fail(FPX_ERR_ARG, "verttransform: 3 <= nz <= 65535, ny <= 65535");
    if (sfc && (!sfc->hmix || !sfc->ustar || !sfc->wstar || !sfc->oli || !sfc->tropopause)) return fail(FPX_ERR_ARG, "verttransform: the 2-D fields hmix, ustar, wstar, oli, tropopause are required (or sfc = NULL and fpx_calcpar)");
    if (sfc && cfg.drydep && !sfc->vdep) return fail(FPX_ERR_ARG, "verttransform: vdep required with DRYDEP");
    return cfg.host_real_bytes == 4 ? verttransform_t<float>(0, slot, m, sfc, out) : verttransform_t<double>(0, slot, m, sfc, out);
  }
  int verttransform_nest(int nest, int slot, const fpx_model_levels *m, const fpx_fields *sfc, const fpx_fields_out *out) override {
    if (nest < 1 || nest > V.numbnests) return fail(FPX_ERR_ARG, "verttransform_nest: nest out of range (fpx_nests_init first)");
    if (slot != 1 && slot != 2) return fail(FPX_ERR_ARG, "verttransform_nest: slot must be 1 or 2");
    if (!height_set) return fail(FPX_ERR_STATE, "verttransform_nest: the z levels come from the mother grid's first transform (or fpx_set_height)");
    if (!m || !m->uuh || !m->vvh || !m->pvh || !m->wwh || !m->tth || !m->qvh || !m->ps || !m->tt2 || !m->td2 || !m->akz || !m->bkz || !m->aknew || !m->bknew)
      return fail(FPX_ERR_ARG, "verttransform_nest: uuhn, vvhn, pvhn, wwhn, tthn, qvhn, psn, tt2n, td2n, akz, bkz, aknew, bknew are required");
    if (m->nuvz != cfg.nz || m->nwz != cfg.nz) return fail(FPX_ERR_ARG, "verttransform_nest: nuvz = nwz = nz expected");
    if (!(m->nest_dy > 0)) return fail(FPX_ERR_ARG, "verttransform_nest: nest_dy (dyn) and nest_ylat0 (ylat0n) of fpx_model_levels are required");
    if (!sfc || !sfc->hmix || !sfc->ustar || !sfc->wstar || !sfc->oli || !sfc->tropopause) return fail(FPX_ERR_ARG, "verttransform_nest: hmixn, ustarn, wstarn, olin, tropopausen are required");
    if (cfg.drydep && !sfc->vdep) return fail(FPX_ERR_ARG, "verttransform_nest: vdepn required with DRYDEP");
    return cfg.host_real_bytes == 4 ? verttransform_t<float>(nest, slot, m, sfc, out) : verttransform_t<double>(nest, slot, m, sfc, out);
  }


  // ---- partoutput -------------------------------------------------------
  void *diag_oro = nullptr, *diag_tropo[2] = {nullptr, nullptr}, *diag_d3 = nullptr;   // in the host's real kind
  bool diag_have[9] = {};          // oro | pv, qv, tt of slot 1, 2 | tropopause of slot 1, 2
  enum { DG_ORO = 0, DG_PV = 1, DG_QV = 3, DG_TT = 5, DG_TROPO = 7 };
  int diag_alloc() {
    int rc;
    const size_t n2 = (size_t)cfg.nxmax * cfg.nymax * cfg.host_real_bytes;
    char *p = nullptr;
    if (!diag_oro) { if ((rc = dalloc(&p, n2))) return rc; diag_oro = p; HIPCHK(hipMemsetAsync(p, 0, n2, stream)); }
    for (int m = 0; m < 2; m++)
      if (!diag_tropo[m]) { if ((rc = dalloc(&p, n2))) return rc; diag_tropo[m] = p; HIPCHK(hipMemsetAsync(p, 0, n2, stream)); }
    if (!diag_d3) {
      const size_t n = (size_t)cfg.nx * cfg.ny * cfg.nz * 6 * cfg.host_real_bytes;
      if ((rc = dalloc(&p, n))) return rc;
      diag_d3 = p;
      HIPCHK(hipMemsetAsync(p, 0, n, stream));
    }
    return 0;
  }
  int diag2_from_host(void *dev, const void *host, int have) {
    HIPCHK(hipMemcpyAsync(dev, host, (size_t)cfg.nxmax * cfg.nymax * cfg.host_real_bytes, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    diag_have[have] = true;
    return 0;
  }
  // one 3-D field in the host's layout (host memory, or device memory when `on_device`) -> component c of slot s of d3
  template <typename H>
  int diag3_pack(const void *src, bool on_device, int c, int s) {
    const size_t n = (size_t)cfg.nxmax * cfg.nymax * cfg.nz;
    const H *in = (const H *)src;
    if (!on_device) {
      int rc = ensure_staging(n * sizeof(H));
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(staging, src, n * sizeof(H), hipMemcpyHostToDevice, stream));
      in = (const H *)staging;
    }
    dim3 grid((cfg.nx + 31) / 32, (cfg.nz + 31) / 32, cfg.ny), block(32, 8);
    k_pack3<H, H><<<grid, block, 0, stream>>>(in, (H *)diag_d3, cfg.nx, cfg.ny, cfg.nz, cfg.nxmax, cfg.nymax, 6, s * 3 + c);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    diag_have[DG_PV + 2 * c + s] = true;
    return 0;
  }
  int diag3(const void *src, bool on_device, int c, int s) {
    return cfg.host_real_bytes == 4 ? diag3_pack<float>(src, on_device, c, s) : diag3_pack<double>(src, on_device, c, s);
  }
  // oron and ttn of a nested wind field for releaseparticles (:216-273): compact copies in the host's real kind
  void *rel_nest_oro[kMaxNests] = {}, *rel_nest_tt2[kMaxNests] = {};
  template <typename H>
  int upload_diag_nest_fields_t(int nest, int slot, const fpx_diag_fields *f) {
    const int l = nest - 1, nxn = h_nest[l].nx, nyn = h_nest[l].ny;
    const size_t n2 = (size_t)nxn * nyn, n2max = (size_t)nest_nxmaxn * nest_nymaxn;
    int rc;
    if (f->oro) {
      if (!rel_nest_oro[l]) { H *q = nullptr; if ((rc = dalloc(&q, n2))) return rc; rel_nest_oro[l] = q; }
      if ((rc = ensure_staging(n2max * sizeof(H)))) return rc;
      HIPCHK(hipMemcpyAsync(staging, f->oro, n2max * sizeof(H), hipMemcpyHostToDevice, stream));
      k_conv_pack<H, H><<<(int)((n2 + kBlock - 1) / kBlock), kBlock, 0, stream>>>((const H *)staging, (H *)rel_nest_oro[l], nxn, nyn, 1, nest_nxmaxn, nest_nymaxn);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(stream));
    }
    if (slot == 2 && f->tt) {
      const size_t n3 = n2 * cfg.nz;
      if (!rel_nest_tt2[l]) { H *q = nullptr; if ((rc = dalloc(&q, n3))) return rc; rel_nest_tt2[l] = q; }
      if ((rc = ensure_staging(n2max * cfg.nz * sizeof(H)))) return rc;
      HIPCHK(hipMemcpyAsync(staging, f->tt, n2max * cfg.nz * sizeof(H), hipMemcpyHostToDevice, stream));
      k_conv_pack<H, H><<<(int)((n3 + kBlock - 1) / kBlock), kBlock, 0, stream>>>((const H *)staging, (H *)rel_nest_tt2[l], nxn, nyn, cfg.nz, nest_nxmaxn, nest_nymaxn);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(stream));
    }
    return 0;
  }
  int upload_diag_nest_fields(int nest, int slot, const fpx_diag_fields *f) override {
    if (nest < 1 || nest > V.numbnests) return fail(FPX_ERR_ARG, "upload_diag_nest_fields: nest out of range (fpx_nests_init first)");
    if (!f || slot < 0 || slot > 2) return fail(FPX_ERR_ARG, "upload_diag_nest_fields: slot 0 (oron only), 1 or 2");
    return cfg.host_real_bytes == 4 ? upload_diag_nest_fields_t<float>(nest, slot, f) : upload_diag_nest_fields_t<double>(nest, slot, f);
  }

  int upload_diag_fields(int slot, const fpx_diag_fields *f) override {
    if (!f || slot < 0 || slot > 2) return fail(FPX_ERR_ARG, "upload_diag_fields: slot 0 (oro only), 1 or 2");
    int rc;
    if ((rc = diag_alloc())) return rc;
    if (f->oro && (rc = diag2_from_host(diag_oro, f->oro, DG_ORO))) return rc;
    if (slot == 0) return 0;
    const int s = slot - 1;
    if (f->pv && (rc = diag3(f->pv, false, 0, s))) return rc;
    if (f->qv && (rc = diag3(f->qv, false, 1, s))) return rc;
    if (f->tt && (rc = diag3(f->tt, false, 2, s))) return rc;
    return 0;
  }

  template <typename H>
  int partoutput_t(int itime, const char *path, int64_t *nrec) {
    const long long n = numpart;
    const int reclen = 8 + (10 + cfg.nspec) * (int)sizeof(H);
    const size_t recwords = (size_t)reclen / 4 + 2;
    unsigned int *flags = nullptr, *idx = nullptr, *out = nullptr;
    void *tmp = nullptr;
    FILE *fh = fopen(path, "wb");
    if (!fh) return fail(FPX_ERR_ARG, std::string("partoutput: cannot open ") + path);
    auto cleanup = [&]() {
      if (flags) (void)hipFree(flags);
      if (idx) (void)hipFree(idx);
      if (out) (void)hipFree(out);
      if (tmp) (void)hipFree(tmp);
      if (fh) fclose(fh);
    };
    unsigned int count = 0;
    bool io_ok = true;
    {
      const int32_t hdr[3] = {4, itime, 4};              // write(unitpartout) itime, partoutput.f90:90
      io_ok = fwrite(hdr, 4, 3, fh) == 3;
    }
    if (n > 0) {
      hipError_t e;
      if ((e = hipMalloc(&flags, n * sizeof(unsigned int))) != hipSuccess || (e = hipMalloc(&idx, n * sizeof(unsigned int))) != hipSuccess) {
        cleanup();
        return fail(FPX_ERR_NOMEM, std::string("partoutput: ") + hipGetErrorString(e));
      }
      const int nb = (int)((n + kBlock - 1) / kBlock);
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, stream);
      k_po_flags<R><<<nb, kBlock, 0, stream>>>(P, n, itime, flags);
      size_t tb = 0;
      (void)rocprim::exclusive_scan(nullptr, tb, flags, idx, 0u, (size_t)n, rocprim::plus<unsigned int>(), stream);
      if ((e = hipMalloc(&tmp, std::max<size_t>(tb, 16))) != hipSuccess) { cleanup(); return fail(FPX_ERR_NOMEM, "partoutput: scan storage"); }
      e = rocprim::exclusive_scan(tmp, tb, flags, idx, 0u, (size_t)n, rocprim::plus<unsigned int>(), stream);
      unsigned int last_idx = 0, last_flag = 0;
      if (e == hipSuccess) e = hipMemcpyAsync(&last_idx, idx + (n - 1), 4, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipMemcpyAsync(&last_flag, flags + (n - 1), 4, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) { cleanup(); return fail(FPX_ERR_DEVICE, std::string("partoutput: ") + hipGetErrorString(e)); }
      count = last_idx + last_flag;
      struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{e0, e1};
      if (count > 0) {
        const size_t words = (size_t)count * recwords;
        if ((e = hipMalloc(&out, words * 4)) != hipSuccess) { cleanup(); return fail(FPX_ERR_NOMEM, std::string("partoutput: record buffer: ") + hipGetErrorString(e)); }
        DiagP<H> D;
        D.oro = (const H *)diag_oro;
        D.tropo[0] = (const H *)diag_tropo[0]; D.tropo[1] = (const H *)diag_tropo[1];
        D.d3 = (const H *)diag_d3;
        D.nxmax = cfg.nxmax; D.nymax = cfg.nymax;
        D.dx = (H)cfg.dx; D.dy = (H)cfg.dy; D.xlon0 = (H)cfg.xlon0; D.ylat0 = (H)cfg.ylat0;
        k_partoutput<R, H><<<nb, kBlock, 0, stream>>>(V, P, D, idx, n, itime, out);
        e = hipGetLastError();
        (void)hipEventRecord(e1, stream);
        if (e == hipSuccess && hipEventSynchronize(e1) == hipSuccess) {
          float ms = 0;
          if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) po_last_ms = ms;
        }
        // stream the record bytes to the file through two pinned bounce buffers: the copy of chunk k+1 runs while
        // chunk k is written (the call is bound by the host's write, about 4 GB/s into tmpfs)
        const size_t chunk = (size_t)64 << 20, total = words * 4;
        void *pin = nullptr;
        hipEvent_t done[2] = {nullptr, nullptr};
        if (e == hipSuccess) e = hipHostMalloc(&pin, 2 * std::min(chunk, total));
        if (e == hipSuccess) e = hipEventCreate(&done[0]);
        if (e == hipSuccess) e = hipEventCreate(&done[1]);
        const size_t bufsz = std::min(chunk, total);
        auto issue = [&](size_t k) -> hipError_t {
          const size_t off = k * chunk, nbytes = std::min(chunk, total - off);
          hipError_t r = hipMemcpyAsync((char *)pin + (k & 1) * bufsz, (const char *)out + off, nbytes, hipMemcpyDeviceToHost, stream);
          return r == hipSuccess ? hipEventRecord(done[k & 1], stream) : r;
        };
        const size_t nchunks = (total + chunk - 1) / chunk;
        if (e == hipSuccess) e = issue(0);
        for (size_t k = 0; e == hipSuccess && io_ok && k < nchunks; k++) {
          e = hipEventSynchronize(done[k & 1]);
          if (e == hipSuccess && k + 1 < nchunks) e = issue(k + 1);
          const size_t nbytes = std::min(chunk, total - k * chunk);
          if (e == hipSuccess) io_ok = fwrite((const char *)pin + (k & 1) * bufsz, 1, nbytes, fh) == nbytes;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream); else (void)hipStreamSynchronize(stream);
        for (int i = 0; i < 2; i++) if (done[i]) (void)hipEventDestroy(done[i]);
        if (pin) (void)hipHostFree(pin);
        if (e != hipSuccess) { cleanup(); return fail(FPX_ERR_DEVICE, std::string("partoutput: ") + hipGetErrorString(e)); }
      }
    }
    {   // the closing record, partoutput.f90:182-184
      std::vector<unsigned char> rec(reclen + 8);
      unsigned char *p = rec.data();
      const int32_t rl = reclen, m5 = -99999;
      const H m4 = (H)-9999.9;
      memcpy(p, &rl, 4); p += 4;
      memcpy(p, &m5, 4); p += 4;
      for (int j = 0; j < 3; j++) { memcpy(p, &m4, sizeof(H)); p += sizeof(H); }
      memcpy(p, &m5, 4); p += 4;
      for (int j = 0; j < 7 + cfg.nspec; j++) { memcpy(p, &m4, sizeof(H)); p += sizeof(H); }
      memcpy(p, &rl, 4);
      io_ok = io_ok && fwrite(rec.data(), 1, rec.size(), fh) == rec.size();
    }
    io_ok = (fclose(fh) == 0) && io_ok;
    fh = nullptr;
    cleanup();
    if (!io_ok) return fail(FPX_ERR_ARG, std::string("partoutput: write error on ") + path);
    if (nrec) *nrec = count;
    return 0;
  }

  int partoutput(int itime, const char *path, int64_t *nrec) override {
    if (!path) return fail(FPX_ERR_ARG, "partoutput: null path");
    if (!height_set || !window_set || !slot_loaded[0] || !slot_loaded[1]) return fail(FPX_ERR_STATE, "partoutput: height, both field slots and the wind-time window must be set first");
    for (int i = 0; i < 9; i++)
      if (!diag_have[i]) return fail(FPX_ERR_STATE, "partoutput: oro, pv, qv, tt (fpx_upload_diag_fields or fpx_verttransform_ecmwf) and tropopause of both slots are needed");
    return cfg.host_real_bytes == 4 ? partoutput_t<float>(itime, path, nrec) : partoutput_t<double>(itime, path, nrec);
  }


  // ---- readpartpositions (warm start from the dump) -----------------------
  // random_mod.f90:12-42 (ran1), for nclass when nclassunc > 1: one serial stream, seed -8
  struct Ran1 {
    int iv[32] = {}, iy = 0, idum = -8;
    template <typename H>
    H next() {
      const int ia = 16807, im = 2147483647, iq = 127773, ir = 2836, ntab = 32, ndiv = 1 + (im - 1) / ntab;
      const H am = (H)1. / (H)im, rnmx = (H)1. - (H)1.2e-7;
      if (idum <= 0 || iy == 0) {
        idum = std::max(-idum, 1);
        for (int j = ntab + 8; j >= 1; j--) {
          const int k = idum / iq;
          idum = ia * (idum - k * iq) - ir * k;
          if (idum < 0) idum += im;
          if (j <= ntab) iv[j - 1] = idum;
        }
        iy = iv[0];
      }
      const int k = idum / iq;
      idum = ia * (idum - k * iq) - ir * k;
      if (idum < 0) idum += im;
      const int j = 1 + iy / ndiv;
      iy = iv[j - 1];
      iv[j - 1] = idum;
      return std::min(am * (H)iy, rnmx);
    }
  };

  template <typename H>
  int readpart_t(const char *path, const fpx_restart *r, int64_t *numpart_out, int32_t *numparticlecount, int32_t *itimein_out) {
    FILE *fh = fopen(path, "rb");
    if (!fh) return fail(FPX_ERR_ARG, std::string("readpartpositions: cannot open ") + path);
    struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } closer{fh};
    if (fseek(fh, 0, SEEK_END) != 0) return fail(FPX_ERR_ARG, "readpartpositions: seek");
    const long long size = ftell(fh);
    rewind(fh);
    const int reclen = 8 + (10 + cfg.nspec) * (int)sizeof(H);
    const long long recbytes = reclen + 8;
    int32_t hdr[3];
    if (size < 12 + recbytes || fread(hdr, 4, 3, fh) != 3 || hdr[0] != 4 || hdr[2] != 4 || (size - 12) % recbytes != 0)
      return fail(FPX_ERR_ARG, "readpartpositions: not a partposit dump of this build (record length = 8 + (10+nspec) reals)");
    const int itimein = hdr[1];
    const long long n = (size - 12) / recbytes - 1;
    {   // the last record must be the closing one (xlonin = -9999.9, readpartpositions.f90:120)
      std::vector<unsigned char> last((size_t)recbytes);
      H xl;
      if (fseek(fh, (long)(size - recbytes), SEEK_SET) != 0 || fread(last.data(), 1, (size_t)recbytes, fh) != (size_t)recbytes)
        return fail(FPX_ERR_ARG, "readpartpositions: short read");
      memcpy(&xl, last.data() + 8, sizeof(H));
      if (!(xl == (H)-9999.9)) return fail(FPX_ERR_ARG, "readpartpositions: the file does not end with the closing record");
      if (fseek(fh, 12, SEEK_SET) != 0) return fail(FPX_ERR_ARG, "readpartpositions: seek");
    }
    if (n > P.cap) return fail(FPX_ERR_ARG, "readpartpositions: more particles in the dump than the engine holds (maxpart)");
    // readpartpositions.f90:133-134: the previous run must end where this one starts
    if (std::abs(r->jul_header + (double)itimein / 86400. - r->bdate) > 1.e-5)
      return fail(FPX_ERR_ARG, "readpartpositions: ending time of the previous run does not agree with the start of this run");
    int *d_status = nullptr, *d_nclass = nullptr;
    unsigned int *raw = nullptr;
    void *pin = nullptr;
    auto cleanup = [&]() {
      if (d_status) (void)hipFree(d_status);
      if (d_nclass) (void)hipFree(d_nclass);
      if (raw) (void)hipFree(raw);
      if (pin) (void)hipHostFree(pin);
    };
    hipError_t e = hipMalloc(&d_status, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(d_status, 0, 2 * sizeof(int), stream);
    int status[2] = {0, 0};
    if (n > 0 && e == hipSuccess) {
      const size_t bytes = (size_t)(n + 1) * recbytes;
      e = hipMalloc(&raw, bytes);
      const size_t chunk = (size_t)64 << 20;
      if (e == hipSuccess) e = hipHostMalloc(&pin, std::min(chunk, bytes));
      for (size_t off = 0; e == hipSuccess && off < bytes; off += chunk) {
        const size_t nb = std::min(chunk, bytes - off);
        if (fread(pin, 1, nb, fh) != nb) { cleanup(); return fail(FPX_ERR_ARG, "readpartpositions: short read"); }
        e = hipMemcpyAsync((char *)raw + off, pin, nb, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
      }
      if (e == hipSuccess && r->nclassunc > 1) {      // nclass(i)=min(int(ran1(idummy)*real(nclassunc))+1,nclassunc), :142-143
        std::vector<int> nc((size_t)n);
        Ran1 g;
        for (long long i = 0; i < n; i++) nc[i] = std::min((int)(g.template next<H>() * (H)r->nclassunc) + 1, r->nclassunc);
        e = hipMalloc(&d_nclass, (size_t)n * sizeof(int));
        if (e == hipSuccess) e = hipMemcpyAsync(d_nclass, nc.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
      }
      if (e == hipSuccess) {
        k_readpart<R, H><<<(int)((n + kBlock - 1) / kBlock), kBlock, 0, stream>>>(P, raw, n, cfg.nspec, (H)cfg.dx, (H)cfg.dy, (H)cfg.xlon0, (H)cfg.ylat0,
                                                                                r->jul_header, r->bdate, r->mintime, r->itrasplit, d_nclass, d_status);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(status, d_status, sizeof status, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    cleanup();
    if (e != hipSuccess) return fail(FPX_ERR_DEVICE, std::string("readpartpositions: ") + hipGetErrorString(e));
    if (status[0] != 0) return fail(FPX_ERR_ARG, status[0] == 1 ? "readpartpositions: record markers do not match this build's record length"
                                                  : "readpartpositions: the file does not hold exactly one dump (header, records, closing record)");
    numpart = n;
    slot_of_pid = nullptr;
    maybe_new = true; births_unknown = true;
    if (numpart_out) *numpart_out = n;
    if (numparticlecount) *numparticlecount = status[1];
    if (itimein_out) *itimein_out = itimein;
    return 0;
  }
  int readpartpositions(const char *path, const fpx_restart *r, int64_t *numpart_out, int32_t *numparticlecount, int32_t *itimein) override {
    if (!path || !r) return fail(FPX_ERR_ARG, "readpartpositions: null argument");
    if (r->nclassunc < 1) return fail(FPX_ERR_ARG, "readpartpositions: nclassunc >= 1");
    return cfg.host_real_bytes == 4 ? readpart_t<float>(path, r, numpart_out, numparticlecount, itimein)
                                    : readpart_t<double>(path, r, numpart_out, numparticlecount, itimein);
  }


  // ---- releaseparticles + splitting ------------------------------------------------
  struct ReleaseTables {
    bool set = false;
    int numpoint = 0, itsplit = 0, ind_rel = 0, nclassunc = 1;
    double bdate = 0;
    std::vector<int> start, end;
    std::vector<short> kindz;
    std::vector<double> xp1, xp2, yp1, yp2, zp1, zp2;          // values of the host's real kind, widened
    std::vector<double> point_hour, area_hour, point_dow, area_dow;   // [24|7][nspec]
  } rel;
  std::vector<double> rel_xmass_h;       // [nspec][numpoint], host copy of the table of fpx_set_release_points
  std::vector<int> rel_npart_h;
  Ran1 rel_ran1 = [] { Ran1 r; r.idum = -7; return r; }();   // releaseparticles.f90:59: integer :: idummy = -7 (SAVE'd): the stream persists between calls

  int release_init(const fpx_release *r) override {
    if (!r || r->struct_bytes != (int32_t)sizeof(fpx_release)) return fail(FPX_ERR_ARG, "release_init: null or fpx_release size mismatch (ABI)");
    if (r->numpoint < 1 || r->numpoint != V.numpoint) return fail(FPX_ERR_ARG, "release_init: numpoint must equal that of fpx_set_release_points (call it first)");
    if (!r->ireleasestart || !r->ireleaseend || !r->kindz || !r->xpoint1 || !r->xpoint2 || !r->ypoint1 || !r->ypoint2 || !r->zpoint1 || !r->zpoint2)
      return fail(FPX_ERR_ARG, "release_init: the release-point arrays are required");
    if (r->nclassunc < 1) return fail(FPX_ERR_ARG, "release_init: nclassunc >= 1");
    const int np = r->numpoint, ms = cfg.maxspec, ns = cfg.nspec;
    auto rd = [&](const void *p, size_t i) -> double { return cfg.host_real_bytes == 4 ? (double)((const float *)p)[i] : ((const double *)p)[i]; };
    rel.numpoint = np; rel.itsplit = r->itsplit; rel.ind_rel = r->ind_rel; rel.nclassunc = r->nclassunc; rel.bdate = r->bdate;
    rel.start.assign(r->ireleasestart, r->ireleasestart + np); rel.end.assign(r->ireleaseend, r->ireleaseend + np);
    rel.kindz.assign(r->kindz, r->kindz + np);
    std::vector<double> *dst[6] = {&rel.xp1, &rel.xp2, &rel.yp1, &rel.yp2, &rel.zp1, &rel.zp2};
    const void *src[6] = {r->xpoint1, r->xpoint2, r->ypoint1, r->ypoint2, r->zpoint1, r->zpoint2};
    for (int a = 0; a < 6; a++) { dst[a]->resize(np); for (int i = 0; i < np; i++) (*dst[a])[i] = rd(src[a], i); }
    // (maxspec, 24|7) column-major -> [hour|day][species]; absent tables mean "no variation" (factor 1)
    auto table = [&](const void *p, int n2, std::vector<double> &out) {
      out.assign((size_t)n2 * ns, 1.0);
      if (p) for (int k = 0; k < n2; k++) for (int sp = 0; sp < ns; sp++) out[(size_t)k * ns + sp] = rd(p, (size_t)k * ms + sp);
    };
    table(r->point_hour, 24, rel.point_hour); table(r->area_hour, 24, rel.area_hour);
    table(r->point_dow, 7, rel.point_dow); table(r->area_dow, 7, rel.area_dow);
    rel.set = true;
    if (cfg.wetbkdep) {   // the height range of every point, which WETBKDEP needs in the particle loop (timemanager.f90:590-591)
      int rc = set_release_heights(np, r->zpoint1, r->zpoint2);
      if (rc) return rc;
    }
    return 0;
  }

  // juldate.f90 / caldate.f90 (the date part) in the host's real kind H
  template <typename H>
  static double jul_of(int yyyymmdd, int hhmiss) {
    const int igreg = 15 + 31 * (10 + 12 * 1582);
    int yyyy = yyyymmdd / 10000, mm = (yyyymmdd - 10000 * yyyy) / 100, dd = yyyymmdd - 10000 * yyyy - 100 * mm;
    const int hh = hhmiss / 10000, mi = (hhmiss - 10000 * hh) / 100, ss = hhmiss - 10000 * hh - 100 * mi;
    int jy, jm;
    if (yyyy < 0) yyyy = yyyy + 1;
    if (mm > 2) { jy = yyyy; jm = mm + 1; } else { jy = yyyy - 1; jm = mm + 13; }
    int julday = (int)((H)365.25 * (H)jy) + (int)((H)30.6001 * (H)jm) + dd + 1720995;
    if (dd + 31 * (mm + 12 * yyyy) >= igreg) {
      const int ja = (int)((H)0.01 * (H)jy);
      julday = julday + 2 - ja + (int)((H)0.25 * (H)ja);
    }
    return (double)julday + (double)hh / 24. + (double)mi / 1440. + (double)ss / 86400.;
  }
  template <typename H>
  static int month_of(double juldate) {
    const int igreg = 2299161;
    int julday = (int)juldate, ja;
    if ((juldate - julday) * 86400. >= 86399.5) { juldate = juldate + juldate - julday - 86399.5 / 86400.; julday = (int)juldate; }
    if (julday >= igreg) {
      const int jalpha = (int)((((H)(julday - 1867216)) - (H)0.25) / (H)36524.25);
      ja = julday + 1 + jalpha - (int)((H)0.25 * (H)jalpha);
    } else ja = julday;
    const int jb = ja + 1524;
    const int jc = (int)((H)6680. + (((H)(jb - 2439870)) - (H)122.1) / (H)365.25);
    const int jd = 365 * jc + (int)((H)0.25 * (H)jc);
    const int je = (int)((H)(jb - jd) / (H)30.6001);
    int mm = je - 1;
    if (mm > 12) mm = mm - 12;
    return mm;
  }

  template <typename H>
  int release_t(int itime, int64_t *numpart_io, int32_t *npc_io, H *xmasssave, H *rho_rel, int64_t *nreleased) {
    const int np = rel.numpoint, ns = cfg.nspec;
    // ---- releaseparticles.f90:63-131 on the host, in the host's real kind -------------------------------
    const double julmonday = jul_of<H>(19000101, 0);
    double jul = rel.bdate + (double)itime / 86400.;
    { const int mm = month_of<H>(jul); if (mm >= 4 && mm <= 9) jul = jul + 1. / 24.; }
    std::vector<long long> first(np + 1, 0), gfirst(np, 0);
    long long gcount = rel_global_count;
    std::vector<H> mass((size_t)ns * np, (H)0), xp1(np), xaux(np), yp1(np), yaux(np), zp1(np), zaux(np);
    // the fractional carry of every release point advances in a copy: the host's xmasssave is written only when the call
    // succeeds ("nothing is released" on error -- a corrected retry must release what the reference would)
    std::vector<H> xsave(xmasssave, xmasssave + np);
    auto commit_carry = [&]() { for (int i = 0; i < np; i++) xmasssave[i] = xsave[i]; };
    bool any_p3 = false;
    for (int i = 0; i < np; i++) {
      long long numrel = 0;
      xp1[i] = (H)rel.xp1[i]; yp1[i] = (H)rel.yp1[i]; zp1[i] = (H)rel.zp1[i];
      xaux[i] = (H)rel.xp2[i] - (H)rel.xp1[i]; yaux[i] = (H)rel.yp2[i] - (H)rel.yp1[i]; zaux[i] = (H)rel.zp2[i] - (H)rel.zp1[i];
      if (itime >= rel.start[i] && itime <= rel.end[i]) {
        H xlonav = (H)cfg.xlon0 + ((H)rel.xp2[i] + (H)rel.xp1[i]) / (H)2. * (H)cfg.dx;
        if (xlonav < (H)-180.) xlonav = xlonav + (H)360.;
        if (xlonav > (H)180.) xlonav = xlonav - (H)360.;
        const double jullocal = jul + (double)xlonav / 360.;
        double juldiff = jullocal - julmonday;
        const int nweeks = (int)(juldiff / 7.);
        juldiff = juldiff - (double)nweeks * 7.;
        int ndayofweek = (int)juldiff + 1;
        int nhour = (int)std::lround((juldiff - (double)(ndayofweek - 1)) * 24.);
        if (nhour == 0) { nhour = 24; ndayofweek = ndayofweek - 1; if (ndayofweek == 0) ndayofweek = 7; }
        const bool point_source = std::fabs((double)((H)rel.xp2[i] - (H)rel.xp1[i])) < 1.e-4 && std::fabs((double)((H)rel.yp2[i] - (H)rel.yp1[i])) < 1.e-4;
        std::vector<H> tc(ns);
        H avg = (H)0;
        for (int k = 0; k < ns; k++) {
          tc[k] = point_source ? (H)rel.point_hour[(size_t)(nhour - 1) * ns + k] * (H)rel.point_dow[(size_t)(ndayofweek - 1) * ns + k]
                               : (H)rel.area_hour[(size_t)(nhour - 1) * ns + k] * (H)rel.area_dow[(size_t)(ndayofweek - 1) * ns + k];
          avg = avg + tc[k];
        }
        avg = avg / (H)ns;
        if (rel.start[i] != rel.end[i]) {
          H rfraction = (H)std::fabs((double)((H)rel_npart_h[i] * (H)cfg.lsynctime / (H)(rel.end[i] - rel.start[i])));
          if (itime == rel.start[i] || itime == rel.end[i]) rfraction = rfraction / (H)2.;
          rfraction = rfraction * avg;
          rfraction = rfraction + xsave[i];
          numrel = (long long)(int)rfraction;
          xsave[i] = rfraction - (H)(int)numrel;
        } else numrel = rel_npart_h[i];
        for (int k = 0; k < ns; k++) mass[(size_t)k * np + i] = (H)rel_xmass_h[(size_t)k * np + i] / (H)rel_npart_h[i] * tc[k] / avg;
        if (numrel > 0 && rel.kindz[i] == 3) any_p3 = true;
      }
      // several ranks (releaseparticles_mpi.f90:139-162): every rank takes numrel / nranks particles of the point, the first
      // mod(numrel, nranks) ranks one more; the particles of all ranks together are those of the single-rank run (the counter
      // RNG is keyed on the particle's number in the release count of the whole run)
      long long numrel_all = std::max<long long>(numrel, 0), roff = 0;
      numrel = numrel_all;
      if (comm_ranks > 1) {
        const long long base = numrel_all / comm_ranks, rem = numrel_all % comm_ranks;
        numrel = base + (comm_rank < rem ? 1 : 0);
        roff = (long long)comm_rank * base + std::min<long long>(comm_rank, rem);
      }
      gfirst[i] = gcount + roff;
      gcount += numrel_all;
      first[i + 1] = first[i] + numrel;
    }
    const long long ntotal = first[np];
    if (nreleased) *nreleased = ntotal;
    if (ntotal == 0) { rel_global_count = gcount; commit_carry(); return 0; }
    const bool dens = rel.ind_rel == 1 || rel.ind_rel == 3 || rel.ind_rel == 4;
    for (int l = 0; l < V.numbnests; l++) {
      if (!rel_nest_oro[l]) return fail(FPX_ERR_STATE, "releaseparticles: oron of every nest is needed (fpx_upload_diag_nest_fields slot 0)");
      if ((any_p3 || dens) && !nest_loaded[l][1]) return fail(FPX_ERR_STATE, "releaseparticles: rhon of time slot 2 is needed (fpx_upload_nest_fields)");
      if (any_p3 && !rel_nest_tt2[l]) return fail(FPX_ERR_STATE, "releaseparticles: ttn of time slot 2 is needed for kindz = 3 (fpx_upload_diag_nest_fields slot 2)");
    }
    if (!height_set || !diag_have[DG_ORO]) return fail(FPX_ERR_STATE, "releaseparticles: height and oro (fpx_upload_diag_fields slot 0) are needed");
    if ((any_p3 || dens) && !slot_loaded[1]) return fail(FPX_ERR_STATE, "releaseparticles: rho of time slot 2 is needed (kindz = 3 or ind_rel = 1, 3, 4)");
    if (any_p3 && !diag_have[DG_TT + 1]) return fail(FPX_ERR_STATE, "releaseparticles: tt of time slot 2 is needed for kindz = 3 (fpx_upload_diag_fields slot 2)");
    numpart = *numpart_io;
    // ---- the k-th particle takes the k-th vacant storage space in particle-number order (:133-137) --------
    const long long cap = P.cap;
    std::vector<void *> mine;
    auto cleanup = [&]() { for (void *q : mine) (void)hipFree(q); };
    auto mal = [&](auto **q, size_t n) -> hipError_t { hipError_t e = hipMalloc((void **)q, std::max<size_t>(n, 1) * sizeof(**q)); if (e == hipSuccess) mine.push_back(*q); return e; };
    unsigned int *flags = nullptr, *rank = nullptr, *target = nullptr, *d_max = nullptr;
    long long *d_first = nullptr, *d_gfirst = nullptr;
    H *d_pts = nullptr, *d_mass = nullptr, *d_uni = nullptr, *d_rho = nullptr;
    short *d_kindz = nullptr;
    void *tmp = nullptr;
    // the capacity-sized work arrays are kept between calls (grow-only): hipFree is a device-wide synchronisation, and at
    // the bench size they are 0.8 GB per call
    hipError_t e = rel_scratch((size_t)cap, &flags, &rank) ? hipErrorOutOfMemory : hipSuccess;
    if (e == hipSuccess) e = mal(&target, (size_t)ntotal);
    if (e == hipSuccess) e = mal(&d_max, 1);
    if (e == hipSuccess) e = mal(&d_first, (size_t)np + 1);
    if (e == hipSuccess) e = mal(&d_gfirst, (size_t)np);
    if (e == hipSuccess) e = mal(&d_pts, (size_t)6 * np);
    if (e == hipSuccess) e = mal(&d_mass, (size_t)ns * np);
    if (e == hipSuccess) e = mal(&d_kindz, (size_t)np);
    if (e == hipSuccess) e = mal(&d_rho, (size_t)np);
    // the vacancies are looked for in the storage spaces 1 .. numpart + ntotal only: the spaces behind numpart are vacant, so
    // the first ntotal vacancies lie in there (or, past the capacity, do not exist) -- not a scan of the whole capacity per call
    const long long scan_n = std::min<long long>(cap, numpart + ntotal);
    size_t tb = 0;
    if (e == hipSuccess) {
      (void)rocprim::exclusive_scan(nullptr, tb, flags, rank, 0u, (size_t)scan_n, rocprim::plus<unsigned int>(), stream);
      e = rel_scan_tmp(tb, &tmp) ? hipErrorOutOfMemory : hipSuccess;
    }
    if (e != hipSuccess) { cleanup(); return fail(FPX_ERR_NOMEM, std::string("releaseparticles: ") + hipGetErrorString(e)); }
    const int nbc = (int)((scan_n + kBlock - 1) / kBlock);
    k_rel_flags<R><<<nbc, kBlock, 0, stream>>>(P, slot_map(), scan_n, itime, flags);
    e = rocprim::exclusive_scan(tmp, tb, flags, rank, 0u, (size_t)scan_n, rocprim::plus<unsigned int>(), stream);
    unsigned int last[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(&last[0], rank + (scan_n - 1), 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&last[1], flags + (scan_n - 1), 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { cleanup(); return fail(FPX_ERR_DEVICE, std::string("releaseparticles: ") + hipGetErrorString(e)); }
    if ((long long)last[0] + last[1] < ntotal) {
      cleanup();
      return fail(FPX_ERR_NOMEM, "releaseparticles: total number of particles required exceeds the maximum allowed number (releaseparticles.f90:369-378)");
    }
    k_rel_targets<<<nbc, kBlock, 0, stream>>>(flags, rank, scan_n, ntotal, target);
    // ---- the four uniforms of every particle: serial ran1 stream (x, y, class, z per particle) or counter RNG ----
    std::vector<H> uni;
    if (cfg.rng_mode == FPX_RNG_TABLE_SEQ) {
      uni.resize((size_t)4 * ntotal);
      for (long long k = 0; k < ntotal; k++)
        for (int d = 0; d < 4; d++) uni[(size_t)d * ntotal + k] = rel_ran1.template next<H>();
      e = mal(&d_uni, (size_t)4 * ntotal);
      if (e == hipSuccess) e = hipMemcpyAsync(d_uni, uni.data(), uni.size() * sizeof(H), hipMemcpyHostToDevice, stream);
    }
    std::vector<H> pts((size_t)6 * np);
    for (int i = 0; i < np; i++) { pts[i] = xp1[i]; pts[np + i] = xaux[i]; pts[2 * np + i] = yp1[i]; pts[3 * np + i] = yaux[i]; pts[4 * np + i] = zp1[i]; pts[5 * np + i] = zaux[i]; }
    std::vector<H> rho_h(np, (H)0);
    if (rho_rel) for (int i = 0; i < np; i++) rho_h[i] = rho_rel[i];
    if (e == hipSuccess) e = hipMemcpyAsync(d_first, first.data(), (np + 1) * sizeof(long long), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_gfirst, gfirst.data(), np * sizeof(long long), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pts, pts.data(), pts.size() * sizeof(H), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_mass, mass.data(), mass.size() * sizeof(H), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_kindz, rel.kindz.data(), np * sizeof(short), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rho, rho_h.data(), np * sizeof(H), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_max, 0, 4, stream);
    if (e != hipSuccess) { (void)hipStreamSynchronize(stream); cleanup(); return fail(FPX_ERR_DEVICE, std::string("releaseparticles: ") + hipGetErrorString(e)); }
    RelPoints<H> RP{d_first, d_gfirst, d_pts, d_pts + np, d_pts + 2 * np, d_pts + 3 * np, d_pts + 4 * np, d_pts + 5 * np, d_mass, d_kindz, np};
    DiagP<H> D;
    D.oro = (const H *)diag_oro; D.tropo[0] = (const H *)diag_tropo[0]; D.tropo[1] = (const H *)diag_tropo[1]; D.d3 = (const H *)diag_d3;
    D.nxmax = cfg.nxmax; D.nymax = cfg.nymax; D.dx = (H)cfg.dx; D.dy = (H)cfg.dy; D.xlon0 = (H)cfg.xlon0; D.ylat0 = (H)cfg.ylat0;
    RelNests<R, H> NS;
    NS.n = V.numbnests;
    NS.eps = (H)cfg.par_nxmax / (H)3.e5;
    for (int l = 0; l < V.numbnests; l++) {
      const NestDesc<R> &N = h_nest[l];
      NS.g[l].nx = N.nx; NS.g[l].ny = N.ny;
      NS.g[l].xl = (H)N.xl; NS.g[l].yl = (H)N.yl; NS.g[l].xr = (H)N.xr; NS.g[l].yr = (H)N.yr; NS.g[l].xres = (H)N.xres; NS.g[l].yres = (H)N.yres;
      NS.g[l].oro = (const H *)rel_nest_oro[l]; NS.g[l].tt2 = (const H *)rel_nest_tt2[l]; NS.g[l].r2 = N.r2;
    }
    k_release<R, H><<<(int)((ntotal + kBlock - 1) / kBlock), kBlock, 0, stream>>>(V, P, D, NS, RP, target, slot_map(), ntotal, itime, d_uni, (int)*npc_io,
                                                                                  cfg.mintime, rel.itsplit, rel.ind_rel, rel.nclassunc, cfg.mquasilag,
                                                                                  dens ? d_rho : (H *)nullptr, d_max);
    e = hipGetLastError();
    unsigned int maxpid = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&maxpid, d_max, 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && rho_rel && dens) e = hipMemcpyAsync(rho_h.data(), d_rho, np * sizeof(H), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream); else (void)hipStreamSynchronize(stream);
    cleanup();
    if (e != hipSuccess) return fail(FPX_ERR_DEVICE, std::string("releaseparticles: ") + hipGetErrorString(e));
    if (rho_rel && dens) for (int i = 0; i < np; i++) rho_rel[i] = rho_h[i];
    numpart = std::max<long long>(numpart, (long long)maxpid + 1);          // :362
    *numpart_io = numpart;
    *npc_io = (int32_t)(*npc_io + ntotal);
    rel_global_count = gcount;
    commit_carry();
    maybe_new = true;
    return 0;
  }
  // grow-only scratch of fpx_releaseparticles / fpx_split_particles: two unsigned arrays of n elements and the scan's temporary
  unsigned int *rel_flags_buf = nullptr, *rel_rank_buf = nullptr;
  size_t rel_buf_n = 0;
  void *rel_tmp_buf = nullptr;
  size_t rel_tmp_bytes = 0;
  int rel_scratch(size_t n, unsigned int **flags, unsigned int **rank) {
    if (n > rel_buf_n) {
      (void)hipStreamSynchronize(stream);
      if (rel_flags_buf) (void)hipFree(rel_flags_buf);
      if (rel_rank_buf) (void)hipFree(rel_rank_buf);
      rel_flags_buf = rel_rank_buf = nullptr; rel_buf_n = 0;
      if (hipMalloc(&rel_flags_buf, n * 4) != hipSuccess) return 1;
      if (hipMalloc(&rel_rank_buf, n * 4) != hipSuccess) { (void)hipFree(rel_flags_buf); rel_flags_buf = nullptr; return 1; }
      rel_buf_n = n;
    }
    *flags = rel_flags_buf; *rank = rel_rank_buf;
    return 0;
  }
  int rel_scan_tmp(size_t bytes, void **tmp) {
    bytes = std::max<size_t>(bytes, 16);
    if (bytes > rel_tmp_bytes) {
      (void)hipStreamSynchronize(stream);
      if (rel_tmp_buf) (void)hipFree(rel_tmp_buf);
      rel_tmp_buf = nullptr; rel_tmp_bytes = 0;
      if (hipMalloc(&rel_tmp_buf, bytes) != hipSuccess) return 1;
      rel_tmp_bytes = bytes;
    }
    *tmp = rel_tmp_buf;
    return 0;
  }
  int releaseparticles(int itime, int64_t *numpart_io, int32_t *npc_io, void *xmasssave, void *rho_rel, int64_t *nreleased) override {
    if (!rel.set) return fail(FPX_ERR_STATE, "releaseparticles: fpx_release_init first");
    if (!numpart_io || !npc_io || !xmasssave) return fail(FPX_ERR_ARG, "releaseparticles: numpart, numparticlecount and xmasssave are required");
    if (*numpart_io < 0 || *numpart_io > P.cap) return fail(FPX_ERR_ARG, "releaseparticles: numpart outside capacity");
    return cfg.host_real_bytes == 4 ? release_t<float>(itime, numpart_io, npc_io, (float *)xmasssave, (float *)rho_rel, nreleased)
                                    : release_t<double>(itime, numpart_io, npc_io, (double *)xmasssave, (double *)rho_rel, nreleased);
  }
  int split_particles(int itime, int64_t *numpart_io) override {
    if (!rel.set) return fail(FPX_ERR_STATE, "split_particles: fpx_release_init first (itsplit)");
    if (!numpart_io || *numpart_io < 0 || *numpart_io > P.cap) return fail(FPX_ERR_ARG, "split_particles: numpart outside capacity");
    numpart = *numpart_io;
    if (!(cfg.ldirect * itime >= cfg.ldirect * rel.itsplit) || numpart == 0 || numpart == P.cap) return 0;   // timemanager.f90:473
    const long long n = numpart, room = P.cap - n;
    unsigned int *flags = nullptr, *rank = nullptr;
    void *tmp = nullptr;
    auto cleanup = [&]() {};      // the work arrays are the engine's grow-only scratch (rel_scratch)
    hipError_t e = rel_scratch((size_t)n, &flags, &rank) ? hipErrorOutOfMemory : hipSuccess;
    size_t tb = 0;
    if (e == hipSuccess) {
      (void)rocprim::exclusive_scan(nullptr, tb, flags, rank, 0u, (size_t)n, rocprim::plus<unsigned int>(), stream);
      e = rel_scan_tmp(tb, &tmp) ? hipErrorOutOfMemory : hipSuccess;
    }
    if (e != hipSuccess) { cleanup(); return fail(FPX_ERR_NOMEM, std::string("split_particles: ") + hipGetErrorString(e)); }
    const int nb = (int)((n + kBlock - 1) / kBlock);
    k_split_flags<R><<<nb, kBlock, 0, stream>>>(P, slot_map(), n, itime, cfg.ldirect, flags);
    e = rocprim::exclusive_scan(tmp, tb, flags, rank, 0u, (size_t)n, rocprim::plus<unsigned int>(), stream);
    unsigned int last[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(&last[0], rank + (n - 1), 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&last[1], flags + (n - 1), 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) {
      k_split<R><<<nb, kBlock, 0, stream>>>(P, slot_map(), flags, rank, n, room, cfg.nspec);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream); else (void)hipStreamSynchronize(stream);
    cleanup();
    if (e != hipSuccess) return fail(FPX_ERR_DEVICE, std::string("split_particles: ") + hipGetErrorString(e));
    numpart = n + std::min<long long>((long long)last[0] + last[1], room);
    *numpart_io = numpart;
    return 0;
  }

  // ---- particle redistribution between ranks, mpi_mod.f90:661-856 ------------------------------
  // The host keeps the transport (its MPI_Send / MPI_Recv, one message instead of the reference's 9 + nspec); the engine
  // packs and places.  buf may be host or device memory (hipMemcpyDefault).
  void *redist_dev = nullptr;
  size_t redist_dev_bytes = 0;
  int redist_stage(size_t bytes) {
    if (bytes <= redist_dev_bytes) return 0;
    if (redist_dev) { (void)hipStreamSynchronize(stream); (void)hipFree(redist_dev); redist_dev = nullptr; redist_dev_bytes = 0; }
    if (hipMalloc(&redist_dev, bytes) != hipSuccess) return -1;
    redist_dev_bytes = bytes;
    return 0;
  }
  size_t redist_bytes(int64_t num_trans) override { return num_trans > 0 ? RedistBuf<R>::bytes(num_trans, cfg.nspec) : 0; }
  int redist_pack(int itime, int64_t num_trans, void *buf, size_t buf_bytes, int64_t *numpart_io) override {
    (void)itime;
    if (!numpart_io || *numpart_io < 0 || *numpart_io > P.cap) return fail(FPX_ERR_ARG, "redist_pack: numpart outside capacity");
    if (num_trans < 0 || num_trans > *numpart_io) return fail(FPX_ERR_ARG, "redist_pack: num_trans must be between 0 and numpart");
    if (num_trans == 0) return 0;
    const size_t need = redist_bytes(num_trans);
    if (!buf || buf_bytes < need) return fail(FPX_ERR_ARG, "redist_pack: buffer smaller than fpx_redist_bytes(num_trans)");
    if (redist_stage(need)) return fail(FPX_ERR_NOMEM, "redist_pack: staging buffer");
    // first back into particle-number order over the extent the engine itself holds (the extent of the last locality sort:
    // its slots are a permutation of exactly those numbers), then the host's count -- as set_numpart does; the other order
    // would sort a range whose numbers are not a permutation and lose live particles
    { const int rc = restore_particle_order(); if (rc) return rc; }   // numpart shrinks below
    numpart = *numpart_io;
    const RedistBuf<R> B = RedistBuf<R>::at(redist_dev, num_trans, cfg.nspec);
    k_redist_pack<R><<<(int)((num_trans + kBlock - 1) / kBlock), kBlock, 0, stream>>>(P, slot_map(), numpart - num_trans, B, cfg.nspec);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(buf, redist_dev, need, hipMemcpyDefault, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream); else (void)hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(FPX_ERR_DEVICE, std::string("redist_pack: ") + hipGetErrorString(e));
    numpart -= num_trans;                                 // :746
    *numpart_io = numpart;
    return 0;
  }
  int redist_unpack(int itime, int64_t num_trans, const void *buf, size_t buf_bytes, int64_t *numpart_io) override {
    if (!numpart_io || *numpart_io < 0 || *numpart_io > P.cap) return fail(FPX_ERR_ARG, "redist_unpack: numpart outside capacity");
    if (num_trans < 0) return fail(FPX_ERR_ARG, "redist_unpack: num_trans < 0");
    if (num_trans == 0) return 0;
    const long long maxnumpart = *numpart_io + num_trans;   // :811
    if (maxnumpart > P.cap) return fail(FPX_ERR_NOMEM, "redist_unpack: numpart + num_trans exceeds the particle capacity (the reference would write past maxpart)");
    const size_t need = redist_bytes(num_trans);
    if (!buf || buf_bytes < need) return fail(FPX_ERR_ARG, "redist_unpack: buffer smaller than fpx_redist_bytes(num_trans)");
    // work arrays: vacancy flags + ranks over 1..maxnumpart, validity flags + ranks and the targets over the message
    unsigned int *flags = nullptr, *rank = nullptr;
    void *tmp = nullptr;
    const size_t nwork = (size_t)maxnumpart + 3 * (size_t)num_trans + 1;
    if (redist_stage(need)) return fail(FPX_ERR_NOMEM, "redist_unpack: staging buffer");
    if (rel_scratch(nwork, &flags, &rank)) return fail(FPX_ERR_NOMEM, "redist_unpack: work arrays");
    unsigned int *valid = flags + maxnumpart, *vrank = rank + maxnumpart;
    unsigned int *target = flags + maxnumpart + num_trans, *d_max = rank + maxnumpart + num_trans;
    size_t tb1 = 0, tb2 = 0;
    (void)rocprim::exclusive_scan(nullptr, tb1, flags, rank, 0u, (size_t)maxnumpart, rocprim::plus<unsigned int>(), stream);
    (void)rocprim::exclusive_scan(nullptr, tb2, valid, vrank, 0u, (size_t)num_trans, rocprim::plus<unsigned int>(), stream);
    if (rel_scan_tmp(std::max(tb1, tb2), &tmp)) return fail(FPX_ERR_NOMEM, "redist_unpack: scan storage");
    numpart = *numpart_io;
    hipError_t e = hipMemcpyAsync(redist_dev, buf, need, hipMemcpyDefault, stream);
    const RedistBuf<R> B = RedistBuf<R>::at(redist_dev, num_trans, cfg.nspec);
    const int nbm = (int)((maxnumpart + kBlock - 1) / kBlock), nbt = (int)((num_trans + kBlock - 1) / kBlock);
    if (e == hipSuccess) {
      k_rel_flags<R><<<nbm, kBlock, 0, stream>>>(P, slot_map(), maxnumpart, itime, flags);          // vacant: itra1 /= itime (:818)
      k_redist_valid<<<nbt, kBlock, 0, stream>>>(B.itra1, num_trans, itime, valid);                  // :816
      e = rocprim::exclusive_scan(tmp, tb1, flags, rank, 0u, (size_t)maxnumpart, rocprim::plus<unsigned int>(), stream);
    }
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, tb2, valid, vrank, 0u, (size_t)num_trans, rocprim::plus<unsigned int>(), stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_max, 0, 4, stream);
    if (e == hipSuccess) {
      // the spaces numpart+1 .. maxnumpart are vacant, so there are at least num_trans targets
      k_rel_targets<<<nbm, kBlock, 0, stream>>>(flags, rank, maxnumpart, num_trans, target);
      k_redist_unpack<R><<<nbt, kBlock, 0, stream>>>(P, slot_map(), B, cfg.nspec, valid, vrank, target, d_max);
      e = hipGetLastError();
    }
    unsigned int maxpid = 0, nvalid[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(&maxpid, d_max, 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&nvalid[0], vrank + (num_trans - 1), 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&nvalid[1], valid + (num_trans - 1), 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream); else (void)hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(FPX_ERR_DEVICE, std::string("redist_unpack: ") + hipGetErrorString(e));
    if (nvalid[0] + nvalid[1] > 0) numpart = std::max<long long>(numpart, (long long)maxpid + 1);   // numpart=max(numpart,ipart), :836
    *numpart_io = numpart;
    maybe_new = true; births_unknown = true;
    return 0;
  }

  // ---- concoutput: the grid_conc files ------------------------------------
  int concoutput(int itime, const fpx_concout *c, const char *prefix, int clear) override {
    if (!Gp.on) return fail(FPX_ERR_STATE, "concoutput: fpx_outgrid_init first");
    if (!c || !c->area || !c->volume || !prefix) return fail(FPX_ERR_ARG, "concoutput: area, volume and the file name prefix are required");
    if (cfg.host_real_bytes != 4) return fail(FPX_ERR_ARG, "concoutput: only for hosts with a 4-byte default real (the reference's concoutput.f90 does not compile with 8)");
    if (cfg.ldirect != 1) return fail(FPX_ERR_ARG, "concoutput: forward runs only (ldirect = 1)");
    if (!(c->outnum > 0)) return fail(FPX_ERR_ARG, "concoutput: outnum > 0");
    const int iout = c->iout ? c->iout : 1;
    if (iout < 1 || iout > 3) return fail(FPX_ERR_ARG, "concoutput: iout 1 (concentration), 2 (mixing ratio) or 3 (both)");
    const bool want_conc = iout == 1 || iout == 3, want_ppt = iout == 2 || iout == 3;
    if (want_ppt && (!c->prefix_pptv || !c->outheight)) return fail(FPX_ERR_ARG, "concoutput: prefix_pptv and outheight are required for iout 2, 3");
    if (want_ppt && (!slot_loaded[0] || !slot_loaded[1] || !window_set || !height_set)) return fail(FPX_ERR_STATE, "concoutput: mixing ratios need the met fields and the wind-time window");
    if (c->nest && !Gp.nested) return fail(FPX_ERR_STATE, "concoutput: nested output grid requested without fpx_outgrid_nest_init");
    // nest = 1: the nested output grid (concoutput_nest.f90: the same algorithm on griduncn, wetgriduncn, drygriduncn, arean, volumen)
    const long long n2 = c->nest ? (long long)Gp.numxgridn * Gp.numygridn : (long long)Gp.numxgrid * Gp.numygrid, n3 = n2 * Gp.numzgrid;
    R *g3 = c->nest ? Gp.griduncn : Gp.gridunc;            // the rank's own partial sums (zeroed afterwards with clear = 1)
    float *gwet = c->nest ? Gp.wetgriduncn : Gp.wetgridunc, *gdry = c->nest ? Gp.drygriduncn : Gp.drygridunc;
    const R *s3 = g3;                                      // what is written
    if (c->reduced) {
      // the sums over all ranks that fpx_get_grids / fpx_get_wetgrid / fpx_get_grids_nest (allreduce = 1) left in the
      // receive buffers -- concoutput_mpi.f90:279,298 writes from gridunc0 / drygridunc0 / wetgridunc0
      const int ig = c->nest ? RG_GRIDN : RG_GRID, id = c->nest ? RG_DRYN : RG_DRY, iw = c->nest ? RG_WETN : RG_WET;
      if (comm_ranks > 1) {
        if (!red_valid[ig] || (c->drydep && !red_valid[id]) || (c->wetdep && !red_valid[iw]))
          return fail(FPX_ERR_STATE, "concoutput: reduced = 1 needs fpx_get_grids / fpx_get_wetgrid / fpx_get_grids_nest with allreduce = 1 first");
        s3 = c->nest ? griduncn0 : gridunc0;
        if (c->drydep) gdry = c->nest ? drygriduncn0 : drygridunc0;
        if (c->wetdep) gwet = c->nest ? wetgriduncn0 : wetgridunc0;
      }
    }
    float *d_area = nullptr, *d_vol = nullptr, *val = nullptr, *wr = nullptr;
    unsigned int *nz = nullptr, *rs = nullptr, *rpos = nullptr, *runid = nullptr;
    int *wi = nullptr;
    void *tmp = nullptr;
    std::vector<void *> mine;
    auto cleanup = [&]() { for (void *q : mine) (void)hipFree(q); if (tmp) (void)hipFree(tmp); };
    auto mal = [&](auto **q, size_t bytes) -> hipError_t { hipError_t e = hipMalloc((void **)q, bytes); if (e == hipSuccess) mine.push_back(*q); return e; };
    float *d_dens = nullptr, *d_outh = nullptr;
    hipError_t e = mal(&d_area, n2 * 4);
    if (e == hipSuccess) e = mal(&d_vol, n3 * 4);
    if (e == hipSuccess && want_ppt) e = mal(&d_dens, n3 * 4);
    if (e == hipSuccess && want_ppt) e = mal(&d_outh, (size_t)Gp.numzgrid * 4);
    if (e == hipSuccess) e = mal(&val, n3 * 4);
    if (e == hipSuccess) e = mal(&wr, n3 * 4);
    if (e == hipSuccess) e = mal(&nz, n3 * 4);
    if (e == hipSuccess) e = mal(&rs, n3 * 4);
    if (e == hipSuccess) e = mal(&rpos, n3 * 4);
    if (e == hipSuccess) e = mal(&runid, n3 * 4);
    if (e == hipSuccess) e = mal(&wi, n3 * 4);
    size_t tb = 0;
    if (e == hipSuccess) {
      (void)rocprim::exclusive_scan(nullptr, tb, nz, rpos, 0u, (size_t)n3, rocprim::plus<unsigned int>(), stream);
      size_t tb2 = 0;
      (void)rocprim::inclusive_scan(nullptr, tb2, rs, runid, (size_t)n3, rocprim::plus<unsigned int>(), stream);
      tb = std::max(tb, tb2);
      e = hipMalloc(&tmp, std::max<size_t>(tb, 16));
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_area, c->area, n2 * 4, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_vol, c->volume, n3 * 4, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && want_ppt) {
      e = hipMemcpyAsync(d_outh, c->outheight, (size_t)Gp.numzgrid * 4, hipMemcpyHostToDevice, stream);
      if (e == hipSuccess) {
        k_co_density<R><<<(int)((n3 + kBlock - 1) / kBlock), kBlock, 0, stream>>>(V, c->nest ? Gp.numxgridn : Gp.numxgrid, c->nest ? Gp.numygridn : Gp.numygrid, Gp.numzgrid, d_outh,
                                                                                (float)(c->nest ? Gp.dxoutn : Gp.dxout), (float)(c->nest ? Gp.dyoutn : Gp.dyout),
                                                                                (float)c->outlon0, (float)c->outlat0, (float)cfg.dx, (float)cfg.dy, (float)cfg.xlon0, (float)cfg.ylat0, d_dens);
        e = hipGetLastError();
      }
    }
    if (e != hipSuccess) { cleanup(); return fail(FPX_ERR_NOMEM, std::string("concoutput: ") + hipGetErrorString(e)); }
    std::vector<int> h_wi((size_t)n3);
    std::vector<float> h_wr((size_t)n3);
    // one compressed dump: device work, then the four records (count, indices, count, values)
    auto dump = [&](auto *grid, size_t class_stride, long long n, const float *scale, int conc, int idx0, FILE *fh, float wm = 1.f) -> int {
      const int nb = (int)((n + kBlock - 1) / kBlock);
      int32_t ci = 0, cr = 0;
      if (grid) {
        k_co_values<<<nb, kBlock, 0, stream>>>(grid, class_stride, Gp.nclassunc, n, val, nz, rs);
        size_t t1 = tb;
        hipError_t e2 = rocprim::exclusive_scan(tmp, t1, nz, rpos, 0u, (size_t)n, rocprim::plus<unsigned int>(), stream);
        t1 = tb;
        if (e2 == hipSuccess) e2 = rocprim::inclusive_scan(tmp, t1, rs, runid, (size_t)n, rocprim::plus<unsigned int>(), stream);
        k_co_write<<<nb, kBlock, 0, stream>>>(val, nz, rs, rpos, runid, n, scale, conc, (float)c->outnum, 1.f, idx0, wi, wr, d_dens, wm);
        unsigned int last[3] = {0, 0, 0};
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(&last[0], rpos + (n - 1), 4, hipMemcpyDeviceToHost, stream);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(&last[1], nz + (n - 1), 4, hipMemcpyDeviceToHost, stream);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(&last[2], runid + (n - 1), 4, hipMemcpyDeviceToHost, stream);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(stream);
        cr = (int32_t)(last[0] + last[1]); ci = (int32_t)last[2];
        if (e2 == hipSuccess && ci > 0) e2 = hipMemcpyAsync(h_wi.data(), wi, (size_t)ci * 4, hipMemcpyDeviceToHost, stream);
        if (e2 == hipSuccess && cr > 0) e2 = hipMemcpyAsync(h_wr.data(), wr, (size_t)cr * 4, hipMemcpyDeviceToHost, stream);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(stream);
        if (e2 != hipSuccess) return fail(FPX_ERR_DEVICE, std::string("concoutput: ") + hipGetErrorString(e2));
      }
      const int32_t four = 4, li = 4 * ci, lr = 4 * cr;
      bool ok = fwrite(&four, 4, 1, fh) == 1 && fwrite(&ci, 4, 1, fh) == 1 && fwrite(&four, 4, 1, fh) == 1;
      ok = ok && fwrite(&li, 4, 1, fh) == 1 && (ci == 0 || fwrite(h_wi.data(), 4, (size_t)ci, fh) == (size_t)ci) && fwrite(&li, 4, 1, fh) == 1;
      ok = ok && fwrite(&four, 4, 1, fh) == 1 && fwrite(&cr, 4, 1, fh) == 1 && fwrite(&four, 4, 1, fh) == 1;
      ok = ok && fwrite(&lr, 4, 1, fh) == 1 && (cr == 0 || fwrite(h_wr.data(), 4, (size_t)cr, fh) == (size_t)cr) && fwrite(&lr, 4, 1, fh) == 1;
      return ok ? 0 : fail(FPX_ERR_ARG, "concoutput: write error");
    };
    int rc = 0;
    for (int ks = 0; ks < cfg.nspec && !rc; ks++)
      for (int pass = 0; pass < 2 && !rc; pass++) {          // 0: grid_conc (:349-447), 1: grid_pptv (:482-590)
        if ((pass == 0 && !want_conc) || (pass == 1 && !want_ppt)) continue;
        char name[1024];
        snprintf(name, sizeof name, "%s%03d", pass == 0 ? prefix : c->prefix_pptv, ks + 1);
        FILE *fh = fopen(name, "wb");
        if (!fh) { rc = fail(FPX_ERR_ARG, std::string("concoutput: cannot open ") + name); break; }
        const int32_t hdr[3] = {4, itime, 4};
        if (fwrite(hdr, 4, 3, fh) != 3) rc = fail(FPX_ERR_ARG, "concoutput: write error");
        for (int kp = 0; kp < Gp.maxpointspec_act && !rc; kp++)
          for (int nage = 0; nage < Gp.nageclass && !rc; nage++) {
            // element (0,0,[1,]ks,kp,class 0,nage) of the 6-D / 7-D arrays; consecutive classes are class_stride apart
            const size_t o2 = ((((size_t)nage * Gp.nclassunc) * Gp.maxpointspec_act + kp) * Gp.maxspec + ks) * (size_t)n2;
            const size_t cs2 = (size_t)Gp.maxpointspec_act * Gp.maxspec * (size_t)n2;
            const size_t o3 = o2 * Gp.numzgrid, cs3 = cs2 * Gp.numzgrid;
            rc = dump(c->wetdep && gwet ? gwet + o2 : (float *)nullptr, cs2, n2, d_area, 0, 0, fh);
            if (!rc) rc = dump(c->drydep && gdry ? gdry + o2 : (float *)nullptr, cs2, n2, d_area, 0, 0, fh);
            if (!rc) rc = dump(s3 + o3, cs3, n3, d_vol, pass == 0 ? 1 : 2, (int)n2 /* kz is 1-based in the index, :425 */, fh,
                               pass == 1 ? (float)c->weightmolar[ks] : 1.f);
          }
        if (fclose(fh) != 0 && !rc) rc = fail(FPX_ERR_ARG, "concoutput: write error");
      }
    cleanup();
    if (rc) return rc;
    if (clear) {   // gridunc(:,:,:,:,:,:,:)=0., concoutput.f90:714 / griduncn, concoutput_nest.f90 (the deposition grids keep accumulating)
      HIPCHK(hipMemsetAsync(g3, 0, (c->nest ? n_grid3n : n_grid3) * sizeof(R), stream));
      HIPCHK(hipStreamSynchronize(stream));
    }
    return 0;
  }

  int set_windtime(const int32_t mt[2], const int32_t mi[2]) override {
    if (!mt || !mi) return fail(FPX_ERR_ARG, "set_windtime: null");
    if ((mi[0] != 1 && mi[0] != 2) || (mi[1] != 1 && mi[1] != 2) || mi[0] == mi[1]) return fail(FPX_ERR_ARG, "set_windtime: memind must be a permutation of (1,2)");
    if (mt[0] == mt[1]) return fail(FPX_ERR_ARG, "set_windtime: memtime(1) == memtime(2)");
    V.memtime0 = mt[0]; V.memtime1 = mt[1]; V.m1 = mi[0] - 1; V.m2 = mi[1] - 1;
    V.lwindinterv = std::abs(mt[1] - mt[0]);
    {   // mesoscale autocorrelation, advance.f90:728-729: the same two numbers for every particle of a step
      const R r = sizeof(R) == 4 ? (R)expf(-2.0f * (float)std::abs(cfg.lsynctime) / (float)V.lwindinterv)
                                 : (R)exp(-2.0 * (double)std::abs(cfg.lsynctime) / (double)V.lwindinterv);
      V.meso_r = r;
      V.meso_rs = sizeof(R) == 4 ? (R)sqrtf(1.0f - (float)r * (float)r) : (R)sqrt(1.0 - (double)r * (double)r);
    }
    window_set = true;
    return 0;
  }

  // ---- RNG -----------------------------------------------------------------
  int rng_fill_table() override {
    const int n = V.maxrand;
    std::vector<R> t(n);
    if (cfg.host_real_bytes == 4) { rng4 = HostRng<float>(); rng4.fill_table(tab4, n); for (int i = 0; i < n; i++) t[i] = (R)tab4[i]; }
    else { rng8 = HostRng<double>(); rng8.fill_table(tab8, n); for (int i = 0; i < n; i++) t[i] = (R)tab8[i]; }
    HIPCHK(hipMemcpyAsync((void *)V.rannumb, t.data(), n * sizeof(R), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    table_set = true;
    return 0;
  }
  int rng_set_table(const void *tab, int n) override {
    if (!tab || n != V.maxrand) return fail(FPX_ERR_ARG, "rng_set_table: need maxrand = 1000000 values");
    std::vector<R> t(n);
    tab4.clear(); tab8.clear();
    if (cfg.host_real_bytes == 4) { tab4.assign((const float *)tab, (const float *)tab + n); for (int i = 0; i < n; i++) t[i] = (R)tab4[i]; }
    else { tab8.assign((const double *)tab, (const double *)tab + n); for (int i = 0; i < n; i++) t[i] = (R)tab8[i]; }
    HIPCHK(hipMemcpyAsync((void *)V.rannumb, t.data(), n * sizeof(R), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    table_set = true;
    return 0;
  }
  int rng_get_table(void *out, int n) override {
    if (!out || n != V.maxrand || !table_set) return fail(FPX_ERR_ARG, "rng_get_table: no table / bad size");
    if (cfg.host_real_bytes == 4) memcpy(out, tab4.data(), n * sizeof(float));
    else memcpy(out, tab8.data(), n * sizeof(double));
    return 0;
  }

  // ---- particles -----------------------------------------------------------
  template <typename H, typename D>
  int put(const H *host, D *dev, long long first, long long count) {
    int rc = ensure_staging((size_t)count * sizeof(H));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(staging, host, (size_t)count * sizeof(H), hipMemcpyHostToDevice, stream));
    k_scatter_in<H, D><<<(int)((count + kBlock - 1) / kBlock), kBlock, 0, stream>>>((const H *)staging, dev, slot_map(), first, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  template <typename H, typename D>
  int get(H *host, const D *dev, long long first, long long count) {
    int rc = ensure_staging((size_t)count * sizeof(H));
    if (rc) return rc;
    k_gather_out<D, H><<<(int)((count + kBlock - 1) / kBlock), kBlock, 0, stream>>>(dev, (H *)staging, slot_map(), first, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host, staging, (size_t)count * sizeof(H), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  template <typename D>
  int fill(D *dev, D val, long long first, long long count) {
    k_fill<D><<<(int)((count + kBlock - 1) / kBlock), kBlock, 0, stream>>>(dev, val, first, count, slot_map());
    HIPCHK(hipGetLastError());
    return 0;
  }
  int put_real(const void *host, R *dev, long long first, long long count) {
    if (!host) return fill<R>(dev, (R)0, first, count);
    return cfg.host_real_bytes == 4 ? put<float, R>((const float *)host, dev, first, count) : put<double, R>((const double *)host, dev, first, count);
  }
  int get_real(void *host, const R *dev, long long first, long long count) {
    if (!host) return 0;
    return cfg.host_real_bytes == 4 ? get<float, R>((float *)host, dev, first, count) : get<double, R>((double *)host, dev, first, count);
  }

  int upload_particles(long long first, long long count, const fpx_particles *p) override {
    if (!p || first < 0 || count < 0 || first + count > P.cap) return fail(FPX_ERR_ARG, "upload_particles: range outside capacity");
    if (count == 0) return 0;
    if (!p->xtra1 || !p->ytra1 || !p->ztra1 || !p->itra1) return fail(FPX_ERR_ARG, "upload_particles: xtra1, ytra1, ztra1, itra1 are required");
    int rc;
    if ((rc = put<double, double>(p->xtra1, P.xt, first, count))) return rc;
    if ((rc = put<double, double>(p->ytra1, P.yt, first, count))) return rc;
    if ((rc = put_real(p->ztra1, P.zt, first, count))) return rc;
    if ((rc = put_real(p->uap, P.up, first, count))) return rc;
    if ((rc = put_real(p->ucp, P.vp, first, count))) return rc;
    if ((rc = put_real(p->uzp, P.wp, first, count))) return rc;
    if ((rc = put_real(p->us, P.us, first, count))) return rc;
    if ((rc = put_real(p->vs, P.vs, first, count))) return rc;
    if ((rc = put_real(p->ws, P.ws, first, count))) return rc;
    if ((rc = put<int, int>(p->itra1, P.itra1, first, count))) return rc;
    if (p->itramem) { if ((rc = put<int, int>(p->itramem, P.itramem, first, count))) return rc; } else if ((rc = fill<int>(P.itramem, 0, first, count))) return rc;
    if (p->idt) { if ((rc = put<int, int>(p->idt, P.idt, first, count))) return rc; } else if ((rc = fill<int>(P.idt, 0, first, count))) return rc;
    if (p->npoint) { if ((rc = put<int, int>(p->npoint, P.npoint, first, count))) return rc; } else if ((rc = fill<int>(P.npoint, 1, first, count))) return rc;
    if (p->nclass) { if ((rc = put<int, int>(p->nclass, P.nclass, first, count))) return rc; } else if ((rc = fill<int>(P.nclass, 1, first, count))) return rc;
    if (p->itrasplit) { if ((rc = put<int, int>(p->itrasplit, P.itrasplit, first, count))) return rc; }
    else if ((rc = fill<int>(P.itrasplit, (cfg.ldirect < 0 ? -1 : 1) * 999999999, first, count))) return rc;   // "never", in the run's direction
    if (p->cbt) { if ((rc = put<short, short>(p->cbt, P.cbt, first, count))) return rc; } else if ((rc = fill<short>(P.cbt, (short)1, first, count))) return rc;
    for (int ks = 0; ks < cfg.nspec; ks++) {
      R *dst = P.xmass1 + (size_t)ks * P.cap;
      if (p->xmass1) {
        const char *src = (const char *)p->xmass1 + (size_t)ks * p->xmass1_ld * cfg.host_real_bytes;
        if ((rc = put_real(src, dst, first, count))) return rc;
      } else if ((rc = fill<R>(dst, (R)1, first, count))) return rc;
    }
    if (P.xscav)
      for (int ks = 0; ks < cfg.nspec; ks++) {
        R *dst = P.xscav + (size_t)ks * P.cap;
        if (p->xscav_frac1) {
          const char *src = (const char *)p->xscav_frac1 + (size_t)ks * p->xmass1_ld * cfg.host_real_bytes;
          if ((rc = put_real(src, dst, first, count))) return rc;
        } else if ((rc = fill<R>(dst, (R)-1, first, count))) return rc;     // as releaseparticles.f90:167-171 leaves a new particle
      }
    HIPCHK(hipStreamSynchronize(stream));
    numpart = std::max(numpart, first + count);
    maybe_new = true; births_unknown = true;
    return 0;
  }

  int download_particles(long long first, long long count, const fpx_particles *p) override {
    if (!p || first < 0 || count < 0 || first + count > P.cap) return fail(FPX_ERR_ARG, "download_particles: range outside capacity");
    if (count == 0) return 0;
    int rc;
    if (p->xtra1 && (rc = get<double, double>(p->xtra1, P.xt, first, count))) return rc;
    if (p->ytra1 && (rc = get<double, double>(p->ytra1, P.yt, first, count))) return rc;
    if ((rc = get_real(p->ztra1, P.zt, first, count))) return rc;
    if ((rc = get_real(p->uap, P.up, first, count))) return rc;
    if ((rc = get_real(p->ucp, P.vp, first, count))) return rc;
    if ((rc = get_real(p->uzp, P.wp, first, count))) return rc;
    if ((rc = get_real(p->us, P.us, first, count))) return rc;
    if ((rc = get_real(p->vs, P.vs, first, count))) return rc;
    if ((rc = get_real(p->ws, P.ws, first, count))) return rc;
    if (p->itra1 && (rc = get<int, int>(p->itra1, P.itra1, first, count))) return rc;
    if (p->itramem && (rc = get<int, int>(p->itramem, P.itramem, first, count))) return rc;
    if (p->idt && (rc = get<int, int>(p->idt, P.idt, first, count))) return rc;
    if (p->npoint && (rc = get<int, int>(p->npoint, P.npoint, first, count))) return rc;
    if (p->nclass && (rc = get<int, int>(p->nclass, P.nclass, first, count))) return rc;
    if (p->itrasplit && (rc = get<int, int>(p->itrasplit, P.itrasplit, first, count))) return rc;
    if (p->cbt && (rc = get<short, short>(p->cbt, P.cbt, first, count))) return rc;
    if (p->xmass1)
      for (int ks = 0; ks < cfg.nspec; ks++) {
        char *dst = (char *)p->xmass1 + (size_t)ks * p->xmass1_ld * cfg.host_real_bytes;
        if ((rc = get_real(dst, P.xmass1 + (size_t)ks * P.cap, first, count))) return rc;
      }
    if (p->xscav_frac1 && P.xscav)
      for (int ks = 0; ks < cfg.nspec; ks++) {
        char *dst = (char *)p->xscav_frac1 + (size_t)ks * p->xmass1_ld * cfg.host_real_bytes;
        if ((rc = get_real(dst, P.xscav + (size_t)ks * P.cap, first, count))) return rc;
      }
    return 0;
  }

  // ---- convective mixing -------------------------------------------------------------------------
  bool conv_on = false;
  int conv_nuvz = 0, conv_nconvlev = 0;
  void *conv_fld[5][2] = {};                 // ps, tt2, td2, tth, qvh of the two slots, compact, host real kind
  bool conv_slot[2] = {false, false};
  void *conv_tab[4] = {};                    // akz, bkz, akm, bkm
  void *conv_cb = nullptr;                   // cbaseflux [ny][nx]
  void *conv_fld_n[kMaxNests][5][2] = {};    // the same five arrays of every nested wind field, compact [nyn][nxn]
  bool conv_slot_n[kMaxNests][2] = {};
  void *conv_cb_n[kMaxNests] = {};           // cbasefluxn(:,:,l)
  size_t conv_ncol_alloc = 0;                // columns (all domains) the per-column arrays are sized for
  int *conv_pcol = nullptr, *conv_act = nullptr, *conv_lconv = nullptr, *conv_ntop = nullptr, *conv_cflag = nullptr, *conv_ntop_raw = nullptr;
  int4 *conv_colslot = nullptr;
  unsigned int *conv_flag = nullptr, *conv_rank = nullptr;
  unsigned char *conv_draws = nullptr;
  void *conv_rn = nullptr;
  void *conv_scr = nullptr, *conv_scan_tmp = nullptr;
  size_t conv_scr_bytes = 0, conv_scan_bytes = 0;
  unsigned long long *conv_nmoved = nullptr;
  double conv_last_ms = 0;
  double conv_ms() override { return conv_last_ms; }

  int conv_init(const fpx_conv_config *c) override {
    if (!c || c->struct_bytes != (int32_t)sizeof(fpx_conv_config)) return fail(FPX_ERR_ARG, "conv_init: null or fpx_conv_config size mismatch (ABI)");
    if (c->nuvz < 4 || c->nconvlev < 2 || c->nconvlev > c->nuvz - 2) return fail(FPX_ERR_ARG, "conv_init: need 2 <= nconvlev <= nuvz - 2");
    if (!c->akz || !c->bkz || !c->akm || !c->bkm) return fail(FPX_ERR_ARG, "conv_init: akz, bkz, akm, bkm are required");
    if (conv_on) return fail(FPX_ERR_STATE, "conv_init: already initialised");
    const size_t hb = (size_t)cfg.host_real_bytes, n2 = (size_t)cfg.nx * cfg.ny, n3 = n2 * c->nuvz;
    int rc;
    const void *tabs[4] = {c->akz, c->bkz, c->akm, c->bkm};
    for (int i = 0; i < 4; i++) {
      unsigned char *q = nullptr;
      if ((rc = dalloc(&q, (size_t)c->nuvz * hb))) return rc;
      HIPCHK(hipMemcpyAsync(q, tabs[i], (size_t)c->nuvz * hb, hipMemcpyHostToDevice, stream));
      conv_tab[i] = q;
    }
    for (int f = 0; f < 5; f++)
      for (int sl = 0; sl < 2; sl++) {
        unsigned char *q = nullptr;
        if ((rc = dalloc(&q, (f < 3 ? n2 : n3) * hb))) return rc;
        conv_fld[f][sl] = q;
      }
    {
      unsigned char *q = nullptr;
      if ((rc = dalloc(&q, n2 * hb))) return rc;
      HIPCHK(hipMemsetAsync(q, 0, n2 * hb, stream));
      conv_cb = q;
    }
    if ((rc = dalloc(&conv_pcol, (size_t)P.cap)) || (rc = dalloc(&conv_draws, (size_t)P.cap))) return rc;
    if ((rc = dalloc(&conv_nmoved, (size_t)1))) return rc;
    {
      unsigned char *q = nullptr;
      if ((rc = dalloc(&q, (size_t)P.cap * hb))) return rc;
      conv_rn = q;
    }
    HIPCHK(hipStreamSynchronize(stream));
    conv_nuvz = c->nuvz; conv_nconvlev = c->nconvlev;
    conv_on = true;
    return 0;
  }

  template <typename H>
  int upload_conv_fields_t(int slot, const fpx_conv_fields *f) {
    const size_t n2max = (size_t)cfg.nxmax * cfg.nymax;
    const void *src[5] = {f->ps, f->tt2, f->td2, f->tth, f->qvh};
    for (int i = 0; i < 5; i++) {
      const int nlev = i < 3 ? 1 : conv_nuvz;
      const size_t bytes = n2max * (size_t)(i < 3 ? 1 : f->nuvzmax) * sizeof(H);
      int rc = ensure_staging(bytes);
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(staging, src[i], bytes, hipMemcpyHostToDevice, stream));
      const long long n = (long long)cfg.nx * cfg.ny * nlev;
      k_conv_pack<H, H><<<(int)((n + kBlock - 1) / kBlock), kBlock, 0, stream>>>((const H *)staging, (H *)conv_fld[i][slot - 1], cfg.nx, cfg.ny, nlev, cfg.nxmax, cfg.nymax);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(stream));        // the staging buffer is reused by the next array
    }
    conv_slot[slot - 1] = true;
    return 0;
  }
  int upload_conv_fields(int slot, const fpx_conv_fields *f) override {
    if (!conv_on) return fail(FPX_ERR_STATE, "upload_conv_fields: fpx_conv_init first");
    if (slot != 1 && slot != 2) return fail(FPX_ERR_ARG, "upload_conv_fields: slot must be 1 or 2");
    if (!f || !f->ps || !f->tt2 || !f->td2 || !f->tth || !f->qvh) return fail(FPX_ERR_ARG, "upload_conv_fields: ps, tt2, td2, tth, qvh are required");
    if (f->nuvzmax < conv_nuvz) return fail(FPX_ERR_ARG, "upload_conv_fields: nuvzmax < nuvz");
    return cfg.host_real_bytes == 4 ? upload_conv_fields_t<float>(slot, f) : upload_conv_fields_t<double>(slot, f);
  }

  template <typename H>
  int upload_conv_nest_fields_t(int nest, int slot, const fpx_conv_fields *f) {
    const int l = nest - 1;
    const int nxn = h_nest[l].nx, nyn = h_nest[l].ny;
    const size_t n2 = (size_t)nxn * nyn, n2max = (size_t)nest_nxmaxn * nest_nymaxn;
    int rc;
    if (!conv_cb_n[l]) {
      for (int i = 0; i < 5; i++)
        for (int sl = 0; sl < 2; sl++) {
          H *q = nullptr;
          if ((rc = dalloc(&q, i < 3 ? n2 : n2 * conv_nuvz))) return rc;
          conv_fld_n[l][i][sl] = q;
        }
      H *q = nullptr;
      if ((rc = dalloc(&q, n2))) return rc;
      HIPCHK(hipMemsetAsync(q, 0, n2 * sizeof(H), stream));
      conv_cb_n[l] = q;
    }
    const void *src[5] = {f->ps, f->tt2, f->td2, f->tth, f->qvh};
    for (int i = 0; i < 5; i++) {
      const int nlev = i < 3 ? 1 : conv_nuvz;
      const size_t bytes = n2max * (size_t)(i < 3 ? 1 : f->nuvzmax) * sizeof(H);
      if ((rc = ensure_staging(bytes))) return rc;
      HIPCHK(hipMemcpyAsync(staging, src[i], bytes, hipMemcpyHostToDevice, stream));
      const long long n = (long long)n2 * nlev;
      k_conv_pack<H, H><<<(int)((n + kBlock - 1) / kBlock), kBlock, 0, stream>>>((const H *)staging, (H *)conv_fld_n[l][i][slot - 1], nxn, nyn, nlev, nest_nxmaxn, nest_nymaxn);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(stream));
    }
    conv_slot_n[l][slot - 1] = true;
    return 0;
  }
  int upload_conv_nest_fields(int nest, int slot, const fpx_conv_fields *f) override {
    if (!conv_on) return fail(FPX_ERR_STATE, "upload_conv_nest_fields: fpx_conv_init first");
    if (nest < 1 || nest > V.numbnests) return fail(FPX_ERR_ARG, "upload_conv_nest_fields: nest out of range (fpx_nests_init first)");
    if (slot != 1 && slot != 2) return fail(FPX_ERR_ARG, "upload_conv_nest_fields: slot must be 1 or 2");
    if (!f || !f->ps || !f->tt2 || !f->td2 || !f->tth || !f->qvh) return fail(FPX_ERR_ARG, "upload_conv_nest_fields: ps, tt2, td2, tth, qvh are required");
    if (f->nuvzmax < conv_nuvz) return fail(FPX_ERR_ARG, "upload_conv_nest_fields: nuvzmax < nuvz");
    return cfg.host_real_bytes == 4 ? upload_conv_nest_fields_t<float>(nest, slot, f) : upload_conv_nest_fields_t<double>(nest, slot, f);
  }
  int cbaseflux_nest_io(int nest, void *host, bool set) override {
    if (!conv_on) return fail(FPX_ERR_STATE, "cbaseflux_nest: fpx_conv_init first");
    if (nest < 1 || nest > V.numbnests || !conv_cb_n[nest - 1]) return fail(FPX_ERR_STATE, "cbaseflux_nest: fpx_upload_conv_nest_fields of this nest first");
    if (!host) return fail(FPX_ERR_ARG, "cbaseflux_nest: null");
    const size_t bytes = (size_t)h_nest[nest - 1].nx * h_nest[nest - 1].ny * cfg.host_real_bytes;
    if (set) HIPCHK(hipMemcpyAsync(conv_cb_n[nest - 1], host, bytes, hipMemcpyHostToDevice, stream));
    else HIPCHK(hipMemcpyAsync(host, conv_cb_n[nest - 1], bytes, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  int cbaseflux_io(void *host, bool set) override {
    if (!conv_on) return fail(FPX_ERR_STATE, "cbaseflux: fpx_conv_init first");
    if (!host) return fail(FPX_ERR_ARG, "cbaseflux: null");
    const size_t bytes = (size_t)cfg.nx * cfg.ny * cfg.host_real_bytes;
    if (set) HIPCHK(hipMemcpyAsync(conv_cb, host, bytes, hipMemcpyHostToDevice, stream));
    else HIPCHK(hipMemcpyAsync(host, conv_cb, bytes, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  // sort2.f90 (the reference's quicksort with insertion sort below 7 elements; not stable): the order in which convmix
  // visits the particles, hence the order of redist's draws from the shared ran3 stream.  Host side of the parity mode.
  static void conv_sort2(int n, int *arr, int *brr) {     // 1-based [1..n]
    const int M = 7, NSTACK = 50;
    int i, ir, j, jstack = 0, k, l = 1, istack[51], a, b;
    ir = n;
    for (;;) {
      if (ir - l < M) {
        for (j = l + 1; j <= ir; j++) {
          a = arr[j]; b = brr[j];
          for (i = j - 1; i >= 1; i--) {
            if (arr[i] <= a) break;
            arr[i + 1] = arr[i]; brr[i + 1] = brr[i];
          }
          if (i < 1) i = 0;
          arr[i + 1] = a; brr[i + 1] = b;
        }
        if (jstack == 0) return;
        ir = istack[jstack]; l = istack[jstack - 1]; jstack -= 2;
      } else {
        k = (l + ir) / 2;
        std::swap(arr[k], arr[l + 1]); std::swap(brr[k], brr[l + 1]);
        if (arr[l + 1] > arr[ir]) { std::swap(arr[l + 1], arr[ir]); std::swap(brr[l + 1], brr[ir]); }
        if (arr[l] > arr[ir]) { std::swap(arr[l], arr[ir]); std::swap(brr[l], brr[ir]); }
        if (arr[l + 1] > arr[l]) { std::swap(arr[l + 1], arr[l]); std::swap(brr[l + 1], brr[l]); }
        i = l + 1; j = ir; a = arr[l]; b = brr[l];
        for (;;) {
          do i++; while (arr[i] < a);
          do j--; while (arr[j] > a);
          if (j < i) break;
          std::swap(arr[i], arr[j]); std::swap(brr[i], brr[j]);
        }
        arr[l] = arr[j]; arr[j] = a; brr[l] = brr[j]; brr[j] = b;
        jstack += 2;
        if (jstack > NSTACK) return;
        if (ir - i + 1 >= j - l) { istack[jstack] = ir; istack[jstack - 1] = i; ir = j - 1; }
        else { istack[jstack] = j - 1; istack[jstack - 1] = l; l = i; }
      }
    }
  }

  // replay of the serial stream: particles by particle number, igrid as convmix builds it, the reference's sort2, ran3 for
  // the particles the probe pass found drawing
  template <typename H>
  int conv_replay(long long n, int ndom, const int *off) {
    std::vector<unsigned char> h_draws((size_t)n);
    std::vector<int> h_pcol((size_t)n);
    std::vector<unsigned int> h_pid((size_t)n);
    HIPCHK(hipMemcpyAsync(h_draws.data(), conv_draws, (size_t)n, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(h_pcol.data(), conv_pcol, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(h_pid.data(), P.pid, (size_t)n * sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<int> col_by_pid((size_t)n, -1), igrid((size_t)n + 1), ipoint((size_t)n + 1);
    std::vector<unsigned char> draws_by_pid((size_t)n, 0);
    for (long long s = 0; s < n; s++) {
      const unsigned int p = h_pid[(size_t)s];
      if (p >= (unsigned int)n) return fail(FPX_ERR_STATE, "convmix: particle numbers beyond numpart");
      col_by_pid[p] = h_pcol[(size_t)s];
      draws_by_pid[p] = h_draws[(size_t)s];
    }
    std::vector<H> rn((size_t)n, (H)-1);
    for (int d = 0; d < ndom; d++) {          // the mother grid first, then nest by nest (convmix.f90:139-196, 198-250)
      for (long long p = 1; p <= n; p++) {
        const int c = col_by_pid[(size_t)p - 1];
        igrid[(size_t)p] = (c >= off[d] && c < off[d + 1]) ? 1 + c - off[d] : -1;
        ipoint[(size_t)p] = (int)p;
      }
      conv_sort2((int)n, igrid.data(), ipoint.data());
      for (long long k = 1; k <= n; k++) {
        if (igrid[(size_t)k] == -1) continue;
        const int p = ipoint[(size_t)k] - 1;
        if (!draws_by_pid[(size_t)p]) continue;
        rn[(size_t)p] = sizeof(H) == 4 ? (H)rng4.ran3(rng4.idummy_redist) : (H)rng8.ran3(rng8.idummy_redist);
      }
    }
    HIPCHK(hipMemcpyAsync(conv_rn, rn.data(), (size_t)n * sizeof(H), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  unsigned int *conv_alive = nullptr;
  size_t conv_alive_cap = 0;
  int conv_last_active = 0, conv_last_survivors = 0;

  template <typename H>
  int convmix_t(int itime, int64_t *nmoved_out) {
    const long long n = numpart;
    const int nx = cfg.nx, ny = cfg.ny;
    const int nv = conv_nuvz + 2;
    // wind-field domains: the mother grid and the nests (fpx_nests_init); columns numbered through all of them
    conv::Fields<H> F;
    int off[conv::kConvMaxDom + 1];
    F.ndom = 1 + V.numbnests;
    off[0] = 0;
    for (int d = 0; d < F.ndom; d++) {
      conv::Dom<H> &D = F.dom[d];
      void *(*fld)[2] = d == 0 ? conv_fld : conv_fld_n[d - 1];
      for (int sl = 0; sl < 2; sl++) {
        D.ps[sl] = (const H *)fld[0][sl]; D.tt2[sl] = (const H *)fld[1][sl]; D.td2[sl] = (const H *)fld[2][sl];
        D.tth[sl] = (const H *)fld[3][sl]; D.qvh[sl] = (const H *)fld[4][sl];
      }
      D.cb = (H *)(d == 0 ? conv_cb : conv_cb_n[d - 1]);
      D.nx = d == 0 ? nx : h_nest[d - 1].nx; D.ny = d == 0 ? ny : h_nest[d - 1].ny;
      D.off = off[d];
      off[d + 1] = off[d] + D.nx * D.ny;
      if (d > 0) {
        const NestDesc<R> &N = h_nest[d - 1];
        D.xl = (H)N.xl; D.yl = (H)N.yl; D.xr = (H)N.xr; D.yr = (H)N.yr; D.xres = (H)N.xres; D.yres = (H)N.yres;
      } else { D.xl = D.yl = D.xr = D.yr = (H)0; D.xres = D.yres = (H)1; }
    }
    const int ncol = off[F.ndom];
    F.akz = (const H *)conv_tab[0]; F.bkz = (const H *)conv_tab[1]; F.akm = (const H *)conv_tab[2]; F.bkm = (const H *)conv_tab[3];
    F.nuvz = conv_nuvz; F.nconvlev = conv_nconvlev;
    F.m1 = V.m1; F.m2 = V.m2;
    F.dt1 = (H)(itime - V.memtime0); F.dt2 = (H)(V.memtime1 - itime);
    F.dtt = (H)1. / (F.dt1 + F.dt2);
    F.delt = (H)std::abs(cfg.lsynctime);
    F.eps = (H)cfg.par_nxmax / (H)3.e5;
    if ((size_t)ncol > conv_ncol_alloc) {
      int rc2;
      if ((rc2 = dalloc(&conv_flag, (size_t)ncol)) || (rc2 = dalloc(&conv_rank, (size_t)ncol)) || (rc2 = dalloc(&conv_act, (size_t)ncol)) ||
          (rc2 = dalloc(&conv_lconv, (size_t)ncol)) || (rc2 = dalloc(&conv_ntop, (size_t)ncol)) || (rc2 = dalloc(&conv_cflag, (size_t)ncol)) ||
          (rc2 = dalloc(&conv_ntop_raw, (size_t)ncol)) || (rc2 = dalloc(&conv_colslot, (size_t)ncol))) return rc2;
      conv_ncol_alloc = (size_t)ncol;
    }
    const int nb = (int)((n + kBlock - 1) / kBlock), nbc = (ncol + kBlock - 1) / kBlock;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{e0, e1};
    HIPCHK(hipEventRecord(e0, stream));
    HIPCHK(hipMemsetAsync(conv_flag, 0, (size_t)ncol * sizeof(unsigned int), stream));
    HIPCHK(hipMemsetAsync(conv_nmoved, 0, sizeof(unsigned long long), stream));
    conv::k_conv_mark<R, H><<<nb, kBlock, 0, stream>>>(F, P.xt, P.yt, P.itra1, n, itime, conv_pcol, conv_flag);
    HIPCHK(hipGetLastError());
    size_t tb = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tb, conv_flag, conv_rank, 0u, (size_t)ncol, rocprim::plus<unsigned int>(), stream));
    if (tb > conv_scan_bytes) {
      if (conv_scan_tmp) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(conv_scan_tmp)); conv_scan_tmp = nullptr; }
      HIPCHK(hipMalloc(&conv_scan_tmp, std::max<size_t>(tb, 16)));
      conv_scan_bytes = std::max<size_t>(tb, 16);
    }
    HIPCHK(rocprim::exclusive_scan(conv_scan_tmp, tb, conv_flag, conv_rank, 0u, (size_t)ncol, rocprim::plus<unsigned int>(), stream));
    unsigned int last_rank = 0, last_flag = 0;
    HIPCHK(hipMemcpyAsync(&last_rank, conv_rank + (ncol - 1), 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&last_flag, conv_flag + (ncol - 1), 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    const int nact = (int)(last_rank + last_flag);
    if (nact == 0) { if (nmoved_out) *nmoved_out = 0; conv_last_ms = 0; return 0; }
    conv::k_conv_list<<<nbc, kBlock, 0, stream>>>(conv_flag, conv_rank, ncol, conv_act);
    HIPCHK(hipGetLastError());
    // scratch: the vectors of every column that holds particles; the matrices only for the columns that get past CONVECT's
    // early exits, in batches that fit the budget (default: a quarter of the free device memory)
    const size_t vec_elems = conv::vec_elems_per_column<H>(nv) * conv::group_round((size_t)nact);
    const size_t cst_elems = conv::group_round((size_t)conv::C_COUNT * nact);
    const size_t vec_bytes = (vec_elems + cst_elems) * sizeof(H);
    const size_t per_mat = conv::mat_elems_per_column<H>(nv) * sizeof(H);
    size_t budget;
    {
      size_t free_b = 0, total_b = 0;
      HIPCHK(hipMemGetInfo(&free_b, &total_b));
      budget = (free_b + conv_scr_bytes) / 4;
      if (opt.conv_scratch_mb > 0) budget = (size_t)opt.conv_scratch_mb << 20;
    }
    if ((size_t)nact > conv_alive_cap) {
      if (conv_alive) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(conv_alive)); conv_alive = nullptr; }
      HIPCHK(hipMalloc(&conv_alive, (size_t)nact * 3 * sizeof(unsigned int)));
      conv_alive_cap = (size_t)nact;
    }
    unsigned int *alive = conv_alive, *srank = conv_alive + nact;
    int *surv = (int *)(conv_alive + 2 * (size_t)nact);
    auto ensure_scratch = [&](size_t bytes) -> int {
      if (bytes <= conv_scr_bytes) return 0;
      if (conv_scr) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(conv_scr)); conv_scr = nullptr; conv_scr_bytes = 0; }
      hipError_t e = hipMalloc(&conv_scr, bytes);
      if (e != hipSuccess) return fail(FPX_ERR_NOMEM, std::string("convmix: scratch: ") + hipGetErrorString(e));
      conv_scr_bytes = bytes;
      return 0;
    };
    // upper bound of the matrix batch before the survivors are known: all active columns, capped by the budget
    int Bm_cap = (int)std::min<size_t>(conv::group_round((size_t)nact), std::max<size_t>(64, budget / per_mat / conv::kGroup * conv::kGroup));
    int rc = ensure_scratch(vec_bytes + (size_t)Bm_cap * per_mat);
    if (rc) return rc;
    H *vbuf = (H *)conv_scr, *cst = vbuf + vec_elems, *mbuf = cst + cst_elems;
    const H height_nz = (H)height_host[cfg.nz - 1];
    const bool seq = cfg.rng_mode == FPX_RNG_TABLE_SEQ;
    const bool conv_one_lane = opt.conv_one_lane != 0;     // the one-lane-per-column kernel (kept as the check of the level-parallel ones)
    // fmassfrac stored along the rows (forward runs) / columns (backward) the particles walk (k_conv_matrix_walk, _walk_t);
    // FPX_CONV_NO_WALK=1: the interleaved form
    const bool conv_walk = !conv_one_lane && !opt.conv_no_walk;
    const bool conv_rows_plain = opt.conv_rows_plain != 0;  // k_conv_rows without the LDS staging of its operands
    conv::k_conv_column_a<H><<<conv::serial_grid(nact), 64, 0, stream>>>(F, vbuf, cst, nv, conv_act, nact, alive);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(conv_lconv, 0, (size_t)nact * sizeof(int), stream));
    HIPCHK(hipMemsetAsync(conv_ntop, 0, (size_t)nact * sizeof(int), stream));
    tb = conv_scan_bytes;
    HIPCHK(rocprim::exclusive_scan(conv_scan_tmp, tb, alive, srank, 0u, (size_t)nact, rocprim::plus<unsigned int>(), stream));
    HIPCHK(hipMemcpyAsync(&last_rank, srank + (nact - 1), 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&last_flag, alive + (nact - 1), 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    const int nsurv = (int)(last_rank + last_flag);
    conv_last_active = nact; conv_last_survivors = nsurv;
    if (nsurv > 0) {
      conv::k_conv_survivors<<<(nact + kBlock - 1) / kBlock, kBlock, 0, stream>>>(alive, srank, nact, surv);
      HIPCHK(hipGetLastError());
      const int Bm = std::min(nsurv, Bm_cap);
      if (seq && nsurv > Bm) return fail(FPX_ERR_UNSUPPORTED, "convmix: the serial-stream parity mode needs all convective columns in one scratch batch (raise FPX_CONV_SCRATCH_MB)");
      // parity mode: the random numbers come from the shared serial stream in the reference's visiting order, which needs to
      // know who draws -- a probe pass records that, the host replays the stream, then the particles are moved
      for (int m0 = 0; m0 < nsurv; m0 += Bm) {
        if (conv_one_lane) {
          conv::k_conv_column_b<H><<<(Bm + 63) / 64, 64, 0, stream>>>(F, vbuf, mbuf, cst, nv, nact, conv_act, surv, m0, Bm, nsurv, conv_lconv, conv_ntop);
        } else {
          const int nlev = F.nconvlev + 1;
          const unsigned int gl = conv::level_grid(Bm, nlev);
          conv::k_conv_prelude<H><<<conv::serial_grid(Bm), 64, 0, stream>>>(F, vbuf, mbuf, cst, nv, nact, surv, m0, Bm, nsurv, conv_cflag, conv_ntop_raw);
          if (conv_rows_plain) conv::k_conv_rows<H><<<gl, 64, 0, stream>>>(vbuf, mbuf, cst, nv, nact, surv, m0, Bm, nsurv, nlev);
          else conv::k_conv_rows_lds<H><<<conv::level_grid(Bm, (nlev + conv::kRowsPerBlock - 1) / conv::kRowsPerBlock), dim3(64, conv::kRowsPerBlock), 0, stream>>>(
                   vbuf, mbuf, cst, nv, nact, surv, m0, Bm, nsurv, nlev);
          conv::k_conv_cols<H><<<gl, 64, 0, stream>>>(vbuf, mbuf, cst, nv, nact, surv, m0, Bm, nsurv, nlev, conv_ntop_raw);
          conv::k_conv_flux<H><<<gl, 64, 0, stream>>>(F, vbuf, mbuf, cst, nv, nact, surv, m0, Bm, nsurv, nlev, conv_cflag);
          if (conv_walk && cfg.ldirect == 1) conv::k_conv_matrix_walk<H><<<gl, 64, 0, stream>>>(F, vbuf, mbuf, cst, nv, nact, conv_act, surv, m0, Bm, nsurv, nlev, conv_cflag, conv_ntop_raw, conv_lconv, conv_ntop);
          else if (conv_walk) conv::k_conv_matrix_walk_t<H><<<conv::level_grid(Bm, (nlev + conv::kWalkRows - 1) / conv::kWalkRows), dim3(64, conv::kWalkRows), 0, stream>>>(
                                  F, vbuf, mbuf, cst, nv, nact, conv_act, surv, m0, Bm, nsurv, nlev, conv_cflag, conv_ntop_raw, conv_lconv, conv_ntop);
          else conv::k_conv_matrix<H><<<gl, 64, 0, stream>>>(F, vbuf, mbuf, cst, nv, nact, conv_act, surv, m0, Bm, nsurv, nlev, conv_cflag, conv_ntop_raw, conv_lconv, conv_ntop);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemsetAsync(conv_colslot, 0, (size_t)ncol * sizeof(int4), stream));
        conv::k_conv_slots<<<(nact + kBlock - 1) / kBlock, kBlock, 0, stream>>>(conv_act, alive, srank, conv_lconv, conv_ntop, nact, m0, Bm, conv_colslot);
        if (seq) {
          HIPCHK(hipMemsetAsync(conv_draws, 0, (size_t)n, stream));
          conv::k_conv_redist<R, H, ConvRngSeq<H>><<<nb, kBlock, 0, stream>>>(conv_pcol, conv_colslot, P.zt, n, vbuf, mbuf, nv, nact, Bm,
                                                                          cfg.ldirect, cfg.lsynctime, height_nz, ConvRngSeq<H>{(const H *)conv_rn, P.pid},
                                                                          conv_draws, 1, nullptr, conv_walk ? 1 : 0);
          HIPCHK(hipGetLastError());
          int rrc = conv_replay<H>(n, F.ndom, off);
          if (rrc) return rrc;
          conv::k_conv_redist<R, H, ConvRngSeq<H>><<<nb, kBlock, 0, stream>>>(conv_pcol, conv_colslot, P.zt, n, vbuf, mbuf, nv, nact, Bm,
                                                                          cfg.ldirect, cfg.lsynctime, height_nz, ConvRngSeq<H>{(const H *)conv_rn, P.pid},
                                                                          conv_draws, 0, conv_nmoved, conv_walk ? 1 : 0);
        } else {
          conv::k_conv_redist<R, H, ConvRngCtr<R, H>><<<nb, kBlock, 0, stream>>>(conv_pcol, conv_colslot, P.zt, n, vbuf, mbuf, nv, nact, Bm,
                                                                             cfg.ldirect, cfg.lsynctime, height_nz, ConvRngCtr<R, H>{V, P.pid, step_counter},
                                                                             conv_draws, 0, conv_nmoved, conv_walk ? 1 : 0);
        }
        HIPCHK(hipGetLastError());
      }
    }
    HIPCHK(hipEventRecord(e1, stream));
    unsigned long long moved = 0;
    HIPCHK(hipMemcpyAsync(&moved, conv_nmoved, sizeof(moved), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) conv_last_ms = ms;
    if (nmoved_out) *nmoved_out = (int64_t)moved;
    return 0;
  }

  int convmix(int itime, int64_t *nmoved) override {
    if (!conv_on) return fail(FPX_ERR_STATE, "convmix: fpx_conv_init first");
    if (!conv_slot[0] || !conv_slot[1]) return fail(FPX_ERR_STATE, "convmix: fpx_upload_conv_fields of both slots first");
    for (int l = 0; l < V.numbnests; l++)
      if (!conv_slot_n[l][0] || !conv_slot_n[l][1]) return fail(FPX_ERR_STATE, "convmix: fpx_upload_conv_nest_fields of both slots of every nest first");
    if (1 + V.numbnests > conv::kConvMaxDom) return fail(FPX_ERR_UNSUPPORTED, "convmix: too many nests");
    if (!window_set) return fail(FPX_ERR_STATE, "convmix: fpx_set_windtime first");
    if (!height_set) return fail(FPX_ERR_STATE, "convmix: set_height first");
    if (numpart == 0) { if (nmoved) *nmoved = 0; return 0; }
    return cfg.host_real_bytes == 4 ? convmix_t<float>(itime, nmoved) : convmix_t<double>(itime, nmoved);
  }

  // ---- lossless checkpoint (last clause) ------------------------------------------------------
  // partoutput / readpartpositions keep position, mass and age of a particle, rounded to the dump's real kind; the
  // turbulent velocity memory (up, vp, wp), the mesoscale components (us, vs, ws), cbt, idt, itramem, itrasplit,
  // nclass are re-initialised by a warm start (readpartpositions.f90:118-148), so the reference's restart is a
  // different realisation of the run.  This pair writes and reads everything the loop carries: all particle arrays in
  // the compute precision and in particle-number order, the step counter the counter RNG is keyed on, the state of
  // the serial ran3 / ran1 streams of the parity mode, and the accumulating output grids.  A run continued from it
  // is the run that was never interrupted, bit for bit.
  struct CkptHeader {
    char magic[8];
    int32_t version, real_bytes, nspec, rng_mode;
    int64_t numpart, particle_base;
    uint64_t seed;
    uint32_t step_counter;
    int32_t itime, numparticlecount, reserved;
    uint64_t n_grid3, n_grid2, n_grid3n, n_grid2n, n_receptor, rng_bytes;
    uint64_t cbase_bytes;        // conv_mod cbaseflux (the convection scheme relaxes it from call to call)
    int64_t rel_global_count;    // particles released so far by all ranks
    // version 2: what the arrays were sized with, and the length of the whole file (checked before anything is restored)
    int32_t nx, ny, nz, maxspec, numbnests, nxn[kMaxNests], nyn[kMaxNests], pad;
    uint64_t total_bytes;
  };
  void ckpt_describe_grid(CkptHeader &h) const {
    h.nx = cfg.nx; h.ny = cfg.ny; h.nz = cfg.nz; h.maxspec = cfg.maxspec; h.numbnests = V.numbnests;
    for (int l = 0; l < kMaxNests; l++) { h.nxn[l] = l < V.numbnests ? h_nest[l].nx : 0; h.nyn[l] = l < V.numbnests ? h_nest[l].ny : 0; }
    h.pad = P.xscav ? 1 : 0;      // the file carries xscav_frac1 (backward runs with DRYBKDEP / WETBKDEP)
  }
  // file length that goes with a header: header + RNG state + particle arrays + grids + receptors + cbaseflux
  static uint64_t ckpt_total_bytes(const CkptHeader &h) {
    const uint64_t per_particle = 2 * 8 + 7 * sizeof(R) + 6 * 4 + 2 + (uint64_t)h.nspec * sizeof(R) * (h.pad ? 2 : 1);   // pad = 1: xscav_frac1 follows xmass1
    return sizeof(CkptHeader) + h.rng_bytes + (uint64_t)h.numpart * per_particle + h.n_grid3 * sizeof(R) + 2 * h.n_grid2 * sizeof(float) +
           h.n_grid3n * sizeof(R) + 2 * h.n_grid2n * sizeof(float) + h.n_receptor * sizeof(R) + h.cbase_bytes;
  }
  // a restore that failed half-way leaves no usable state behind: no particles, no valid reductions
  int ckpt_invalidate(int rc) {
    numpart = 0;
    maybe_new = true; births_unknown = true;
    for (bool &v : red_valid) v = false;
    (void)hipStreamSynchronize(stream);
    const int nb = (int)((P.cap + kBlock - 1) / kBlock);
    k_fill<int><<<nb, kBlock, 0, stream>>>(P.itra1, kDead, 0, P.cap, nullptr);
    (void)hipStreamSynchronize(stream);
    g_err += " -- the engine holds no particles now (numpart = 0); restore from a complete checkpoint or release again";
    return rc;
  }
  static constexpr long long kCkptChunk = 1ll << 22;
  template <typename T>
  int ckpt_put_array(FILE *fh, const T *dev, long long n, std::vector<unsigned char> &buf) {
    for (long long f = 0; f < n; f += kCkptChunk) {
      const long long c = std::min(kCkptChunk, n - f);
      int rc = get<T, T>((T *)buf.data(), dev, f, c);
      if (rc) return rc;
      if (fwrite(buf.data(), sizeof(T), (size_t)c, fh) != (size_t)c) return fail(FPX_ERR_ARG, "checkpoint_write: write error");
    }
    return 0;
  }
  template <typename T>
  int ckpt_get_array(FILE *fh, T *dev, long long n, std::vector<unsigned char> &buf) {
    for (long long f = 0; f < n; f += kCkptChunk) {
      const long long c = std::min(kCkptChunk, n - f);
      if (fread(buf.data(), sizeof(T), (size_t)c, fh) != (size_t)c) return fail(FPX_ERR_ARG, "checkpoint_read: file too short");
      int rc = put<T, T>((const T *)buf.data(), dev, f, c);
      if (rc) return rc;
    }
    return 0;
  }
  template <typename T>
  int ckpt_put_plain(FILE *fh, const T *dev, size_t n, std::vector<unsigned char> &buf) {
    for (size_t f = 0; f < n; f += (size_t)kCkptChunk) {
      const size_t c = std::min((size_t)kCkptChunk, n - f);
      HIPCHK(hipMemcpyAsync(buf.data(), dev + f, c * sizeof(T), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      if (fwrite(buf.data(), sizeof(T), c, fh) != c) return fail(FPX_ERR_ARG, "checkpoint_write: write error");
    }
    return 0;
  }
  template <typename T>
  int ckpt_get_plain(FILE *fh, T *dev, size_t n, std::vector<unsigned char> &buf) {
    for (size_t f = 0; f < n; f += (size_t)kCkptChunk) {
      const size_t c = std::min((size_t)kCkptChunk, n - f);
      if (fread(buf.data(), sizeof(T), c, fh) != c) return fail(FPX_ERR_ARG, "checkpoint_read: file too short");
      HIPCHK(hipMemcpyAsync(dev + f, buf.data(), c * sizeof(T), hipMemcpyHostToDevice, stream));
      HIPCHK(hipStreamSynchronize(stream));
    }
    return 0;
  }
  struct CkptRng { HostRng<float> r4; HostRng<double> r8; Ran1 rel; };
  uint64_t conv_cbase_bytes() const {      // cbaseflux of the mother grid and of every nest that has convection fields
    if (!conv_on) return 0;
    uint64_t b = (uint64_t)cfg.nx * cfg.ny * cfg.host_real_bytes;
    for (int l = 0; l < V.numbnests; l++)
      if (conv_cb_n[l]) b += (uint64_t)h_nest[l].nx * h_nest[l].ny * cfg.host_real_bytes;
    return b;
  }

  int checkpoint_write(const char *path, int itime, int numparticlecount) override {
    if (!path) return fail(FPX_ERR_ARG, "checkpoint_write: path is required");
    FILE *fh = fopen(path, "wb");
    if (!fh) return fail(FPX_ERR_ARG, std::string("checkpoint_write: cannot open ") + path);
    struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } closer{fh};
    CkptHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "FPXCKPT1", 8);
    h.version = 2; h.real_bytes = (int)sizeof(R); h.nspec = cfg.nspec; h.rng_mode = cfg.rng_mode;
    h.numpart = numpart; h.particle_base = cfg.particle_base; h.seed = V.seed; h.step_counter = step_counter;
    h.itime = itime; h.numparticlecount = numparticlecount;
    h.n_grid3 = Gp.on ? n_grid3 : 0; h.n_grid2 = Gp.on ? n_grid2 : 0;
    h.n_grid3n = Gp.on && Gp.nested ? n_grid3n : 0; h.n_grid2n = Gp.on && Gp.nested ? n_grid2n : 0;
    h.n_receptor = Gp.creceptor ? (uint64_t)Gp.numreceptor * cfg.maxspec : 0;
    h.rng_bytes = sizeof(CkptRng);
    h.cbase_bytes = conv_cbase_bytes();
    h.rel_global_count = rel_global_count;
    ckpt_describe_grid(h);
    h.total_bytes = ckpt_total_bytes(h);
    if (fwrite(&h, sizeof(h), 1, fh) != 1) return fail(FPX_ERR_ARG, "checkpoint_write: write error");
    CkptRng rs{rng4, rng8, rel_ran1};
    if (fwrite(&rs, sizeof(rs), 1, fh) != 1) return fail(FPX_ERR_ARG, "checkpoint_write: write error");
    std::vector<unsigned char> buf((size_t)kCkptChunk * 8);
    const long long n = numpart;
    int rc;
    if ((rc = ckpt_put_array(fh, P.xt, n, buf)) || (rc = ckpt_put_array(fh, P.yt, n, buf))) return rc;
    for (R *a : {P.zt, P.up, P.vp, P.wp, P.us, P.vs, P.ws}) if ((rc = ckpt_put_array(fh, a, n, buf))) return rc;
    for (int *a : {P.idt, P.itra1, P.itramem, P.npoint, P.nclass, P.itrasplit}) if ((rc = ckpt_put_array(fh, a, n, buf))) return rc;
    if ((rc = ckpt_put_array(fh, P.cbt, n, buf))) return rc;
    for (int ks = 0; ks < cfg.nspec; ks++) if ((rc = ckpt_put_array(fh, P.xmass1 + (size_t)ks * P.cap, n, buf))) return rc;
    if (P.xscav) for (int ks = 0; ks < cfg.nspec; ks++) if ((rc = ckpt_put_array(fh, P.xscav + (size_t)ks * P.cap, n, buf))) return rc;
    if (h.n_grid3 && ((rc = ckpt_put_plain(fh, Gp.gridunc, n_grid3, buf)) || (rc = ckpt_put_plain(fh, Gp.drygridunc, n_grid2, buf)) ||
                      (rc = ckpt_put_plain(fh, Gp.wetgridunc, n_grid2, buf)))) return rc;
    if (h.n_grid3n && ((rc = ckpt_put_plain(fh, Gp.griduncn, n_grid3n, buf)) || (rc = ckpt_put_plain(fh, Gp.drygriduncn, n_grid2n, buf)) ||
                       (rc = ckpt_put_plain(fh, Gp.wetgriduncn, n_grid2n, buf)))) return rc;
    if (h.n_receptor && (rc = ckpt_put_plain(fh, Gp.creceptor, (size_t)h.n_receptor, buf))) return rc;
    if (h.cbase_bytes) {
      if ((rc = ckpt_put_plain(fh, (const unsigned char *)conv_cb, (size_t)cfg.nx * cfg.ny * cfg.host_real_bytes, buf))) return rc;
      for (int l = 0; l < V.numbnests; l++)
        if (conv_cb_n[l] && (rc = ckpt_put_plain(fh, (const unsigned char *)conv_cb_n[l], (size_t)h_nest[l].nx * h_nest[l].ny * cfg.host_real_bytes, buf))) return rc;
    }
    closer.f = nullptr;
    if (fclose(fh) != 0) return fail(FPX_ERR_ARG, std::string("checkpoint_write: write error on ") + path);
    return 0;
  }

  int checkpoint_read(const char *path, int32_t *itime, int64_t *numpart_out, int32_t *numparticlecount) override {
    if (!path) return fail(FPX_ERR_ARG, "checkpoint_read: path is required");
    FILE *fh = fopen(path, "rb");
    if (!fh) return fail(FPX_ERR_ARG, std::string("checkpoint_read: cannot open ") + path);
    struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } closer{fh};
    CkptHeader h;
    if (fread(&h, sizeof(h), 1, fh) != 1 || memcmp(h.magic, "FPXCKPT1", 8) != 0 || h.version != 2)
      return fail(FPX_ERR_ARG, "checkpoint_read: not a checkpoint of this engine (version 2)");
    if (h.real_bytes != (int)sizeof(R) || h.nspec != cfg.nspec || h.rng_mode != cfg.rng_mode || h.rng_bytes != sizeof(CkptRng))
      return fail(FPX_ERR_ARG, "checkpoint_read: written with another precision, species count or random-number mode");
    if (h.numpart < 0 || h.numpart > P.cap) return fail(FPX_ERR_ARG, "checkpoint_read: more particles than storage spaces");
    if (h.particle_base != cfg.particle_base || h.seed != V.seed) return fail(FPX_ERR_ARG, "checkpoint_read: another shard or seed");
    const uint64_t g3 = Gp.on ? n_grid3 : 0, g2 = Gp.on ? n_grid2 : 0, g3n = Gp.on && Gp.nested ? n_grid3n : 0, g2n = Gp.on && Gp.nested ? n_grid2n : 0;
    const uint64_t nr = Gp.creceptor ? (uint64_t)Gp.numreceptor * cfg.maxspec : 0;
    if (h.n_grid3 != g3 || h.n_grid2 != g2 || h.n_grid3n != g3n || h.n_grid2n != g2n || h.n_receptor != nr)
      return fail(FPX_ERR_STATE, "checkpoint_read: the output grids of the checkpoint are not the ones configured (call fpx_outgrid_init ... first)");
    if (h.cbase_bytes != conv_cbase_bytes())
      return fail(FPX_ERR_STATE, "checkpoint_read: the checkpoint was written with (without) convection; call fpx_conv_init first (or not at all)");
    {
      CkptHeader mine = h;
      ckpt_describe_grid(mine);
      if (memcmp(&mine.nx, &h.nx, (const char *)&h.pad - (const char *)&h.nx) != 0)
        return fail(FPX_ERR_ARG, "checkpoint_read: written on another grid (nx, ny, nz, maxspec or the nest extents differ)");
      if (mine.pad != h.pad) return fail(FPX_ERR_ARG, "checkpoint_read: written with (without) DRYBKDEP / WETBKDEP");
      // the whole file must be there before a single array is touched
      if (h.total_bytes != ckpt_total_bytes(h)) return fail(FPX_ERR_ARG, "checkpoint_read: inconsistent header");
      if (fseek(fh, 0, SEEK_END) != 0) return fail(FPX_ERR_ARG, "checkpoint_read: cannot seek");
      const long long len = (long long)ftell(fh);
      if (len < 0 || (uint64_t)len != h.total_bytes)
        return fail(FPX_ERR_ARG, "checkpoint_read: the file is truncated or has trailing bytes (" + std::to_string(len) + " bytes, header says " + std::to_string(h.total_bytes) + "); nothing was restored");
      if (fseek(fh, (long)sizeof(h), SEEK_SET) != 0) return fail(FPX_ERR_ARG, "checkpoint_read: cannot seek");
    }
    CkptRng rs;
    if (fread(&rs, sizeof(rs), 1, fh) != 1) return fail(FPX_ERR_ARG, "checkpoint_read: file too short");
    // storage spaces in particle-number order again
    HIPCHK(hipStreamSynchronize(stream));
    slot_of_pid = nullptr;
    slot_map_dirty = false;
    {
      const int nb = (int)((P.cap + kBlock - 1) / kBlock);
      k_iota_pid<<<nb, kBlock, 0, stream>>>(P.pid, 0, P.cap);
      k_fill<int><<<nb, kBlock, 0, stream>>>(P.itra1, kDead, 0, P.cap, nullptr);
      HIPCHK(hipGetLastError());
    }
    std::vector<unsigned char> buf((size_t)kCkptChunk * 8);
    const long long n = h.numpart;
    int rc;
    if ((rc = ckpt_get_array(fh, P.xt, n, buf)) || (rc = ckpt_get_array(fh, P.yt, n, buf))) return ckpt_invalidate(rc);
    for (R *a : {P.zt, P.up, P.vp, P.wp, P.us, P.vs, P.ws}) if ((rc = ckpt_get_array(fh, a, n, buf))) return ckpt_invalidate(rc);
    for (int *a : {P.idt, P.itra1, P.itramem, P.npoint, P.nclass, P.itrasplit}) if ((rc = ckpt_get_array(fh, a, n, buf))) return ckpt_invalidate(rc);
    if ((rc = ckpt_get_array(fh, P.cbt, n, buf))) return ckpt_invalidate(rc);
    for (int ks = 0; ks < cfg.nspec; ks++) if ((rc = ckpt_get_array(fh, P.xmass1 + (size_t)ks * P.cap, n, buf))) return ckpt_invalidate(rc);
    if (P.xscav) for (int ks = 0; ks < cfg.nspec; ks++) if ((rc = ckpt_get_array(fh, P.xscav + (size_t)ks * P.cap, n, buf))) return ckpt_invalidate(rc);
    if (g3 && ((rc = ckpt_get_plain(fh, Gp.gridunc, n_grid3, buf)) || (rc = ckpt_get_plain(fh, Gp.drygridunc, n_grid2, buf)) ||
               (rc = ckpt_get_plain(fh, Gp.wetgridunc, n_grid2, buf)))) return ckpt_invalidate(rc);
    if (g3n && ((rc = ckpt_get_plain(fh, Gp.griduncn, n_grid3n, buf)) || (rc = ckpt_get_plain(fh, Gp.drygriduncn, n_grid2n, buf)) ||
                (rc = ckpt_get_plain(fh, Gp.wetgriduncn, n_grid2n, buf)))) return ckpt_invalidate(rc);
    if (nr && (rc = ckpt_get_plain(fh, Gp.creceptor, (size_t)nr, buf))) return ckpt_invalidate(rc);
    if (h.cbase_bytes) {
      if ((rc = ckpt_get_plain(fh, (unsigned char *)conv_cb, (size_t)cfg.nx * cfg.ny * cfg.host_real_bytes, buf))) return ckpt_invalidate(rc);
      for (int l = 0; l < V.numbnests; l++)
        if (conv_cb_n[l] && (rc = ckpt_get_plain(fh, (unsigned char *)conv_cb_n[l], (size_t)h_nest[l].nx * h_nest[l].ny * cfg.host_real_bytes, buf))) return ckpt_invalidate(rc);
    }
    for (bool &v : red_valid) v = false;
    rng4 = rs.r4; rng8 = rs.r8; rel_ran1 = rs.rel;
    step_counter = h.step_counter;
    rel_global_count = h.rel_global_count;
    numpart = n;
    maybe_new = true; births_unknown = true;
    if (itime) *itime = h.itime;
    if (numpart_out) *numpart_out = n;
    if (numparticlecount) *numparticlecount = h.numparticlecount;
    return 0;
  }

  // point_mod zpoint1, zpoint2(numpoint): the height range of a release, which WETBKDEP multiplies the scavenging
  // coefficient with (timemanager.f90:590-591)
  R *d_zspan = nullptr;
  int zspan_n = 0;
  int set_release_heights(int numpoint, const void *zpoint1, const void *zpoint2) override {
    if (numpoint < 1 || !zpoint1 || !zpoint2) return fail(FPX_ERR_ARG, "set_release_heights: numpoint >= 1, zpoint1 and zpoint2 are required");
    std::vector<R> z((size_t)numpoint);
    for (int i = 0; i < numpoint; i++)
      z[i] = cfg.host_real_bytes == 4 ? (R)(((const float *)zpoint2)[i] - ((const float *)zpoint1)[i]) : (R)(((const double *)zpoint2)[i] - ((const double *)zpoint1)[i]);
    int rc;
    if (numpoint > zspan_n) { if ((rc = dalloc(&d_zspan, (size_t)numpoint))) return rc; zspan_n = numpoint; }
    HIPCHK(hipMemcpyAsync(d_zspan, z.data(), (size_t)numpoint * sizeof(R), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  // point_mod xmass(numpoint,maxspec), npart(numpoint): device tables indexed by npoint(j)
  int set_release_points(int numpoint, const void *xmass, const int32_t *npart) override {
    if (numpoint < 1 || !xmass || !npart) return fail(FPX_ERR_ARG, "set_release_points: numpoint >= 1, xmass and npart are required");
    std::vector<R> xm((size_t)numpoint * cfg.nspec);
    std::vector<signed char> nsp((size_t)numpoint);
    for (int kp = 0; kp < numpoint; kp++) {
      // advance.f90:518-524: first species with xmass(nrelpoint,nsp) > eps3 = tiny(1.0) of the host's real kind, else nspec
      int pick = cfg.nspec - 1;
      for (int ks = cfg.nspec - 1; ks >= 0; ks--) {
        const size_t i = (size_t)ks * numpoint + kp;
        bool above;
        if (cfg.host_real_bytes == 4) { const float v = ((const float *)xmass)[i]; xm[i] = (R)v; above = v > 1.17549435e-38f; }
        else { const double v = ((const double *)xmass)[i]; xm[i] = (R)v; above = v > 2.2250738585072014e-308; }
        if (above) pick = ks;
      }
      nsp[kp] = (signed char)pick;
    }
    HIPCHK(hipStreamSynchronize(stream));
    R *d_x; int *d_n; signed char *d_s;
    int rc;
    if ((rc = dalloc(&d_x, xm.size())) || (rc = dalloc(&d_n, (size_t)numpoint)) || (rc = dalloc(&d_s, (size_t)numpoint))) return rc;
    HIPCHK(hipMemcpyAsync(d_x, xm.data(), xm.size() * sizeof(R), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_n, npart, (size_t)numpoint * sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_s, nsp.data(), (size_t)numpoint, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    V.rel_xmass = d_x; V.rel_npart = d_n; V.rel_nsp = d_s; V.numpoint = numpoint;
    rel_xmass_h.resize(xm.size());
    for (size_t i = 0; i < xm.size(); i++) rel_xmass_h[i] = cfg.host_real_bytes == 4 ? (double)((const float *)xmass)[i] : ((const double *)xmass)[i];
    rel_npart_h.assign(npart, npart + numpoint);
    rel.set = false;         // the release tables refer to these points: fpx_release_init again
    return 0;
  }

  int set_numpart(long long n) override {
    if (n < 0 || n > P.cap) return fail(FPX_ERR_ARG, "set_numpart: outside capacity");
    if (n < numpart) { const int rc = restore_particle_order(); if (rc) return rc; }
    numpart = n;
    return 0;
  }

  int seed_particles(long long n, unsigned long long seed, double frac_pbl, double zmax, double lat_margin, int itime0) override {
    if (n < 1 || n > P.cap) return fail(FPX_ERR_ARG, "seed_particles: n outside capacity");
    if (!slot_loaded[0] || !slot_loaded[1]) return fail(FPX_ERR_STATE, "seed_particles: upload both field slots first (mixing heights are needed)");
    k_seed<R><<<(int)((n + kBlock - 1) / kBlock), kBlock, 0, stream>>>(V, P, n, seed, frac_pbl, zmax, lat_margin, itime0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    slot_of_pid = nullptr;
    numpart = n;
    maybe_new = true; births_unknown = true;
    return 0;
  }

  // ---- TABLE_SEQ: per-step start indices in the reference's serial order -----
  int prepare_seq(int itime) {
    const long long n = numpart;
    int rc;
    if (!d_flags) {
      const size_t cap = (size_t)P.cap;
      if ((rc = dalloc(&d_flags, cap))) return rc;
      if ((rc = dalloc(&d_nrand_adv, cap))) return rc;
      if ((rc = dalloc(&d_nrand_init, cap))) return rc;
      if ((rc = dalloc(&d_dcas4, cap))) return rc;
      if ((rc = dalloc(&d_dcas14, cap))) return rc;
      if ((rc = dalloc(&d_dcas8, cap))) return rc;
      if ((rc = dalloc(&d_dcas18, cap))) return rc;
    }
    h_flags.resize(n); h_nrand_adv.assign(n, 1); h_nrand_init.assign(n, 1);
    h_dcas4.assign(n, 0.f); h_dcas14.assign(n, 0.f); h_dcas8.assign(n, 0.); h_dcas18.assign(n, 0.);
    k_classify<R><<<(int)((n + kBlock - 1) / kBlock), kBlock, 0, stream>>>(V, P, n, itime, d_flags);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_flags.data(), d_flags, n, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    // walk the particles in index order exactly as the serial loop does
    // (timemanager.f90:531-611): [initialize: ran3 -> nrand, (ran3, gasdev)] then advance: ran3 -> nrand
    const bool h4 = cfg.host_real_bytes == 4;
    for (long long j = 0; j < n; j++) {
      unsigned char f = h_flags[j];
      if (!(f & 1)) continue;
      if (f & 2) {
        if (h4) {
          h_nrand_init[j] = rng4.start_index(rng4.idummy_init, V.maxrand);
          if (f & 4) { h_dcas4[j] = rng4.ran3(rng4.idummy_init); h_dcas14[j] = rng4.gasdev(rng4.idummy_init); }
        } else {
          h_nrand_init[j] = rng8.start_index(rng8.idummy_init, V.maxrand);
          if (f & 4) { h_dcas8[j] = rng8.ran3(rng8.idummy_init); h_dcas18[j] = rng8.gasdev(rng8.idummy_init); }
        }
      }
      h_nrand_adv[j] = h4 ? rng4.start_index(rng4.idummy_adv, V.maxrand) : rng8.start_index(rng8.idummy_adv, V.maxrand);
    }
    if (h4) for (long long j = 0; j < n; j++) { h_dcas8[j] = h_dcas4[j]; h_dcas18[j] = h_dcas14[j]; }
    else for (long long j = 0; j < n; j++) { h_dcas4[j] = (float)h_dcas8[j]; h_dcas14[j] = (float)h_dcas18[j]; }
    HIPCHK(hipMemcpyAsync(d_nrand_adv, h_nrand_adv.data(), n * sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_nrand_init, h_nrand_init.data(), n * sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_dcas4, h_dcas4.data(), n * sizeof(float), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_dcas14, h_dcas14.data(), n * sizeof(float), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_dcas8, h_dcas8.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_dcas18, h_dcas18.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  // ---- the step --------------------------------------------------------------
  int step(int itime, fpx_step_stats *out, bool async) override {
    if (!height_set || !window_set || !slot_loaded[0] || !slot_loaded[1]) return fail(FPX_ERR_STATE, "step: height, both field slots and the wind-time window must be set first");
    if (cfg.rng_mode != FPX_RNG_PHILOX && !table_set) return fail(FPX_ERR_STATE, "step: the table RNG modes need fpx_rng_fill_table/fpx_rng_set_table");
    if (V.memtime0 == V.memtime1) return fail(FPX_ERR_STATE, "step: empty wind-time window");
    if (cfg.wetbkdep && (!wet_on || !wet_slot[0] || !wet_slot[1])) return fail(FPX_ERR_STATE, "step: WETBKDEP needs the wet-deposition set-up (fpx_wet_init, fpx_upload_wet_fields for both slots)");
    if (cfg.wetbkdep && (!d_zspan || V.numpoint > zspan_n)) return fail(FPX_ERR_STATE, "step: WETBKDEP needs the release heights zpoint1, zpoint2 (fpx_set_release_heights)");
    if (cfg.drybkdep && !cfg.drydep) return fail(FPX_ERR_STATE, "step: DRYBKDEP without DRYDEP: there is no deposition velocity to take");
    if (V.numpoint == 0 && cfg.mdomainfill == 0 && cfg.mquasilag == 0)
      return fail(FPX_ERR_STATE, "step: the release-point tables xmass, npart are needed for the mass-fraction test (fpx_set_release_points)");
    for (int l = 0; l < V.numbnests; l++)
      if (!nest_loaded[l][0] || !nest_loaded[l][1]) return fail(FPX_ERR_STATE, "step: nest fields missing (fpx_upload_nest_fields, both slots)");
    if (out) memset(out, 0, sizeof(*out));
    if (numpart == 0) return 0;
    if (cfg.drydep) red_valid[RG_DRY] = red_valid[RG_DRYN] = false;
    if (cfg.sort_interval > 0 && step_counter > 0 && step_counter % (unsigned)cfg.sort_interval == 0) {
      int rc = sort_particles();
      if (rc) return rc;
    }
    SeqRng S;
    memset(&S, 0, sizeof(S));
    if (cfg.rng_mode == FPX_RNG_TABLE_SEQ) {
      int rc = prepare_seq(itime);
      if (rc) return rc;
      S.nrand_adv = d_nrand_adv; S.nrand_init = d_nrand_init;
      S.cbl_dcas = d_dcas4; S.cbl_dcas1 = d_dcas14; S.cbl_dcas_d = d_dcas8; S.cbl_dcas1_d = d_dcas18;
    }
    if (ev_used >= 64) {   // bound the pool in long runs that never ask for the timings: fold them into the totals
      int rc = harvest_events();
      if (rc) return rc;
    }
    if (ev_used == ev_pool.size()) {
      StepEvents se;
      for (int i = 0; i < 5; i++) HIPCHK(hipEventCreate(&se.e[i]));
      ev_pool.push_back(se);
    }
    auto &ev = ev_pool[ev_used++];
    const int nb = (int)((numpart + kBlock - 1) / kBlock);
    // the pass budgets of the Langevin kernel's launches (time slices, k_pbl_loop); the last launch has none
    if (slice_caps.empty()) {
      if (!opt.pbl_slices.empty()) slice_caps = opt.pbl_slices;
      else if (cfg.pbl_slice_passes < 0) slice_caps = {0};
      else if (cfg.pbl_slice_passes > 0) slice_caps = std::vector<int>(kMaxSlices - 1, cfg.pbl_slice_passes);
      else slice_caps = {FPX_SLICE_SCHEDULE};
      if ((int)slice_caps.size() > kMaxSlices - 1) slice_caps.resize(kMaxSlices - 1);
      if (slice_caps.empty() || slice_caps.back() != 0) slice_caps.push_back(0);
      if (slice_caps.size() > 1 && !d_surv[0]) {
        int rc;
        for (int k = 0; k < 2; k++) if ((rc = dalloc(&d_surv[k], (size_t)P.cap))) return rc;
      }
    }
    if (pbl_grid == 0) {
      // persistent grid: as many blocks as the chip holds at this kernel's register budget
      hipDeviceProp_t prop;
      HIPCHK(hipGetDeviceProperties(&prop, cfg.device));
      int per_cu = 0;
      HIPCHK(hipFuncSetAttribute((const void *)loop_kernel(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)loop_smem_bytes()));
      HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, loop_kernel(), kBlock, loop_smem_bytes()));
      if (opt.pbl_blocks_per_cu > 0) per_cu = std::min(per_cu, opt.pbl_blocks_per_cu);   // experiments: fewer resident waves
      pbl_grid = prop.multiProcessorCount * std::max(per_cu, 1);
      pbl_per_cu = std::max(per_cu, 1);
      if (opt.verbose) fprintf(stderr, "[fpx] Langevin kernel: %d blocks per CU by the occupancy query, %zu B of dynamic LDS, grid %d\n", per_cu, loop_smem_bytes(), pbl_grid);
    }
    {
      size_t need = 0;
      HIPCHK(rocprim::radix_sort_pairs(nullptr, need, d_pbl_flag, d_pbl_flag2, d_iota, d_pbl_list, (size_t)numpart, 0u, 6u, stream));
      if (need > sel_tmp_bytes) {
        if (d_sel_tmp) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(d_sel_tmp)); d_sel_tmp = nullptr; }
        HIPCHK(hipMalloc(&d_sel_tmp, need));
        sel_tmp_bytes = need;
      }
    }
    HIPCHK(hipMemsetAsync(d_pbl_ctr, 0, kCtrWords * sizeof(unsigned int), stream));
    // Cost buckets in the work list (k_prep): pure scheduling, no effect on any result, so the rank's own particle count
    // may decide.  They pay where particles per grid cell are few -- the shard one of eight GPUs runs: -4.5 % of the
    // Langevin kernel at 1.25e7 particles -- and cost locality: at 1e8 on one GPU they gain under 1 % and take the kernel's
    // HBM traffic from 1.04 to 2.4 times the algorithmic bytes (the level-pair fetches of neighbouring list entries no
    // longer share cache lines).
    V.pbl_cost_buckets = opt.pbl_cost_buckets >= 0 ? opt.pbl_cost_buckets : (numpart < 50000000ll ? 3 : 0);
    { const int rc = blend_winds(itime); if (rc) return rc; }      // its own kernel (k_blend_w3), ahead of the per-kernel events
    HIPCHK(hipEventRecord(ev.e[0], stream));
    if (P.xscav) {   // timemanager.f90:564-598, before the particle is moved
      k_bkdep<R><<<nb, kBlock, 0, stream>>>(V, Wp, P, numpart, itime, cfg.drybkdep, cfg.wetbkdep, d_zspan);
      HIPCHK(hipGetLastError());
    }
    {
      // specialised variants: dry deposition (aerosols), initialize() only when new particles can
      // exist (after an upload/seed or at itime 0), polar maps only on grids with poles
      if (births_unknown) { const int rc = find_birth_horizon(); if (rc) return rc; }
      const bool init = maybe_new || itime == 0 || (long long)cfg.ldirect * itime <= birth_horizon || opt.prep_init_always;
      const bool polar = cfg.nglobal || cfg.sglobal;
      typedef void (*prep_fn)(View<R>, GridP<R>, Parts<R>, SeqRng, PblRec<R>, long long, int, unsigned int, Stats *, unsigned char *, unsigned int *);
      const bool nest = V.numbnests > 0;
      const prep_fn f = (prep_fn)step_kernel_prep((int)sizeof(R), cfg.drydep != 0, init, polar, nest);
      const int prep_pad = opt.prep_lds_pad;   // experiments: unused dynamic LDS lowers the occupancy
      if (prep_pad > 0) HIPCHK(hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, prep_pad));
      f<<<nb, kBlock, (size_t)prep_pad, stream>>>(V, Gp, P, S, Q, numpart, itime, step_counter, d_stats, d_pbl_flag, d_pbl_ctr);
      maybe_new = false;
    }
    HIPCHK(hipEventRecord(ev.e[4], stream));
    {
      // work list = slots stably sorted by the 6-bit key of k_prep: class, cost bucket (non-PBL slots, keys 32 and 33, end
      // up behind the d_pbl_ctr[0] entries that are used)
      size_t need = sel_tmp_bytes;
      HIPCHK(rocprim::radix_sort_pairs(d_sel_tmp, need, d_pbl_flag, d_pbl_flag2, d_iota, d_pbl_list, (size_t)numpart, 0u, 6u, stream));
      k_list_counts<<<1, 64, 0, stream>>>(d_pbl_flag2, numpart, d_pbl_ctr, d_stats);
    }
    const int fin_grid = std::min(nb, 8 * 256 * 4);
    HIPCHK(hipEventRecord(ev.e[1], stream));
    for (size_t j = 0; j < slice_caps.size(); j++) {
      // launch j works through the particles launch j-1 suspended (k_pbl_loop); a launch whose list is empty ends at once;
      // the last launch suspends nothing
      const bool last = j + 1 == slice_caps.size();
      const unsigned int *list = j == 0 ? d_pbl_list : d_surv[(j - 1) & 1];
      loop_kernel()<<<pbl_grid, kBlock, loop_smem_bytes(), stream>>>(V, P, Q, itime, step_counter, d_stats, list, d_pbl_ctr + kCtrBase + kCtrStride * j, last ? 0 : slice_caps[j],
                                                                     last ? 0 : drain_lanes(), d_surv[j & 1]);
    }
    HIPCHK(hipEventRecord(ev.e[2], stream));
    {
      const bool polar = cfg.nglobal || cfg.sglobal, nest = V.numbnests > 0;
      typedef void (*fin_fn)(View<R>, GridP<R>, Parts<R>, PblRec<R>, int, unsigned int, Stats *, const unsigned char *, long long, const unsigned int *, const unsigned int *);
      const fin_fn f = (fin_fn)step_kernel_finish((int)sizeof(R), cfg.drydep != 0, polar, nest);
      f<<<fin_grid, kBlock, 0, stream>>>(V, Gp, P, Q, itime, step_counter, d_stats, d_pbl_flag, numpart,
                                         V.pbl_cost_buckets ? (const unsigned int *)nullptr : d_pbl_list, d_pbl_ctr);
    }
    HIPCHK(hipEventRecord(ev.e[3], stream));
    HIPCHK(hipGetLastError());
    V.w3t0 = nullptr; V.w3t1 = nullptr; V.r2t0 = nullptr;      // the blended packs belong to this step's itime only
    if (opt.verbose > 1) {   // the lists of the launches (a synchronisation per step: diagnostics only)
      unsigned int hc[kCtrWords];
      HIPCHK(hipMemcpyAsync(hc, d_pbl_ctr, sizeof(hc), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      fprintf(stderr, "[fpx] step %u: Langevin lists (per class)", step_counter);
      for (size_t j = 0; j < slice_caps.size(); j++) {
        const unsigned int *m = hc + kCtrBase + kCtrStride * j;
        fprintf(stderr, " | %u %u %u %u", m[0], m[1], m[2], m[3]);
      }
      fprintf(stderr, "\n");
    }
    step_counter++;
    if (async) return 0;
    Stats hs;
    HIPCHK(hipMemcpyAsync(&hs, d_stats, sizeof(Stats), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (out) {
      fill_stats(out, hs, snap);
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, ev.e[0], ev.e[3]));
      out->kernel_ms = ms;
    }
    snap = hs;
    return 0;
  }

  // device counters accumulate over steps; `snap`/`base` are host snapshots
  Stats snap{}, base{};
  static void fill_stats(fpx_step_stats *out, const Stats &a, const Stats &b) {
    out->n_due = (int64_t)(a.n_due - b.n_due); out->n_initialized = (int64_t)(a.n_init - b.n_init);
    out->n_left_domain = (int64_t)(a.n_left - b.n_left); out->n_min_mass = (int64_t)(a.n_minmass - b.n_minmass);
    out->n_max_age = (int64_t)(a.n_maxage - b.n_maxage); out->nan_count = (int64_t)(a.nan_count - b.nan_count);
    out->nan_count2 = (int64_t)(a.nan_count2 - b.nan_count2); out->n_bad_position = (int64_t)(a.n_badpos - b.n_badpos);
  }
  int lane_stats(uint64_t *out, int n, int reset) override {
    Stats hs;
    HIPCHK(hipMemcpyAsync(&hs, d_stats, sizeof(Stats), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    for (int i = 0; i < n && i < 32; i++) out[i] = hs.lanes[i / 2][i % 2];
    if (reset) HIPCHK(hipMemsetAsync((char *)d_stats + offsetof(Stats, lanes), 0, sizeof(hs.lanes), stream));
    return 0;
  }
  int set_option(const char *name, const char *value) override {
    const std::string n(name), v(value);
    char *end = nullptr;
    const long iv = strtol(value, &end, 10);
    const bool is_int = end != value && *end == 0;
    auto need_int = [&](long lo) -> bool { return is_int && iv >= lo; };
    if (n == "verbose") { if (!need_int(0)) goto bad; opt.verbose = (int)iv; return 0; }
    if (n == "pbl_blocks_per_cu") { if (!need_int(0)) goto bad; opt.pbl_blocks_per_cu = (int)iv; pbl_grid = 0; return 0; }
    if (n == "prep_lds_pad") { if (!need_int(0) || iv > 160 * 1024) goto bad; opt.prep_lds_pad = (int)iv; return 0; }
    if (n == "vt_unfused") { if (!need_int(0)) goto bad; opt.vt_unfused = iv != 0; return 0; }
    if (n == "conv_scratch_mb") { if (!need_int(0)) goto bad; opt.conv_scratch_mb = iv; return 0; }
    if (n == "conv_one_lane") { if (!need_int(0)) goto bad; opt.conv_one_lane = iv != 0; return 0; }
    if (n == "conv_no_walk") { if (!need_int(0)) goto bad; opt.conv_no_walk = iv != 0; return 0; }
    if (n == "conv_rows_plain") { if (!need_int(0)) goto bad; opt.conv_rows_plain = iv != 0; return 0; }
    if (n == "pbl_cost_buckets") { if (!is_int || iv < -1 || iv > 3) goto bad; opt.pbl_cost_buckets = (int)iv; return 0; }   // -1 automatic, 0 none, 1: four buckets, 2: two, 3: eight
    if (n == "prep_init_always") { if (!need_int(0)) goto bad; opt.prep_init_always = iv != 0; return 0; }
    if (n == "pbl_drain_lanes") { if (!is_int || iv < -1 || iv > 64) goto bad; opt.pbl_drain_lanes = (int)iv; return 0; }
    if (n == "permute") {
      if (v == "auto") opt.permute = 0; else if (v == "direct") opt.permute = 1; else if (v == "staged") opt.permute = 2; else goto bad;
      return 0;
    }
    if (n == "pbl_slices") {   // "48,96,0": pass budgets of the successive launches; "" = back to the configuration's
      std::vector<int> caps;
      const char *q = value;
      while (*q) {
        char *e2 = nullptr;
        const long c = strtol(q, &e2, 10);
        if (e2 == q || c < 0 || c > 1000000) goto bad;
        caps.push_back((int)c);
        q = e2;
        if (*q == ',') q++; else if (*q) goto bad;
      }
      if ((int)caps.size() > kMaxSlices - 1) goto bad;
      opt.pbl_slices = caps;
      slice_caps.clear();
      pbl_grid = 0;      // another instance of the kernel may run the next step
      return 0;
    }
  bad:
    return fail(FPX_ERR_ARG, "fpx_set_option: unknown option or malformed value: " + n + " = " + v);
  }
  int get_info(const char *name, int64_t *value) override {
    const std::string n(name);
    if (n == "time_blended_packs") *value = blend_on() ? 1 : 0;
    else if (n == "blended_steps") *value = (int64_t)blended_steps;
    else if (n == "pbl_launches_per_step") {
      if (!slice_caps.empty()) *value = (int64_t)slice_caps.size();
      else if (!opt.pbl_slices.empty()) *value = (int64_t)opt.pbl_slices.size() + (opt.pbl_slices.back() != 0 ? 1 : 0);
      else if (cfg.pbl_slice_passes < 0) *value = 1;
      else if (cfg.pbl_slice_passes > 0) *value = kMaxSlices;
      else { const int d[] = {FPX_SLICE_SCHEDULE}; *value = (int64_t)(sizeof(d) / sizeof(d[0])); }
    } else if (n == "pbl_grid") *value = pbl_grid;
    else if (n == "pbl_blocks_per_cu") *value = pbl_per_cu;
    else return fail(FPX_ERR_ARG, "fpx_get_info: unknown name: " + n);
    return 0;
  }
  int counters(fpx_step_stats *out, int reset) override {
    Stats hs;
    HIPCHK(hipMemcpyAsync(&hs, d_stats, sizeof(Stats), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (out) { memset(out, 0, sizeof(*out)); fill_stats(out, hs, base); }
    snap = hs;
    if (reset) base = hs;
    return 0;
  }

  // the Langevin kernel specialised for the run's switches (gases: LEAN) or the general one
  typedef void (*loop_fn)(View<R>, Parts<R>, PblRec<R>, int, unsigned int, Stats *, const unsigned int *, unsigned int *, int, int, unsigned int *);
  // The kernel instance of this engine: by the run's switches, and by whether a step is one launch or several (time slices);
  // the gas kernels -- LEAN instances, see loop_table -- leave the aerosol slots of the stash out of the block's LDS.
  bool loop_sliced() const {
    if (!slice_caps.empty()) return slice_caps.size() > 1;
    if (!opt.pbl_slices.empty()) return opt.pbl_slices.size() > 1 || opt.pbl_slices.back() != 0;
    if (cfg.pbl_slice_passes != 0) return cfg.pbl_slice_passes > 0;
    const int d[] = {FPX_SLICE_SCHEDULE};
    return sizeof(d) / sizeof(d[0]) > 1 || d[0] != 0;
  }
  loop_fn loop_kernel(bool *is_lean = nullptr) const {
    bool lean = false;
    const loop_fn f = (loop_fn)step_kernel_loop((int)sizeof(R), !cfg.drydep && !cfg.lsettling, cfg.turboff ? -1 : cfg.turbswitch, cfg.cblflag, cfg.rng_mode, loop_sliced(), &lean);
    if (is_lean) *is_lean = lean;
    return f;
  }
  size_t loop_smem_bytes() const {
    bool lean = false;
    (void)loop_kernel(&lean);
    return sizeof(R) * ((size_t)stash_slots<R>(lean, loop_sliced()) * kStashStride + (size_t)cfg.nz);
  }

  int sync() override {
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  int harvest_events() {
    HIPCHK(hipStreamSynchronize(stream));
    for (size_t i = 0; i < ev_used; i++) {
      float t = 0;
      HIPCHK(hipEventElapsedTime(&t, ev_pool[i].e[0], ev_pool[i].e[3]));
      acc_ms += t;
      for (int k = 0; k < 3; k++) {
        HIPCHK(hipEventElapsedTime(&t, ev_pool[i].e[k], ev_pool[i].e[k + 1]));
        acc_part_ms[k] += t;
      }
      HIPCHK(hipEventElapsedTime(&t, ev_pool[i].e[0], ev_pool[i].e[4]));
      acc_part_ms[3] += t;
      acc_launches++;
    }
    ev_used = 0;
    return 0;
  }
  int kernel_time(double *ms, long long *launches, int reset, double *parts) override {
    int rc = harvest_events();
    if (rc) return rc;
    if (ms) *ms = acc_ms;
    if (launches) *launches = acc_launches;
    if (parts) for (int k = 0; k < 4; k++) parts[k] = acc_part_ms[k];
    if (reset) { acc_ms = 0; acc_launches = 0; acc_part_ms[0] = acc_part_ms[1] = acc_part_ms[2] = acc_part_ms[3] = 0; }
    return 0;
  }

  SortRecord<R> *d_sort_rec = nullptr;
  size_t sort_rec_cap = 0;
  unsigned long long *d_disorder = nullptr;
  bool last_permute_staged = false;
  int alloc_parts(Parts<R> &Q) {
    const size_t cap = (size_t)P.cap;
    int rc;
    memset(&Q, 0, sizeof(Q));
    Q.cap = P.cap;
    if ((rc = dalloc(&Q.xt, cap))) return rc;
    if ((rc = dalloc(&Q.yt, cap))) return rc;
    R **rs[] = {&Q.zt, &Q.up, &Q.vp, &Q.wp, &Q.us, &Q.vs, &Q.ws};
    for (auto q : rs) if ((rc = dalloc(q, cap))) return rc;
    int **is[] = {&Q.idt, &Q.itra1, &Q.itramem, &Q.npoint, &Q.nclass, &Q.itrasplit};
    for (auto q : is) if ((rc = dalloc(q, cap))) return rc;
    if ((rc = dalloc(&Q.cbt, cap))) return rc;
    if ((rc = dalloc(&Q.xmass1, cap * cfg.nspec))) return rc;
    if (P.xscav) {
      if ((rc = dalloc(&Q.xscav, cap * cfg.nspec))) return rc;
      k_fill<R><<<(int)((cap * cfg.nspec + kBlock - 1) / kBlock), kBlock, 0, stream>>>(Q.xscav, (R)-1, 0, (long long)(cap * cfg.nspec), nullptr);
    }
    if ((rc = dalloc(&Q.pid, cap))) return rc;
    {   // zeroed like the first set (storage spaces behind numpart keep their contents across the ping-pong)
      R *rz[] = {Q.zt, Q.up, Q.vp, Q.wp, Q.us, Q.vs, Q.ws};
      for (R *q : rz) HIPCHK(hipMemsetAsync(q, 0, cap * sizeof(R), stream));
      HIPCHK(hipMemsetAsync(Q.xt, 0, cap * sizeof(double), stream));
      HIPCHK(hipMemsetAsync(Q.yt, 0, cap * sizeof(double), stream));
      HIPCHK(hipMemsetAsync(Q.xmass1, 0, cap * cfg.nspec * sizeof(R), stream));
      int *iz[] = {Q.idt, Q.itramem, Q.npoint, Q.nclass, Q.itrasplit};
      for (int *q : iz) HIPCHK(hipMemsetAsync(q, 0, cap * sizeof(int), stream));
      HIPCHK(hipMemsetAsync(Q.cbt, 0, cap * sizeof(short), stream));
    }
    return 0;
  }

  // The locality sort permutes the storage spaces 1..numpart among themselves; an operation that SHRINKS numpart (the
  // sender of a redistribution, fpx_set_numpart) must first put every particle back into the space of its number, or the
  // spaces beyond the new numpart would still hold live particles.
  int restore_particle_order() {
    if (!slot_of_pid) return 0;
    return sort_impl(true);
  }
  int sort_particles() override {
    if (!height_set) return fail(FPX_ERR_STATE, "sort_particles: set_height first");
    return sort_impl(false);
  }
  int sort_impl(bool by_pid) {
    const long long n = numpart;
    if (n < 2) { if (by_pid) slot_of_pid = nullptr; return 0; }
    int rc;
    if (!have_p2) {
      if ((rc = alloc_parts(P2))) return rc;
      const size_t cap = (size_t)P.cap;
      if ((rc = dalloc(&d_keys, cap))) return rc;
      if ((rc = dalloc(&d_keys2, cap))) return rc;
      if ((rc = dalloc(&d_vals, cap))) return rc;
      if ((rc = dalloc(&d_vals2, cap))) return rc;
      if ((rc = dalloc(&d_slot_of_pid, cap))) return rc;
      if ((rc = dalloc(&d_disorder, (size_t)1))) return rc;
      have_p2 = true;
    }
    const unsigned long long nkeys = (unsigned long long)cfg.nx * cfg.ny * cfg.nz + 1ull;
    if (nkeys > 0xFFFFFFFFull) return fail(FPX_ERR_UNSUPPORTED, "sort_particles: grid too large for 32-bit keys");
    unsigned int bits = 1;
    while ((1ull << bits) < (by_pid ? (unsigned long long)n : nkeys) + 1ull) bits++;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    if (by_pid) k_sort_keys_pid<<<nb, kBlock, 0, stream>>>(P.pid, n, d_keys, d_vals);
    else k_sort_keys<R><<<nb, kBlock, 0, stream>>>(V, P, n, d_keys, d_vals, (unsigned int)(nkeys - 1ull));
    HIPCHK(hipGetLastError());
    size_t need = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, need, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0u, bits, stream));
    if (need > sort_tmp_bytes) {
      if (d_sort_tmp) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(d_sort_tmp)); d_sort_tmp = nullptr; }
      HIPCHK(hipMalloc(&d_sort_tmp, need));
      sort_tmp_bytes = need;
    }
    HIPCHK(rocprim::radix_sort_pairs(d_sort_tmp, need, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0u, bits, stream));
    // which gather: by the locality of the permutation (fpx_set_option "permute" = direct|staged overrides, for tests)
    bool staged = false;
    {
      if (opt.permute == 2) staged = true;
      else if (opt.permute != 1 && n >= (1 << 16)) {
        HIPCHK(hipMemsetAsync(d_disorder, 0, sizeof(unsigned long long), stream));
        k_perm_disorder<<<std::min(nb, 4096), kBlock, 0, stream>>>(d_vals2, n, 1u << 14, d_disorder);
        unsigned long long far = 0;
        HIPCHK(hipMemcpyAsync(&far, d_disorder, sizeof(far), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        staged = far * 2ull > (unsigned long long)n;
      }
    }
    last_permute_staged = staged;
    if (staged) {
      if ((size_t)n > sort_rec_cap) {
        if (d_sort_rec) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(d_sort_rec)); d_sort_rec = nullptr; sort_rec_cap = 0; }
        HIPCHK(hipMalloc(&d_sort_rec, (size_t)n * sizeof(SortRecord<R>)));
        sort_rec_cap = (size_t)n;
      }
      k_permute_pack<R><<<nb, kBlock, 0, stream>>>(P, d_sort_rec, n);
      k_permute_unpack<R><<<nb, kBlock, 0, stream>>>(d_sort_rec, P, P2, d_vals2, n, cfg.nspec);
      // 12.8 GB at 1e8 particles, needed once in a run: give it back
      HIPCHK(hipStreamSynchronize(stream));
      HIPCHK(hipFree(d_sort_rec)); d_sort_rec = nullptr; sort_rec_cap = 0;
    } else {
      const int tiles_per_xcd = (nb + 7) / 8;
      k_permute<R, 0><<<8 * tiles_per_xcd, kBlock, 0, stream>>>(P, P2, d_vals2, n, cfg.nspec, tiles_per_xcd);
      k_permute<R, 1><<<8 * tiles_per_xcd, kBlock, 0, stream>>>(P, P2, d_vals2, n, cfg.nspec, tiles_per_xcd);
      k_permute<R, 2><<<8 * tiles_per_xcd, kBlock, 0, stream>>>(P, P2, d_vals2, n, cfg.nspec, tiles_per_xcd);
      k_permute<R, 3><<<8 * tiles_per_xcd, kBlock, 0, stream>>>(P, P2, d_vals2, n, cfg.nspec, tiles_per_xcd);
    }
    HIPCHK(hipGetLastError());
    // slots >= n keep their (dead) contents in both sets; swap roles
    std::swap(P, P2);
    if (n < P.cap) {
      // the untouched tail of the new front set must also be dead and identity-numbered
      const long long rest = P.cap - n;
      const int nbr = (int)((rest + kBlock - 1) / kBlock);
      k_fill<int><<<nbr, kBlock, 0, stream>>>(P.itra1, kDead, n, rest, nullptr);
      k_iota_pid<<<nbr, kBlock, 0, stream>>>(P.pid, n, rest);
      HIPCHK(hipGetLastError());
    }
    slot_of_pid = by_pid ? nullptr : d_slot_of_pid;      // back in particle-number order: the identity again
    slot_map_dirty = !by_pid;
    return 0;
  }

  // ---- output grids -----------------------------------------------------------
  int outgrid_init(const fpx_outgrid *g, const void *outheight) override {
    if (!g || !outheight) return fail(FPX_ERR_ARG, "outgrid_init: null argument");
    if (g->struct_bytes != (int32_t)sizeof(fpx_outgrid)) return fail(FPX_ERR_ARG, "outgrid_init: fpx_outgrid size mismatch (ABI)");
    if (g->numxgrid < 1 || g->numygrid < 1 || g->numzgrid < 1 || g->numzgrid > kMaxNz) return fail(FPX_ERR_ARG, "outgrid_init: bad grid extents");
    if (g->maxpointspec_act < 1 || g->nclassunc < 1 || g->nageclass < 1 || g->nageclass > kMaxAge) return fail(FPX_ERR_ARG, "outgrid_init: bad maxpointspec_act/nclassunc/nageclass");
    if (Gp.on) return fail(FPX_ERR_STATE, "outgrid_init: already initialised");
    Gp.numxgrid = g->numxgrid; Gp.numygrid = g->numygrid; Gp.numzgrid = g->numzgrid;
    Gp.maxspec = cfg.maxspec; Gp.maxpointspec_act = g->maxpointspec_act; Gp.nclassunc = g->nclassunc; Gp.nageclass = g->nageclass;
    for (int i = 0; i < kMaxAge; i++) Gp.lage[i] = i < g->nageclass ? g->lage[i] : 0x7fffffff;
    Gp.dxout = (R)g->dxout; Gp.dyout = (R)g->dyout; Gp.xoutshift = (R)g->xoutshift; Gp.youtshift = (R)g->youtshift;
    Gp.ind_samp = g->ind_samp; Gp.ioutputforeachrelease = g->ioutputforeachrelease; Gp.lusekerneloutput = g->lusekerneloutput;
    Gp.loutnext = 0; Gp.loutstep = 0;
    n_grid2 = (size_t)g->numxgrid * g->numygrid * cfg.maxspec * g->maxpointspec_act * g->nclassunc * g->nageclass;
    n_grid3 = n_grid2 * g->numzgrid;
    int rc;
    R *oh;
    if ((rc = dalloc(&oh, g->numzgrid))) return rc;
    std::vector<R> tmp(g->numzgrid);
    for (int k = 0; k < g->numzgrid; k++) tmp[k] = cfg.host_real_bytes == 4 ? (R)((const float *)outheight)[k] : (R)((const double *)outheight)[k];
    HIPCHK(hipMemcpyAsync(oh, tmp.data(), g->numzgrid * sizeof(R), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    Gp.outheight = oh;
    if ((rc = dalloc(&Gp.gridunc, n_grid3))) return rc;
    if ((rc = dalloc(&Gp.drygridunc, n_grid2))) return rc;
    if ((rc = dalloc(&Gp.wetgridunc, n_grid2))) return rc;
    HIPCHK(hipMemsetAsync(Gp.wetgridunc, 0, n_grid2 * sizeof(float), stream));
    HIPCHK(hipMemsetAsync(Gp.gridunc, 0, n_grid3 * sizeof(R), stream));
    HIPCHK(hipMemsetAsync(Gp.drygridunc, 0, n_grid2 * sizeof(float), stream));
    HIPCHK(hipStreamSynchronize(stream));
    Gp.nested = 0; Gp.numreceptor = 0;
    Gp.on = 1;
    return 0;
  }
  // nested output grid: readoutgrid_nest.f90, outgrid_init_nest.f90 (same levels, classes, species as the mother grid)
  int outgrid_nest_init(const fpx_outgrid_nest *g) override {
    if (!g || g->struct_bytes != (int32_t)sizeof(fpx_outgrid_nest)) return fail(FPX_ERR_ARG, "outgrid_nest_init: null or fpx_outgrid_nest size mismatch (ABI)");
    if (!Gp.on) return fail(FPX_ERR_STATE, "outgrid_nest_init: fpx_outgrid_init first");
    if (Gp.nested) return fail(FPX_ERR_STATE, "outgrid_nest_init: already initialised");
    if (g->numxgridn < 1 || g->numygridn < 1 || !(g->dxoutn > 0) || !(g->dyoutn > 0)) return fail(FPX_ERR_ARG, "outgrid_nest_init: bad grid");
    Gp.numxgridn = g->numxgridn; Gp.numygridn = g->numygridn;
    Gp.dxoutn = (R)g->dxoutn; Gp.dyoutn = (R)g->dyoutn; Gp.xoutshiftn = (R)g->xoutshiftn; Gp.youtshiftn = (R)g->youtshiftn;
    n_grid2n = (size_t)g->numxgridn * g->numygridn * Gp.maxspec * Gp.maxpointspec_act * Gp.nclassunc * Gp.nageclass;
    n_grid3n = n_grid2n * Gp.numzgrid;
    int rc;
    if ((rc = dalloc(&Gp.griduncn, n_grid3n))) return rc;
    if ((rc = dalloc(&Gp.drygriduncn, n_grid2n))) return rc;
    if ((rc = dalloc(&Gp.wetgriduncn, n_grid2n))) return rc;
    HIPCHK(hipMemsetAsync(Gp.griduncn, 0, n_grid3n * sizeof(R), stream));
    HIPCHK(hipMemsetAsync(Gp.drygriduncn, 0, n_grid2n * sizeof(float), stream));
    HIPCHK(hipMemsetAsync(Gp.wetgriduncn, 0, n_grid2n * sizeof(float), stream));
    HIPCHK(hipStreamSynchronize(stream));
    Gp.nested = 1;
    return 0;
  }
  int download_real(void *host, const R *dev, size_t n) {   // device R -> host real kind
    if (!host) return 0;
    if ((size_t)cfg.host_real_bytes == sizeof(R)) {
      HIPCHK(hipMemcpyAsync(host, dev, n * sizeof(R), hipMemcpyDeviceToHost, stream));
      return 0;
    }
    int rc = ensure_staging(n * cfg.host_real_bytes);
    if (rc) return rc;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    k_convert<R><<<nb, kBlock, 0, stream>>>(dev, cfg.host_real_bytes == 8 ? (double *)staging : nullptr,
                                            cfg.host_real_bytes == 4 ? (float *)staging : nullptr, (long long)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host, staging, n * cfg.host_real_bytes, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));   // staging is reused
    return 0;
  }
  int get_grids_nest(void *griduncn, void *drygriduncn, void *wetgriduncn, int allreduce, int clear) override {
    if (!Gp.nested) return fail(FPX_ERR_STATE, "get_grids_nest: fpx_outgrid_nest_init first");
    const R *g3 = Gp.griduncn;
    const float *gd = Gp.drygriduncn, *gw = Gp.wetgriduncn;
    if (allreduce && comm_ranks > 1) {
      // mpi_mod.f90:2543-2569 (mpif_tm_reduce_grid_nest): into the receive buffers, the partial sums stay
      int rc;
      if ((rc = reduce_into(Gp.griduncn, &griduncn0, n_grid3n, "get_grids_nest")) || (rc = reduce_into(Gp.drygriduncn, &drygriduncn0, n_grid2n, "get_grids_nest")) ||
          (rc = reduce_into(Gp.wetgriduncn, &wetgriduncn0, n_grid2n, "get_grids_nest"))) return rc;
      red_valid[RG_GRIDN] = red_valid[RG_DRYN] = red_valid[RG_WETN] = true;
      g3 = griduncn0; gd = drygriduncn0; gw = wetgriduncn0;
    }
    int rc = download_real(griduncn, g3, n_grid3n);
    if (rc) return rc;
    if (drygriduncn) HIPCHK(hipMemcpyAsync(drygriduncn, gd, n_grid2n * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (wetgriduncn) HIPCHK(hipMemcpyAsync(wetgriduncn, gw, n_grid2n * sizeof(float), hipMemcpyDeviceToHost, stream));
    // concoutput_nest.f90 zeroes griduncn only; drygriduncn / wetgriduncn accumulate over the run
    if (clear) HIPCHK(hipMemsetAsync(Gp.griduncn, 0, n_grid3n * sizeof(R), stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  // receptor points: xreceptor, yreceptor (grid coordinates), receptorarea, readreceptors.f90:88-92
  int receptors_init(int n, const void *x, const void *y, const void *area) override {
    if (!Gp.on) return fail(FPX_ERR_STATE, "receptors_init: fpx_outgrid_init first");
    if (n < 1 || n > 1000 || !x || !y || !area) return fail(FPX_ERR_ARG, "receptors_init: bad argument");
    if (Gp.numreceptor) return fail(FPX_ERR_STATE, "receptors_init: already initialised");
    std::vector<R> tmp(3 * (size_t)n);
    const void *src[3] = {x, y, area};
    for (int k = 0; k < 3; k++)
      for (int i = 0; i < n; i++) tmp[(size_t)k * n + i] = cfg.host_real_bytes == 4 ? (R)((const float *)src[k])[i] : (R)((const double *)src[k])[i];
    int rc;
    R *d;
    if ((rc = dalloc(&d, 3 * (size_t)n))) return rc;
    HIPCHK(hipMemcpyAsync(d, tmp.data(), tmp.size() * sizeof(R), hipMemcpyHostToDevice, stream));
    if ((rc = dalloc(&Gp.creceptor, (size_t)n * cfg.maxspec))) return rc;
    HIPCHK(hipMemsetAsync(Gp.creceptor, 0, (size_t)n * cfg.maxspec * sizeof(R), stream));
    HIPCHK(hipStreamSynchronize(stream));
    Gp.receptor = d;
    Gp.numreceptor = n;
    return 0;
  }
  // creceptor(ld, maxspec) of the host (com_mod.f90:660): rows 1..numreceptor, columns 1..nspec are written
  int get_receptors(void *creceptor, int ld, int allreduce, int clear) override {
    if (!Gp.numreceptor) return fail(FPX_ERR_STATE, "get_receptors: fpx_receptors_init first");
    if (creceptor && ld < Gp.numreceptor) return fail(FPX_ERR_ARG, "get_receptors: leading dimension smaller than numreceptor");
    const size_t n = (size_t)Gp.numreceptor * cfg.nspec;
    const R *src = Gp.creceptor;
    if (allreduce && comm_ranks > 1) {
      int rc = reduce_into(Gp.creceptor, &creceptor0, n, "get_receptors");   // mpi_mod.f90:2480-2484, into creceptor0
      if (rc) return rc;
      red_valid[RG_REC] = true;
      src = creceptor0;
    }
    if (creceptor) {
      std::vector<R> tmp(n);
      HIPCHK(hipMemcpyAsync(tmp.data(), src, n * sizeof(R), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      for (int ks = 0; ks < cfg.nspec; ks++)
        for (int i = 0; i < Gp.numreceptor; i++) {
          const R v = tmp[(size_t)ks * Gp.numreceptor + i];
          if (cfg.host_real_bytes == 4) ((float *)creceptor)[(size_t)ks * ld + i] = (float)v; else ((double *)creceptor)[(size_t)ks * ld + i] = (double)v;
        }
    }
    if (clear) HIPCHK(hipMemsetAsync(Gp.creceptor, 0, n * sizeof(R), stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }
  int set_output_times(int loutnext, int loutstep) override {
    Gp.loutnext = loutnext; Gp.loutstep = loutstep;
    return 0;
  }
  int conccalc(int itime, double weight) override {
    if (!Gp.on) return fail(FPX_ERR_STATE, "conccalc: fpx_outgrid_init first");
    if (!height_set || (Gp.ind_samp == -1 && (!slot_loaded[0] || !slot_loaded[1]))) return fail(FPX_ERR_STATE, "conccalc: height / fields not set");
    if (numpart == 0) return 0;
    const int nb = (int)((numpart + kBlock - 1) / kBlock);
    red_valid[RG_GRID] = red_valid[RG_GRIDN] = red_valid[RG_REC] = false;   // the partial sums move on: earlier reductions are stale
    k_conccalc<R><<<nb, kBlock, 0, stream>>>(V, Gp, P, numpart, itime, (R)weight);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // sum over the ranks of `send` into `recv` (device pointers, distinct buffers), on the handle's stream
  template <typename T>
  int reduce_into(const T *send, T **recv, size_t n, const char *who) {
    if (!comm && !host_allreduce) return fail(FPX_ERR_STATE, std::string(who) + ": allreduce requested without fpx_comm_init / fpx_comm_init_host");
    int rc;
    if (!*recv && (rc = dalloc(recv, n))) return rc;
    if (comm) {
      ncclResult_t r = ncclAllReduce(send, *recv, n, sizeof(T) == 8 ? ncclDouble : ncclFloat, ncclSum, comm, stream);
      if (r != ncclSuccess) return fail(FPX_ERR_DEVICE, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
      return 0;
    }
    const size_t bytes = n * sizeof(T);
    if (2 * bytes > red_pin_bytes) {
      if (red_pin) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipHostFree(red_pin)); red_pin = nullptr; red_pin_bytes = 0; }
      HIPCHK(hipHostMalloc(&red_pin, 2 * bytes));
      red_pin_bytes = 2 * bytes;
    }
    HIPCHK(hipMemcpyAsync(red_pin, send, bytes, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (host_allreduce(host_allreduce_user, red_pin, (char *)red_pin + bytes, (int64_t)n, sizeof(T) == 8 ? 1 : 0) != 0)
      return fail(FPX_ERR_DEVICE, std::string(who) + ": the host's all-reduce callback failed");
    HIPCHK(hipMemcpyAsync(*recv, (char *)red_pin + bytes, bytes, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));    // the bounce buffer is reused by the next grid
    return 0;
  }
  // [live particles, numpart] of this rank and summed over the ranks: the reduction of numpart the reference's root does at
  // every output time (timemanager_mpi.f90:552-562) -- here with the count of the particles that are still alive next to it
  long long *d_count = nullptr;          // [0..1] local, [2..3] sums
  int count_particles(int64_t *local, int64_t *total, int allreduce) override {
    int rc;
    if (!d_count && (rc = dalloc(&d_count, 4))) return rc;
    HIPCHK(hipMemsetAsync(d_count, 0, 4 * sizeof(long long), stream));
    if (numpart > 0) {
      const int nb = (int)std::min<long long>((numpart + 255) / 256, 256 * 16);
      k_count_live<<<nb, 256, 0, stream>>>(P.itra1, numpart, (unsigned long long *)d_count);
      HIPCHK(hipGetLastError());
    }
    long long h[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, d_count, sizeof(long long), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    h[1] = numpart;
    h[2] = h[0]; h[3] = h[1];
    if (allreduce && comm_ranks > 1) {
      if (comm) {
        HIPCHK(hipMemcpyAsync(d_count, h, 2 * sizeof(long long), hipMemcpyHostToDevice, stream));
        ncclResult_t r = ncclAllReduce(d_count, d_count + 2, 2, ncclInt64, ncclSum, comm, stream);
        if (r != ncclSuccess) return fail(FPX_ERR_DEVICE, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
        HIPCHK(hipMemcpyAsync(h + 2, d_count + 2, 2 * sizeof(long long), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
      } else if (host_allreduce) {
        double snd[2] = {(double)h[0], (double)h[1]}, rcv[2] = {0, 0};   // exact below 2^53
        if (host_allreduce(host_allreduce_user, snd, rcv, 2, 1) != 0) return fail(FPX_ERR_DEVICE, "count_particles: the host's all-reduce callback failed");
        h[2] = (long long)rcv[0]; h[3] = (long long)rcv[1];
      } else {
        return fail(FPX_ERR_STATE, "count_particles: allreduce requested without fpx_comm_init / fpx_comm_init_host");
      }
    }
    if (local) { local[0] = h[0]; local[1] = h[1]; }
    if (total) { total[0] = h[2]; total[1] = h[3]; }
    return 0;
  }
  int comm_init_host(int nranks, int rank, fpx_allreduce_fn fn, void *user) override {
    if (nranks < 1 || rank < 0 || rank >= nranks || !fn) return fail(FPX_ERR_ARG, "comm_init_host: bad argument");
    if (comm || host_allreduce) return fail(FPX_ERR_STATE, "comm_init_host: communicator exists");
    host_allreduce = fn; host_allreduce_user = user;
    comm_ranks = nranks; comm_rank = rank;
    return 0;
  }
  int comm_init(const void *id, int nbytes, int nranks, int rank) override {
    if (!id || nbytes != (int)sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks) return fail(FPX_ERR_ARG, "comm_init: bad argument");
    if (comm || host_allreduce) return fail(FPX_ERR_STATE, "comm_init: communicator exists");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    HIPCHK(hipSetDevice(cfg.device));
    ncclResult_t r = ncclCommInitRank(&comm, nranks, uid, rank);
    if (r != ncclSuccess) return fail(FPX_ERR_DEVICE, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    comm_ranks = nranks; comm_rank = rank;
    return 0;
  }
  int get_grids(void *gridunc, void *drygridunc, int allreduce, int clear) override {
    if (!Gp.on) return fail(FPX_ERR_STATE, "get_grids: fpx_outgrid_init first");
    const R *g3 = Gp.gridunc;
    const float *gd = Gp.drygridunc;
    if (allreduce && comm_ranks > 1) {
      // the one collective of the path (mpi_mod.f90:2471-2492): the sums land in gridunc0 / drygridunc0, the
      // partial sums stay where they are -- drygridunc keeps accumulating and is reduced again at the next output
      int rc;
      if ((rc = reduce_into(Gp.gridunc, &gridunc0, n_grid3, "get_grids")) || (rc = reduce_into(Gp.drygridunc, &drygridunc0, n_grid2, "get_grids"))) return rc;
      red_valid[RG_GRID] = red_valid[RG_DRY] = true;
      g3 = gridunc0; gd = drygridunc0;
    }
    int rc = download_real(gridunc, g3, n_grid3);
    if (rc) return rc;
    if (drygridunc) HIPCHK(hipMemcpyAsync(drygridunc, gd, n_grid2 * sizeof(float), hipMemcpyDeviceToHost, stream));
    // concoutput.f90:719-720 zeroes gridunc (and creceptor: fpx_get_receptors) only; the deposition grids are
    // cumulative over the run (zeroed once, outgrid_init.f90:317-318)
    if (clear) HIPCHK(hipMemsetAsync(Gp.gridunc, 0, n_grid3 * sizeof(R), stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  // ---- nested grids --------------------------------------------------------------
  int nest_nxmaxn = 0, nest_nymaxn = 0;
  bool nest_loaded[kMaxNests][2] = {};
  NestDesc<R> h_nest[kMaxNests] = {};   // host copy of the device table V.nest
  int nests_init(const fpx_nests *n) override {
    if (!n || n->struct_bytes != (int32_t)sizeof(fpx_nests)) return fail(FPX_ERR_ARG, "nests_init: null or fpx_nests size mismatch (ABI)");
    if (n->numbnests < 1 || n->numbnests > kMaxNests) return fail(FPX_ERR_ARG, "nests_init: numbnests out of range");
    if (V.numbnests) return fail(FPX_ERR_STATE, "nests_init: already initialised");
    if (cfg.interpolhmix) return fail(FPX_ERR_UNSUPPORTED, "nests_init: interpolhmix with nested wind fields -- the reference reads the unset h1 for a particle inside a nest (advance.f90:254-266)");
    int rc;
    for (int l = 0; l < n->numbnests; l++) {
      if (n->nxn[l] < 2 || n->nyn[l] < 2 || n->nxn[l] > n->nxmaxn || n->nyn[l] > n->nymaxn) return fail(FPX_ERR_ARG, "nests_init: bad nest extents");
      if ((long long)n->nxn[l] * n->nyn[l] * cfg.nz > 0xFFFFFFF0ll) return fail(FPX_ERR_ARG, "nests_init: nest too large: nxn*nyn*nz must fit 32 bits");
      if (!(n->xln[l] >= 0 && n->yln[l] >= 0 && n->xrn[l] <= cfg.nx - 1 && n->yrn[l] <= cfg.ny - 1)) return fail(FPX_ERR_ARG, "nests_init: nest outside the mother grid (gridcheck_nests.f90:381)");
      NestDesc<R> &N = h_nest[l];
      N.nx = n->nxn[l]; N.ny = n->nyn[l];
      N.xl = (R)n->xln[l]; N.yl = (R)n->yln[l]; N.xr = (R)n->xrn[l]; N.yr = (R)n->yrn[l];
      N.xres = (R)n->xresoln[l]; N.yres = (R)n->yresoln[l];
      const size_t ncol = (size_t)n->nxn[l] * n->nyn[l], nlev = ncol * cfg.nz;
      R *p;
      if ((rc = dalloc(&p, nlev * 6))) return rc; N.w3 = p;
      if ((rc = dalloc(&p, nlev * 4))) return rc; N.r2 = p;
      if ((rc = dalloc(&p, ncol * 8))) return rc; N.sfc = p;
      if ((rc = dalloc(&p, ncol))) return rc; N.hcell = p;
      if ((rc = dalloc(&p, ncol))) return rc; N.tropo = p;
      if (cfg.drydep) { if ((rc = dalloc(&p, ncol * 2 * cfg.nspec))) return rc; N.vdep = p; }
    }
    nest_nxmaxn = n->nxmaxn; nest_nymaxn = n->nymaxn;
    NestDesc<R> *dn;
    if ((rc = dalloc(&dn, (size_t)n->numbnests))) return rc;
    HIPCHK(hipMemcpyAsync(dn, h_nest, sizeof(NestDesc<R>) * n->numbnests, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    V.nest = dn;
    V.numbnests = n->numbnests;
    return 0;
  }
  int upload_nest_fields(int nest, int slot, const fpx_fields *f) override {
    if (nest < 1 || nest > V.numbnests) return fail(FPX_ERR_ARG, "upload_nest_fields: nest out of range (fpx_nests_init first)");
    if (slot != 1 && slot != 2) return fail(FPX_ERR_ARG, "upload_nest_fields: slot must be 1 or 2");
    if (!f || !f->uu || !f->vv || !f->ww || !f->rho || !f->drhodz || !f->hmix || !f->ustar || !f->wstar || !f->oli || !f->tropopause)
      return fail(FPX_ERR_ARG, "upload_nest_fields: uun, vvn, wwn, rhon, drhodzn, hmixn, ustarn, wstarn, olin, tropopausen are required");
    if (cfg.drydep && !f->vdep) return fail(FPX_ERR_ARG, "upload_nest_fields: vdepn required with DRYDEP");
    const int l = nest - 1, s = slot - 1;
    g_nx = h_nest[l].nx; g_ny = h_nest[l].ny; g_nxmax = nest_nxmaxn; g_nymax = nest_nymaxn;
    int rc = 0;
    do {
      if ((rc = p3(f->uu, (R *)h_nest[l].w3, 6, s * 3 + 0))) break;
      if ((rc = p3(f->vv, (R *)h_nest[l].w3, 6, s * 3 + 1))) break;
      if ((rc = p3(f->ww, (R *)h_nest[l].w3, 6, s * 3 + 2))) break;
      if ((rc = p3(f->rho, (R *)h_nest[l].r2, 4, s * 2 + 0))) break;
      if ((rc = p3(f->drhodz, (R *)h_nest[l].r2, 4, s * 2 + 1))) break;
      if ((rc = p2(f->ustar, (R *)h_nest[l].sfc, 8, s * 4 + 0))) break;
      if ((rc = p2(f->wstar, (R *)h_nest[l].sfc, 8, s * 4 + 1))) break;
      if ((rc = p2(f->oli, (R *)h_nest[l].sfc, 8, s * 4 + 2))) break;
      if ((rc = p2(f->hmix, (R *)h_nest[l].sfc, 8, s * 4 + 3))) break;
      if (slot == 1 && (rc = p2(f->tropopause, (R *)h_nest[l].tropo, 1, 0))) break;   // tropopausen(nix,njy,1,1,ngrid), advance.f90:263
      if ((R *)h_nest[l].vdep) {
        const size_t plane = (size_t)nest_nxmaxn * nest_nymaxn * cfg.host_real_bytes;
        for (int ks = 0; ks < cfg.nspec && !rc; ks++) rc = p2((const char *)f->vdep + plane * ks, (R *)h_nest[l].vdep, 2 * cfg.nspec, s * cfg.nspec + ks);
      }
    } while (0);
    const int nxl = g_nx, nyl = g_ny;
    g_nx = cfg.nx; g_ny = cfg.ny; g_nxmax = cfg.nxmax; g_nymax = cfg.nymax;
    if (rc) return rc;
    int tot = nxl * nyl;
    k_hcell<R><<<(tot + kBlock - 1) / kBlock, kBlock, 0, stream>>>((R *)h_nest[l].sfc, (R *)h_nest[l].hcell, nxl, nyl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    nest_loaded[l][s] = true;
    return 0;
  }

  // ---- wet deposition -----------------------------------------------------------
  int wet_init(const fpx_wet_config *w) override {
    if (!w || w->struct_bytes != (int32_t)sizeof(fpx_wet_config)) return fail(FPX_ERR_ARG, "wet_init: null or fpx_wet_config size mismatch (ABI)");
    if (wet_on) return fail(FPX_ERR_STATE, "wet_init: already initialised");
    memset(&Wp, 0, sizeof(Wp));
    for (int i = 0; i < FPX_MAXSPEC; i++) {
      Wp.wetdepspec[i] = w->wetdepspec[i];
      Wp.weta_gas[i] = (R)w->weta_gas[i]; Wp.wetb_gas[i] = (R)w->wetb_gas[i];
      Wp.crain_aero[i] = (R)w->crain_aero[i]; Wp.csnow_aero[i] = (R)w->csnow_aero[i];
      Wp.ccn_aero[i] = (R)std::max(w->ccn_aero[i], 0.0);   // get_wetscav.f90:257-258
      Wp.in_aero[i] = (R)std::max(w->in_aero[i], 0.0);
      Wp.henry[i] = (R)w->henry[i];
    }
    Wp.readclouds = w->readclouds;
    const size_t ncol = (size_t)cfg.nx * cfg.ny, nlev = ncol * cfg.nz;
    int rc;
    R *p;
    signed char *c8;
    if ((rc = dalloc(&p, ncol * 6))) return rc; Wp.prec = p;
    if ((rc = dalloc(&p, ncol * 2))) return rc; Wp.ctwc = p;
    HIPCHK(hipMemsetAsync(p, 0, ncol * 2 * sizeof(R), stream));
    if ((rc = dalloc(&p, nlev * 2))) return rc; Wp.ttw = p;
    if ((rc = dalloc(&c8, nlev * 2))) return rc; Wp.clouds = c8;
    HIPCHK(hipStreamSynchronize(stream));
    wet_on = true;
    return 0;
  }
  int upload_wet_fields(int slot, const fpx_wet_fields *f) override {
    if (!wet_on) return fail(FPX_ERR_STATE, "upload_wet_fields: fpx_wet_init first");
    if (slot != 1 && slot != 2) return fail(FPX_ERR_ARG, "upload_wet_fields: slot must be 1 or 2");
    if (!f || !f->lsprec || !f->convprec || !f->tcc || !f->tt || !f->clouds) return fail(FPX_ERR_ARG, "upload_wet_fields: lsprec, convprec, tcc, tt, clouds are required");
    if (Wp.readclouds && !f->ctwc) return fail(FPX_ERR_ARG, "upload_wet_fields: ctwc required with readclouds");
    const int s = slot - 1;
    int rc;
    if ((rc = p2(f->lsprec, Wp.prec, 6, s * 3 + 0))) return rc;
    if ((rc = p2(f->convprec, Wp.prec, 6, s * 3 + 1))) return rc;
    if ((rc = p2(f->tcc, Wp.prec, 6, s * 3 + 2))) return rc;
    if (f->ctwc && (rc = p2(f->ctwc, Wp.ctwc, 2, s))) return rc;
    if ((rc = p3(f->tt, Wp.ttw, 2, s))) return rc;
    {   // integer(1) cloud classes: same x<->z transpose, element type int8
      const size_t n = (size_t)cfg.nxmax * cfg.nymax * cfg.nz;
      if ((rc = ensure_staging(n))) return rc;
      HIPCHK(hipMemcpyAsync(staging, f->clouds, n, hipMemcpyHostToDevice, stream));
      dim3 grid((cfg.nx + 31) / 32, (cfg.nz + 31) / 32, cfg.ny), block(32, 8);
      k_pack3<signed char, signed char><<<grid, block, 0, stream>>>((const signed char *)staging, (signed char *)Wp.clouds, cfg.nx, cfg.ny, cfg.nz,
                                                                    cfg.nxmax, cfg.nymax, 2, s);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(stream));
    }
    wet_slot[s] = true;
    return 0;
  }
  // lsprecn, convprecn, tccn, ctwcn, ttn, cloudsn of one nest and time slot (strides nxmaxn, nymaxn, nzmax)
  int upload_wet_nest_fields(int nest, int slot, const fpx_wet_fields *f, int readclouds_nest) override {
    if (!wet_on) return fail(FPX_ERR_STATE, "upload_wet_nest_fields: fpx_wet_init first");
    if (nest < 1 || nest > V.numbnests) return fail(FPX_ERR_ARG, "upload_wet_nest_fields: nest out of range (fpx_nests_init first)");
    if (slot != 1 && slot != 2) return fail(FPX_ERR_ARG, "upload_wet_nest_fields: slot must be 1 or 2");
    if (!f || !f->lsprec || !f->convprec || !f->tcc || !f->tt || !f->clouds) return fail(FPX_ERR_ARG, "upload_wet_nest_fields: lsprecn, convprecn, tccn, ttn, cloudsn are required");
    if (readclouds_nest && !f->ctwc) return fail(FPX_ERR_ARG, "upload_wet_nest_fields: ctwcn required with readclouds_nest");
    const int l = nest - 1, s = slot - 1;
    const int nxl = h_nest[l].nx, nyl = h_nest[l].ny;
    int rc;
    WetNest<R> &W = h_wnest[l];
    if (!W.prec) {
      const size_t ncol = (size_t)nxl * nyl;
      R *p; signed char *q;
      if ((rc = dalloc(&p, ncol * 6))) return rc; W.prec = p;
      if ((rc = dalloc(&p, ncol * 2))) return rc; W.ctwc = p;
      if ((rc = dalloc(&p, ncol * cfg.nz * 2))) return rc; W.ttw = p;
      if ((rc = dalloc(&q, ncol * cfg.nz * 2))) return rc; W.clouds = q;
      if (!d_wnest) { if ((rc = dalloc(&d_wnest, (size_t)kMaxNests))) return rc; }
    }
    W.readclouds = readclouds_nest ? 1 : 0;
    g_nx = nxl; g_ny = nyl; g_nxmax = nest_nxmaxn; g_nymax = nest_nymaxn;
    do {
      if ((rc = p2(f->lsprec, (R *)W.prec, 6, s * 3 + 0))) break;
      if ((rc = p2(f->convprec, (R *)W.prec, 6, s * 3 + 1))) break;
      if ((rc = p2(f->tcc, (R *)W.prec, 6, s * 3 + 2))) break;
      if (f->ctwc && (rc = p2(f->ctwc, (R *)W.ctwc, 2, s))) break;
      if ((rc = p3(f->tt, (R *)W.ttw, 2, s))) break;
    } while (0);
    g_nx = cfg.nx; g_ny = cfg.ny; g_nxmax = cfg.nxmax; g_nymax = cfg.nymax;
    if (rc) return rc;
    {
      const size_t n = (size_t)nest_nxmaxn * nest_nymaxn * cfg.nz;
      if ((rc = ensure_staging(n))) return rc;
      HIPCHK(hipMemcpyAsync(staging, f->clouds, n, hipMemcpyHostToDevice, stream));
      dim3 grid((nxl + 31) / 32, (cfg.nz + 31) / 32, nyl), block(32, 8);
      k_pack3<signed char, signed char><<<grid, block, 0, stream>>>((const signed char *)staging, (signed char *)W.clouds, nxl, nyl, cfg.nz,
                                                                    nest_nxmaxn, nest_nymaxn, 2, s);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(d_wnest, h_wnest, sizeof(h_wnest), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    Wp.nest = d_wnest;
    wet_nest_slot[l][s] = true;
    return 0;
  }
  int wetdepo(int itime, int ltsample, int loutnext) override {
    if (!wet_on || !wet_slot[0] || !wet_slot[1]) return fail(FPX_ERR_STATE, "wetdepo: fpx_wet_init and both slots of fpx_upload_wet_fields first");
    for (int l = 0; l < V.numbnests; l++)   // get_wetscav.f90:126-128 reads the nest's own fields for a particle inside a nest
      if (!wet_nest_slot[l][0] || !wet_nest_slot[l][1]) return fail(FPX_ERR_STATE, "wetdepo: nested grids are configured: fpx_upload_wet_nest_fields (both slots) for every nest first");
    if (!height_set || !window_set) return fail(FPX_ERR_STATE, "wetdepo: height / wind-time window not set");
    if (numpart == 0) return 0;
    const int nb = (int)((numpart + kBlock - 1) / kBlock);
    red_valid[RG_WET] = red_valid[RG_WETN] = false;
    k_wetdepo<R><<<nb, kBlock, 0, stream>>>(V, Gp, Wp, P, numpart, itime, ltsample, loutnext);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int get_wetgrid(void *wetgridunc, int allreduce) override {
    if (!Gp.on) return fail(FPX_ERR_STATE, "get_wetgrid: fpx_outgrid_init first");
    const float *gw = Gp.wetgridunc;
    if (allreduce && comm_ranks > 1) {
      int rc = reduce_into(Gp.wetgridunc, &wetgridunc0, n_grid2, "get_wetgrid");   // mpi_mod.f90:2486-2488, into wetgridunc0
      if (rc) return rc;
      red_valid[RG_WET] = true;
      gw = wetgridunc0;
    }
    if (wetgridunc) HIPCHK(hipMemcpyAsync(wetgridunc, gw, n_grid2 * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
  }

  void *stream_ptr() override { return (void *)stream; }
};

FPX_TU_CLOSE
#if FPX_TU_PART == 1 || FPX_TU_PART < 0
template <typename R>
static const void *prep_table(bool drydep, bool init, bool polar, bool nest) {
#define FPX_PREP(DD, II) (polar ? (nest ? (const void *)k_prep<R, DD, II, true, true> : (const void *)k_prep<R, DD, II, true, false>) \
                                : (nest ? (const void *)k_prep<R, DD, II, false, true> : (const void *)k_prep<R, DD, II, false, false>))
  if (drydep) return init ? FPX_PREP(true, true) : FPX_PREP(true, false);
  return init ? FPX_PREP(false, true) : FPX_PREP(false, false);
#undef FPX_PREP
}
#if FPX_TU_REAL != 4
const void *step_kernel_prep_f64(bool drydep, bool init, bool polar, bool nest) { return prep_table<double>(drydep, init, polar, nest); }
#endif
#if FPX_TU_REAL != 8
const void *step_kernel_prep_f32(bool drydep, bool init, bool polar, bool nest) { return prep_table<float>(drydep, init, polar, nest); }
#endif
#endif
#if FPX_TU_PART == 2 || FPX_TU_PART < 0
// the Langevin kernel specialised for the run's switches (gases: LEAN) or the general one
template <typename R, bool SUSP>
static const void *loop_table_s(bool lean, int turbswitch, int cblflag, int rng_mode, bool *is_lean) {
  const bool philox = rng_mode == FPX_RNG_PHILOX;
  *is_lean = false;
  if (turbswitch < 0) return (const void *)k_pbl_loop<R, false, -1, -1, -1, SUSP>;   // turboff: the general instance (it alone has the switch)
  if (lean) {
    *is_lean = true;
    if (turbswitch && cblflag == 1) return philox ? (const void *)k_pbl_loop<R, true, 1, 1, 2, SUSP> : (const void *)k_pbl_loop<R, true, 1, 1, 0, SUSP>;
    if (turbswitch && cblflag != 1) return philox ? (const void *)k_pbl_loop<R, true, 1, 0, 2, SUSP> : (const void *)k_pbl_loop<R, true, 1, 0, 0, SUSP>;
    if (!turbswitch && cblflag != 1) return philox ? (const void *)k_pbl_loop<R, true, 0, 0, 2, SUSP> : (const void *)k_pbl_loop<R, true, 0, 0, 0, SUSP>;
    *is_lean = false;
  } else if (philox) {   // aerosols (settling / dry deposition) with the counter RNG: same switch specialisation
    if (turbswitch && cblflag == 1) return (const void *)k_pbl_loop<R, false, 1, 1, 2, SUSP>;
    if (turbswitch && cblflag != 1) return (const void *)k_pbl_loop<R, false, 1, 0, 2, SUSP>;
  }
  return (const void *)k_pbl_loop<R, false, -1, -1, -1, SUSP>;
}
// (a step of several launches runs the SUSP twin of the SAME specialisation: the arithmetic of a particle must not depend on
// how its step is scheduled -- two instances with different compile-time switches round differently)
template <typename R>
static const void *loop_table(bool lean, int turbswitch, int cblflag, int rng_mode, bool susp, bool *is_lean) {
  return susp ? loop_table_s<R, true>(lean, turbswitch, cblflag, rng_mode, is_lean) : loop_table_s<R, false>(lean, turbswitch, cblflag, rng_mode, is_lean);
}
template <typename R>
static const void *finish_table(bool drydep, bool polar, bool nest) {
  if (drydep) return polar ? (nest ? (const void *)k_pbl_finish<R, true, true, true> : (const void *)k_pbl_finish<R, true, true, false>)
                           : (nest ? (const void *)k_pbl_finish<R, true, false, true> : (const void *)k_pbl_finish<R, true, false, false>);
  return polar ? (nest ? (const void *)k_pbl_finish<R, false, true, true> : (const void *)k_pbl_finish<R, false, true, false>)
               : (nest ? (const void *)k_pbl_finish<R, false, false, true> : (const void *)k_pbl_finish<R, false, false, false>);
}
#if FPX_TU_REAL != 4
const void *step_kernel_loop_f64(bool lean, int turbswitch, int cblflag, int rng_mode, bool susp, bool *is_lean) { return loop_table<double>(lean, turbswitch, cblflag, rng_mode, susp, is_lean); }
const void *step_kernel_finish_f64(bool drydep, bool polar, bool nest) { return finish_table<double>(drydep, polar, nest); }
#endif
#if FPX_TU_REAL != 8
const void *step_kernel_loop_f32(bool lean, int turbswitch, int cblflag, int rng_mode, bool susp, bool *is_lean) { return loop_table<float>(lean, turbswitch, cblflag, rng_mode, susp, is_lean); }
const void *step_kernel_finish_f32(bool drydep, bool polar, bool nest) { return finish_table<float>(drydep, polar, nest); }
#endif
#endif
#if FPX_TU_PART <= 0
#if FPX_TU_REAL != 8
EngineBase *make_engine_f32(const fpx_config *cfg, int *rc) {
  auto *p = new (std::nothrow) Engine<float>();
  if (!p) { *rc = fail(FPX_ERR_NOMEM, "fpx_create: out of host memory"); return nullptr; }
  *rc = p->init(cfg);
  return p;
}
#endif
#if FPX_TU_REAL != 4
const void *step_kernel_prep_f64(bool, bool, bool, bool);
const void *step_kernel_prep_f32(bool, bool, bool, bool);
const void *step_kernel_loop_f64(bool, int, int, int, bool, bool *);
const void *step_kernel_loop_f32(bool, int, int, int, bool, bool *);
const void *step_kernel_finish_f64(bool, bool, bool);
const void *step_kernel_finish_f32(bool, bool, bool);
const void *step_kernel_prep(int rb, bool drydep, bool init, bool polar, bool nest) {
  return rb == 8 ? step_kernel_prep_f64(drydep, init, polar, nest) : step_kernel_prep_f32(drydep, init, polar, nest);
}
const void *step_kernel_loop(int rb, bool lean, int turbswitch, int cblflag, int rng_mode, bool susp, bool *is_lean) {
  return rb == 8 ? step_kernel_loop_f64(lean, turbswitch, cblflag, rng_mode, susp, is_lean) : step_kernel_loop_f32(lean, turbswitch, cblflag, rng_mode, susp, is_lean);
}
const void *step_kernel_finish(int rb, bool drydep, bool polar, bool nest) {
  return rb == 8 ? step_kernel_finish_f64(drydep, polar, nest) : step_kernel_finish_f32(drydep, polar, nest);
}
#endif
#endif   // FPX_TU_PART <= 0
}  // namespace fpx

#if FPX_TU_REAL != 4 && FPX_TU_PART <= 0
// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
struct fpx_engine {
  fpx::EngineBase *impl;
};

#define FPX_GUARD(h)                                                   \
  if (!(h) || !(h)->impl) return fpx::fail(FPX_ERR_ARG, "null handle")

extern "C" {

int fpx_abi_version(void) { return 4; }

int fpx_polar_maps(int32_t host_real_bytes, double dy, double north[9], double south[9]) {
  if (!north || !south || !(dy > 0)) return fpx::fail(FPX_ERR_ARG, "fpx_polar_maps: bad argument");
  if (host_real_bytes == 4) {
    float n[9], s[9];
    fpx::polar_maps<float>((float)dy, n, s);
    for (int i = 0; i < 9; i++) { north[i] = n[i]; south[i] = s[i]; }
  } else if (host_real_bytes == 8) {
    fpx::polar_maps<double>(dy, north, south);
  } else {
    return fpx::fail(FPX_ERR_ARG, "fpx_polar_maps: host_real_bytes must be 4 or 8");
  }
  return FPX_OK;
}
const char *fpx_last_error(void) { return fpx::g_err.c_str(); }

int fpx_create(fpx_handle *out, const fpx_config *cfg) {
  if (!out || !cfg) return fpx::fail(FPX_ERR_ARG, "fpx_create: null argument");
  *out = nullptr;
  if (cfg->struct_bytes != (int32_t)sizeof(fpx_config)) return fpx::fail(FPX_ERR_ARG, "fpx_create: fpx_config size mismatch (ABI)");
  if (cfg->host_real_bytes != 4 && cfg->host_real_bytes != 8) return fpx::fail(FPX_ERR_ARG, "fpx_create: host_real_bytes must be 4 or 8");
  if (cfg->rng_mode < 0 || cfg->rng_mode > 2) return fpx::fail(FPX_ERR_ARG, "fpx_create: bad rng_mode");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fpx::fail(FPX_ERR_DEVICE, "fpx_create: no HIP device visible");
  if (cfg->device < 0 || cfg->device >= ndev) return fpx::fail(FPX_ERR_ARG, "fpx_create: device ordinal out of range");
  fpx::EngineBase *e = nullptr;
  int rc;
  if (cfg->compute_real_bytes == 8) {
    auto *p = new (std::nothrow) fpx::Engine<double>();
    if (!p) return fpx::fail(FPX_ERR_NOMEM, "fpx_create: out of host memory");
    rc = p->init(cfg);
    e = p;
  } else if (cfg->compute_real_bytes == 4) {
    e = fpx::make_engine_f32(cfg, &rc);
    if (!e) return rc;
  } else {
    return fpx::fail(FPX_ERR_ARG, "fpx_create: compute_real_bytes must be 4 or 8");
  }
  if (rc) { delete e; return rc; }
  fpx_engine *h = new (std::nothrow) fpx_engine{e};
  if (!h) { delete e; return fpx::fail(FPX_ERR_NOMEM, "fpx_create: out of host memory"); }
  *out = h;
  return FPX_OK;
}

int fpx_destroy(fpx_handle h) {
  if (!h) return FPX_OK;
  delete h->impl;
  delete h;
  return FPX_OK;
}

int fpx_set_height(fpx_handle h, const void *height, int32_t n) { FPX_GUARD(h); return h->impl->set_height(height, n); }
int fpx_upload_fields(fpx_handle h, int32_t slot, const fpx_fields *f) { FPX_GUARD(h); return h->impl->upload_fields(slot, f); }
int fpx_verttransform_ecmwf(fpx_handle h, int32_t slot, const fpx_model_levels *m, const fpx_fields *sfc, const fpx_fields_out *out) { FPX_GUARD(h); return h->impl->verttransform(slot, m, sfc, out); }
int fpx_verttransform_nest(fpx_handle h, int32_t nest, int32_t slot, const fpx_model_levels *m, const fpx_fields *sfc, const fpx_fields_out *out) { FPX_GUARD(h); return h->impl->verttransform_nest(nest, slot, m, sfc, out); }
int fpx_upload_diag_fields(fpx_handle h, int32_t slot, const fpx_diag_fields *f) { FPX_GUARD(h); return h->impl->upload_diag_fields(slot, f); }
int fpx_partoutput(fpx_handle h, int32_t itime, const char *path, int64_t *nparticles) { FPX_GUARD(h); return h->impl->partoutput(itime, path, nparticles); }
int fpx_readpartpositions(fpx_handle h, const char *path, const fpx_restart *r, int64_t *numpart, int32_t *numparticlecount, int32_t *itimein) {
  FPX_GUARD(h);
  return h->impl->readpartpositions(path, r, numpart, numparticlecount, itimein);
}
int fpx_concoutput(fpx_handle h, int32_t itime, const fpx_concout *c, const char *prefix, int32_t clear) { FPX_GUARD(h); return h->impl->concoutput(itime, c, prefix, clear); }
int fpx_partoutput_time(fpx_handle h, double *ms) { FPX_GUARD(h); if (!ms) return fpx::fail(FPX_ERR_ARG, "fpx_partoutput_time: null"); *ms = h->impl->po_ms(); return FPX_OK; }
int fpx_calcpar(fpx_handle h, int32_t slot, const fpx_calcpar_in *c, const fpx_calcpar_out *out) { FPX_GUARD(h); return h->impl->calcpar(slot, c, out); }
int fpx_calcpar_time(fpx_handle h, double *ms) { FPX_GUARD(h); if (!ms) return fpx::fail(FPX_ERR_ARG, "fpx_calcpar_time: null"); *ms = h->impl->cp_ms(); return FPX_OK; }
int fpx_verttransform_time(fpx_handle h, double *ms) { FPX_GUARD(h); if (!ms) return fpx::fail(FPX_ERR_ARG, "fpx_verttransform_time: null"); *ms = h->impl->vt_ms(); return FPX_OK; }
int fpx_set_windtime(fpx_handle h, const int32_t memtime[2], const int32_t memind[2]) { FPX_GUARD(h); return h->impl->set_windtime(memtime, memind); }
int fpx_rng_fill_table(fpx_handle h) { FPX_GUARD(h); return h->impl->rng_fill_table(); }
int fpx_rng_set_table(fpx_handle h, const void *t, int32_t n) { FPX_GUARD(h); return h->impl->rng_set_table(t, n); }
int fpx_rng_get_table(fpx_handle h, void *t, int32_t n) { FPX_GUARD(h); return h->impl->rng_get_table(t, n); }
int fpx_upload_particles(fpx_handle h, int64_t first, int64_t count, const fpx_particles *p) { FPX_GUARD(h); return h->impl->upload_particles(first, count, p); }
int fpx_download_particles(fpx_handle h, int64_t first, int64_t count, const fpx_particles *p) { FPX_GUARD(h); return h->impl->download_particles(first, count, p); }
int fpx_set_numpart(fpx_handle h, int64_t n) { FPX_GUARD(h); return h->impl->set_numpart(n); }
int fpx_set_release_points(fpx_handle h, int32_t numpoint, const void *xmass, const int32_t *npart) { FPX_GUARD(h); return h->impl->set_release_points(numpoint, xmass, npart); }
int fpx_release_init(fpx_handle h, const fpx_release *r) { FPX_GUARD(h); return h->impl->release_init(r); }
int fpx_releaseparticles(fpx_handle h, int32_t itime, int64_t *numpart, int32_t *numparticlecount, void *xmasssave, void *rho_rel, int64_t *nreleased) {
  FPX_GUARD(h);
  return h->impl->releaseparticles(itime, numpart, numparticlecount, xmasssave, rho_rel, nreleased);
}
int fpx_split_particles(fpx_handle h, int32_t itime, int64_t *numpart) { FPX_GUARD(h); return h->impl->split_particles(itime, numpart); }
// mpi_mod.f90:566-658 (mpif_calculate_part_redist): which pairs of ranks exchange how many particles.  Pure host logic.
int fpx_redist_plan(const int64_t *npart_per_process, int32_t nranks, int32_t rank, int32_t ipout, int32_t *role, int32_t *peer, int64_t *num_trans) {
  if (!role || !peer || !num_trans) return FPX_ERR_ARG;
  *role = 0; *peer = -1; *num_trans = 0;
  if (!npart_per_process || nranks < 1 || rank < 0 || rank >= nranks) return FPX_ERR_ARG;
  if (nranks == 1 || ipout == 3) return 0;                           // :597, :613
  const double mp_redist_fract = 0.2;                                // mpi_mod.f90:156-157
  const int64_t mp_min_redist = 100000;
  std::vector<float> sorted((size_t)nranks);                         // the reference sorts the counts as default reals
  std::vector<int> idx((size_t)nranks);
  for (int i = 0; i < nranks; i++) { sorted[i] = (float)npart_per_process[i]; idx[i] = i; }
  for (int i = 0; i <= nranks - 2; i++) {                            // :616-631, the reference's selection sort, ties included
    float pmin = sorted[i];
    int imin = idx[i];
    for (int jj = i + 1; jj <= nranks - 1; jj++) {
      if (pmin <= sorted[jj]) continue;
      const float z = pmin; pmin = sorted[jj]; sorted[jj] = z;
      const int nn = imin; imin = idx[jj]; idx[jj] = nn;
    }
    sorted[i] = pmin; idx[i] = imin;
  }
  int m = nranks - 1;
  for (int i = 0; i <= nranks / 2 - 1; i++, m--) {                   // :639-655
    const int64_t hi = npart_per_process[idx[m]], lo = npart_per_process[idx[i]], nt = hi - lo;
    if (rank != idx[m] && rank != idx[i]) continue;
    if (hi > mp_min_redist && (float)nt / (float)hi > (float)mp_redist_fract) {
      *role = rank == idx[m] ? 1 : 2;
      *peer = rank == idx[m] ? idx[i] : idx[m];
      *num_trans = nt / 2;
    }
  }
  return 0;
}
uint64_t fpx_redist_bytes(fpx_handle h, int64_t num_trans) { if (!h || !h->impl) return 0; return (uint64_t)h->impl->redist_bytes(num_trans); }
int fpx_redist_pack(fpx_handle h, int32_t itime, int64_t num_trans, void *buf, uint64_t buf_bytes, int64_t *numpart) { FPX_GUARD(h); return h->impl->redist_pack(itime, num_trans, buf, (size_t)buf_bytes, numpart); }
int fpx_redist_unpack(fpx_handle h, int32_t itime, int64_t num_trans, const void *buf, uint64_t buf_bytes, int64_t *numpart) { FPX_GUARD(h); return h->impl->redist_unpack(itime, num_trans, buf, (size_t)buf_bytes, numpart); }
int fpx_step(fpx_handle h, int32_t itime, fpx_step_stats *st) { FPX_GUARD(h); return h->impl->step(itime, st, false); }
int fpx_step_async(fpx_handle h, int32_t itime) { FPX_GUARD(h); return h->impl->step(itime, nullptr, true); }
int fpx_sync(fpx_handle h) { FPX_GUARD(h); return h->impl->sync(); }
int fpx_counters(fpx_handle h, fpx_step_stats *st, int32_t reset) { FPX_GUARD(h); return h->impl->counters(st, reset); }
int fpx_kernel_time(fpx_handle h, double *ms, int64_t *launches, int32_t reset) {
  FPX_GUARD(h);
  long long l = 0;
  int rc = h->impl->kernel_time(ms, &l, reset, nullptr);
  if (launches) *launches = l;
  return rc;
}
int fpx_kernel_times(fpx_handle h, double ms[4], int64_t *launches, int32_t reset) {
  FPX_GUARD(h);
  long long l = 0;
  double tot = 0;
  int rc = h->impl->kernel_time(&tot, &l, reset, ms);
  if (launches) *launches = l;
  return rc;
}
int fpx_sort_particles(fpx_handle h) { FPX_GUARD(h); return h->impl->sort_particles(); }
int fpx_conv_init(fpx_handle h, const fpx_conv_config *c) { FPX_GUARD(h); return h->impl->conv_init(c); }
int fpx_upload_conv_fields(fpx_handle h, int32_t slot, const fpx_conv_fields *f) { FPX_GUARD(h); return h->impl->upload_conv_fields(slot, f); }
int fpx_convmix(fpx_handle h, int32_t itime, int64_t *nmoved) { FPX_GUARD(h); return h->impl->convmix(itime, nmoved); }
int fpx_convmix_time(fpx_handle h, double *ms) { FPX_GUARD(h); if (!ms) return FPX_ERR_ARG; *ms = h->impl->conv_ms(); return 0; }
int fpx_upload_diag_nest_fields(fpx_handle h, int32_t nest, int32_t slot, const fpx_diag_fields *f) { FPX_GUARD(h); return h->impl->upload_diag_nest_fields(nest, slot, f); }
int fpx_upload_conv_nest_fields(fpx_handle h, int32_t nest, int32_t slot, const fpx_conv_fields *f) { FPX_GUARD(h); return h->impl->upload_conv_nest_fields(nest, slot, f); }
int fpx_get_cbaseflux_nest(fpx_handle h, int32_t nest, void *cb) { FPX_GUARD(h); return h->impl->cbaseflux_nest_io(nest, cb, false); }
int fpx_set_cbaseflux_nest(fpx_handle h, int32_t nest, const void *cb) { FPX_GUARD(h); return h->impl->cbaseflux_nest_io(nest, (void *)cb, true); }
int fpx_get_cbaseflux(fpx_handle h, void *cb) { FPX_GUARD(h); return h->impl->cbaseflux_io(cb, false); }
int fpx_set_cbaseflux(fpx_handle h, const void *cb) { FPX_GUARD(h); return h->impl->cbaseflux_io((void *)cb, true); }
int fpx_checkpoint_write(fpx_handle h, const char *path, int32_t itime, int32_t numparticlecount) {
  FPX_GUARD(h);
  return h->impl->checkpoint_write(path, itime, numparticlecount);
}
int fpx_checkpoint_read(fpx_handle h, const char *path, int32_t *itime, int64_t *numpart, int32_t *numparticlecount) {
  FPX_GUARD(h);
  return h->impl->checkpoint_read(path, itime, numpart, numparticlecount);
}
int fpx_seed_particles(fpx_handle h, int64_t n, uint64_t seed, double frac_pbl, double zmax, double lat_margin_cells, int32_t itime0) {
  FPX_GUARD(h);
  return h->impl->seed_particles(n, seed, frac_pbl, zmax, lat_margin_cells, itime0);
}
int fpx_outgrid_init(fpx_handle h, const fpx_outgrid *g, const void *outheight) { FPX_GUARD(h); return h->impl->outgrid_init(g, outheight); }
int fpx_set_output_times(fpx_handle h, int32_t loutnext, int32_t loutstep) { FPX_GUARD(h); return h->impl->set_output_times(loutnext, loutstep); }
int fpx_conccalc(fpx_handle h, int32_t itime, double weight) { FPX_GUARD(h); return h->impl->conccalc(itime, weight); }
int fpx_get_grids(fpx_handle h, void *gridunc, void *drygridunc, int32_t allreduce, int32_t clear) { FPX_GUARD(h); return h->impl->get_grids(gridunc, drygridunc, allreduce, clear); }
int fpx_comm_unique_id(void *id, int32_t nbytes) {
  if (!id || nbytes != (int32_t)sizeof(ncclUniqueId)) return fpx::fail(FPX_ERR_ARG, "fpx_comm_unique_id: the id is 128 bytes");
  ncclUniqueId uid;
  ncclResult_t r = ncclGetUniqueId(&uid);
  if (r != ncclSuccess) return fpx::fail(FPX_ERR_DEVICE, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
  memcpy(id, &uid, sizeof(uid));
  return FPX_OK;
}
int fpx_count_particles(fpx_handle h, int64_t local[2], int64_t total[2], int32_t allreduce) { FPX_GUARD(h); return h->impl->count_particles(local, total, allreduce); }
int fpx_set_release_heights(fpx_handle h, int32_t numpoint, const void *zpoint1, const void *zpoint2) { FPX_GUARD(h); return h->impl->set_release_heights(numpoint, zpoint1, zpoint2); }
int fpx_set_option(fpx_handle h, const char *name, const char *value) { FPX_GUARD(h); if (!name || !value) return fpx::fail(FPX_ERR_ARG, "fpx_set_option: null argument"); return h->impl->set_option(name, value); }
int fpx_get_info(fpx_handle h, const char *name, int64_t *value) { FPX_GUARD(h); if (!name || !value) return fpx::fail(FPX_ERR_ARG, "fpx_get_info: null argument"); return h->impl->get_info(name, value); }
int fpx_lane_stats(fpx_handle h, uint64_t *out, int32_t n, int32_t reset) { FPX_GUARD(h); if (!out || n < 0) return fpx::fail(FPX_ERR_ARG, "fpx_lane_stats: bad argument"); return h->impl->lane_stats(out, n, reset); }
int fpx_comm_init(fpx_handle h, const void *id, int32_t nbytes, int32_t nranks, int32_t rank) { FPX_GUARD(h); return h->impl->comm_init(id, nbytes, nranks, rank); }
int fpx_comm_init_host(fpx_handle h, int32_t nranks, int32_t rank, fpx_allreduce_fn fn, void *user) { FPX_GUARD(h); return h->impl->comm_init_host(nranks, rank, fn, user); }
int fpx_nests_init(fpx_handle h, const fpx_nests *n) { FPX_GUARD(h); return h->impl->nests_init(n); }
int fpx_upload_nest_fields(fpx_handle h, int32_t nest, int32_t slot, const fpx_fields *f) { FPX_GUARD(h); return h->impl->upload_nest_fields(nest, slot, f); }
int fpx_wet_init(fpx_handle h, const fpx_wet_config *w) { FPX_GUARD(h); return h->impl->wet_init(w); }
int fpx_upload_wet_fields(fpx_handle h, int32_t slot, const fpx_wet_fields *f) { FPX_GUARD(h); return h->impl->upload_wet_fields(slot, f); }
int fpx_wetdepo(fpx_handle h, int32_t itime, int32_t ltsample, int32_t loutnext) { FPX_GUARD(h); return h->impl->wetdepo(itime, ltsample, loutnext); }
int fpx_get_wetgrid(fpx_handle h, void *wetgridunc, int32_t allreduce) { FPX_GUARD(h); return h->impl->get_wetgrid(wetgridunc, allreduce); }
void *fpx_stream(fpx_handle h) { return (h && h->impl) ? h->impl->stream_ptr() : nullptr; }
int fpx_upload_wet_nest_fields(fpx_handle h, int32_t nest, int32_t slot, const fpx_wet_fields *f, int32_t readclouds_nest) { FPX_GUARD(h); return h->impl->upload_wet_nest_fields(nest, slot, f, readclouds_nest); }
int fpx_outgrid_nest_init(fpx_handle h, const fpx_outgrid_nest *g) { FPX_GUARD(h); return h->impl->outgrid_nest_init(g); }
int fpx_get_grids_nest(fpx_handle h, void *griduncn, void *drygriduncn, void *wetgriduncn, int32_t allreduce, int32_t clear) { FPX_GUARD(h); return h->impl->get_grids_nest(griduncn, drygriduncn, wetgriduncn, allreduce, clear); }
int fpx_receptors_init(fpx_handle h, int32_t numreceptor, const void *xreceptor, const void *yreceptor, const void *receptorarea) { FPX_GUARD(h); return h->impl->receptors_init(numreceptor, xreceptor, yreceptor, receptorarea); }
int fpx_get_receptors(fpx_handle h, void *creceptor, int32_t ld, int32_t allreduce, int32_t clear) { FPX_GUARD(h); return h->impl->get_receptors(creceptor, ld, allreduce, clear); }

int fpx_math_probe(int32_t fn, const double *x, double *y, int64_t n) {
  if (fn < 0 || fn > 11 || !x || !y || n < 0) return FPX_ERR_ARG;
  if (n == 0) return FPX_OK;
  double *dx = nullptr, *dy = nullptr;
  if (hipMalloc(&dx, n * sizeof(double)) != hipSuccess) return FPX_ERR_NOMEM;
  if (hipMalloc(&dy, n * sizeof(double)) != hipSuccess) { (void)hipFree(dx); return FPX_ERR_NOMEM; }
  hipError_t e = hipMemcpy(dx, x, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    fpx::k_math_probe<<<(unsigned)((n + fpx::kBlock - 1) / fpx::kBlock), fpx::kBlock>>>(fn, dx, dy, n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(y, dy, n * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy);
  return e == hipSuccess ? FPX_OK : FPX_ERR_DEVICE;
}

}  // extern "C"
#endif   // FPX_TU_REAL != 4 && FPX_TU_PART <= 0
