// Device side of fpx_verttransform_ecmwf: the reference's verttransform_ecmwf
// (src/verttransform_ecmwf.f90:118-590 -- eta levels -> terrain-following z levels, rho,
// drhodz, w conversion and eta-slope correction, polar-stereographic winds) as HIP kernels
// for gfx950.  SURVEY.md section 8 (f) item 1: the grid-sized work that otherwise serialises
// the host once per wind interval, and the 722 MB field upload it makes unnecessary.
//
// Design: every array keeps the host's (ix,jy,level) x-fastest layout (strides nxmax, nymax) and
// consecutive lanes own consecutive ix, so each access of a wave is one contiguous row segment.
// The transform is level-parallel except for the running sums and level searches of a column,
// which one light kernel does per column (kernel plan below).  HBM-bound.
// Arithmetic is done in the host's real kind H (the reference computes in its default real)
// with FMA contraction off, so only the libm calls (log, 10**x, cos, sin, atan) can differ
// from the CPU result.
#pragma once
#include "fpx_tu.hpp"
#include <hip/hip_runtime.h>

namespace fpx {
FPX_TU_OPEN
namespace vt {

template <typename H> struct M;
template <> struct M<float> {
  static __device__ __forceinline__ float log(float x) { return ::logf(x); }
  static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
  static __device__ __forceinline__ float pow(float x, float y) { return ::powf(x, y); }
  static __device__ __forceinline__ float cos(float x) { return ::cosf(x); }
  static __device__ __forceinline__ float sin(float x) { return ::sinf(x); }
  static __device__ __forceinline__ float atan(float x) { return ::atanf(x); }
  static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
  static __device__ __forceinline__ float fmod(float x, float y) { return ::fmodf(x, y); }
};
template <> struct M<double> {
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
  static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
  static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
  static __device__ __forceinline__ double atan(double x) { return ::atan(x); }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double fmod(double x, double y) { return ::fmod(x, y); }
};

template <typename H>
struct Geo {
  int nx, ny, nz, nuvz, nwz, nxmax, nymax;
  H dx, dy, xlon0, ylat0, dxconst, dyconst;
  H xres, yres;               // 1 on the mother grid; xresoln(l), yresoln(l) on a nest (verttransform_nests.f90:384-385)
  int nglobal, sglobal;
  H switchnorthg, switchsouthg;
  H northpolemap[9], southpolemap[9];
  __device__ __forceinline__ size_t at(int ix, int jy, int k /* 1-based */) const {
    return (size_t)ix + (size_t)nxmax * ((size_t)jy + (size_t)nymax * (size_t)(k - 1));
  }
  __device__ __forceinline__ size_t at2(int ix, int jy) const { return (size_t)ix + (size_t)nxmax * (size_t)jy; }
};

template <typename H>
struct In {   // model-level input, host layout
  const H *uuh, *vvh, *pvh, *wwh, *tth, *qvh, *ps, *tt2, *td2;
  const H *akz, *bkz, *aknew, *bknew, *height;
};
template <typename H>
struct Out {  // z-level output (host layout) and scratch
  H *uu, *vv, *ww, *tt, *qv, *pv, *rho, *drhodz, *uupol, *vvpol;
  H *uvzlev, *wzlev, *rhoh, *pinmconv;
};

#define VK(x) ((H)(x))

// ew.f90:4-29 (Goff-Gratch saturation vapour pressure over water)
template <typename H>
__device__ __forceinline__ H ew(H x) {
#pragma clang fp contract(off)
  H y = VK(373.16) / x;
  H a = VK(-7.90298) * (y - VK(1.));
  a = a + (VK(5.02808) * VK(0.43429) * M<H>::log(y));
  H c = (VK(1.) - (VK(1.) / y)) * VK(11.344);
  c = VK(-1.) + M<H>::pow(VK(10.), c);
  c = VK(-1.3816) * c / VK(1.e7);
  H d = (VK(1.) - y) * VK(3.49149);
  d = VK(-1.) + M<H>::pow(VK(10.), d);
  d = VK(8.1328) * d / VK(1.e3);
  y = a + c + d;
  return VK(101324.6) * M<H>::pow(VK(10.), y);
}

// cmapf_mod.f90:494-524
template <typename H>
__device__ __forceinline__ H cspanf(H value, H begin, H end) {
#pragma clang fp contract(off)
  const H first = begin < end ? begin : end, last = begin > end ? begin : end;
  const H val = M<H>::fmod(value - first, last - first);
  return val <= VK(0.) ? val + last : val + first;
}

// cmapf_mod.f90:24-52 (its own pi, :19; double internals as declared there)
template <typename H>
__device__ __forceinline__ void cc2gll(const H *s, H xlat, H xlong, H ue, H vn, H &ug, H &vg) {
#pragma clang fp contract(off)
  const H radpdg = VK(3.14159265358979) / VK(180.);
  const double along = (double)cspanf<H>(xlong - s[1], VK(-180.), VK(180.));
  double rot;
  if (xlat > VK(89.985)) rot = -(double)s[0] * along + (double)xlong - 180.;
  else if (xlat < VK(-89.985)) rot = -(double)s[0] * along - (double)xlong;
  else rot = -(double)s[0] * along;
  const double slong = ::sin((double)radpdg * rot), clong = ::cos((double)radpdg * rot);
  const double xpolg = slong * (double)s[4] + clong * (double)s[5];
  const double ypolg = clong * (double)s[4] - slong * (double)s[5];
  ug = (H)(ypolg * (double)ue + xpolg * (double)vn);
  vg = (H)(ypolg * (double)vn - xpolg * (double)ue);
}

// Kernel plan.  Only the running sums and the level searches of a column are sequential; they are
// kept in one light kernel (k_vt_column) and everything else is level-parallel: one lane per
// (ix,jy,level) with ix fastest, i.e. every load and store of a wave is a contiguous row segment.
//   k_vt_inc    (ix,jy,kz)  layer thickness of every eta layer and the density on eta levels
//   k_vt_column (ix,jy)     uvzlev = running sum, wzlev, pinmconv
//   k_vt_search (ix,jy) x 2 the level index of every z level in the u/v and in the w sweep
//   k_vt_fill   (ix,jy,iz)  vertical interpolation of u,v,T,q,pv,rho and w
//   k_vt_post   (ix,jy,iz)  drhodz and the eta-slope correction of w
//   k_vt_polar / k_vt_polerow  the polar caps

// layer thickness uvzlev(kz)-uvzlev(kz-1) (stored in uvzlev until k_vt_column sums it up) and rhoh,
// verttransform_ecmwf.f90:203-237
template <typename H>
__global__ void __launch_bounds__(256) k_vt_inc(Geo<H> G, In<H> I, Out<H> O) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x, kz = 1 + (int)blockIdx.y;
  if (t >= G.nx * G.ny) return;
  const int ix = t % G.nx, jy = t / G.nx;
  const H r_air = VK(287.05), konst = VK(287.05) / VK(9.81);
  const H p = I.ps[G.at2(ix, jy)];
  const size_t o = G.at(ix, jy, kz);
  if (kz == 1) {
    const H tvold = I.tt2[G.at2(ix, jy)] * (VK(1.) + VK(0.378) * ew<H>(I.td2[G.at2(ix, jy)]) / p);
    O.uvzlev[o] = VK(0.);
    O.rhoh[o] = p / (r_air * tvold);
    return;
  }
  H tvold, pold;
  if (kz == 2) {
    tvold = I.tt2[G.at2(ix, jy)] * (VK(1.) + VK(0.378) * ew<H>(I.td2[G.at2(ix, jy)]) / p);
    pold = p;
  } else {
    const size_t l = G.at(ix, jy, kz - 1);
    tvold = I.tth[l] * (VK(1.) + VK(0.608) * I.qvh[l]);
    pold = I.akz[kz - 2] + I.bkz[kz - 2] * p;
  }
  const H pint = I.akz[kz - 1] + I.bkz[kz - 1] * p;
  const H tv = I.tth[o] * (VK(1.) + VK(0.608) * I.qvh[o]);
  O.rhoh[o] = pint / (r_air * tv);
  const H dtv = tv - tvold;
  H inc;
  if ((dtv < 0 ? -dtv : dtv) > VK(0.2)) inc = konst * M<H>::log(pold / pint) * (tv - tvold) / M<H>::log(tv / tvold);
  else inc = konst * M<H>::log(pold / pint) * tv;
  O.uvzlev[o] = inc;
}

// the sequential part of a column, 1: uvzlev as the running sum of the layer thicknesses (:226-237),
// wzlev (:240-244) and pinmconv (:248-258) from the sliding window of the last three levels
template <typename H>
__global__ void __launch_bounds__(256) k_vt_column(Geo<H> G, In<H> I, Out<H> O) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.nx * G.ny) return;
  const int ix = t % G.nx, jy = t / G.nx;
  const int nz = G.nz, nuvz = G.nuvz, nwz = G.nwz;
  const H p = I.ps[G.at2(ix, jy)];
  {
    H uvz = VK(0.), um1 = VK(0.), um2 = VK(0.);   // uvzlev(kz), (kz-1), (kz-2)
    O.wzlev[G.at(ix, jy, 1)] = VK(0.);
    for (int kz = 2; kz <= nuvz; kz++) {
      um2 = um1; um1 = uvz;
      uvz = uvz + O.uvzlev[G.at(ix, jy, kz)];
      O.uvzlev[G.at(ix, jy, kz)] = uvz;
      if (kz >= 3 && kz - 1 <= nwz - 1) O.wzlev[G.at(ix, jy, kz - 1)] = (uvz + um1) / VK(2.);   // wzlev(kz-1)
      // pinmconv(kz-1) from uvzlev(kz), uvzlev(kz-2)
      if (kz == 2) O.pinmconv[G.at(ix, jy, 1)] = (uvz) / ((I.aknew[1] + I.bknew[1] * p) - (I.aknew[0] + I.bknew[0] * p));
      else if (kz - 1 <= nz - 1)
        O.pinmconv[G.at(ix, jy, kz - 1)] = (uvz - um2) / ((I.aknew[kz - 1] + I.bknew[kz - 1] * p) - (I.aknew[kz - 3] + I.bknew[kz - 3] * p));
    }
    // uvz = uvzlev(nuvz), um1 = uvzlev(nuvz-1)
    const H wlast = nwz - 1 >= 2 ? (O.uvzlev[G.at(ix, jy, nwz)] + O.uvzlev[G.at(ix, jy, nwz - 1)]) / VK(2.) : VK(0.);
    O.wzlev[G.at(ix, jy, nwz)] = wlast + uvz - um1;
    O.pinmconv[G.at(ix, jy, nz)] = (O.uvzlev[G.at(ix, jy, nz)] - O.uvzlev[G.at(ix, jy, nz - 1)]) /
                                   ((I.aknew[nz - 1] + I.bknew[nz - 1] * p) - (I.aknew[nz - 2] + I.bknew[nz - 2] * p));
  }
}

// the running level index idx(ix,jy) of the sweeps (:294-312 = :416-425, and :366-378) for every z level;
// blockIdx.y = 0: u/v (and eta-slope) sweep over uvzlev, 1: w sweep over wzlev
template <typename H>
__global__ void __launch_bounds__(256) k_vt_search(Geo<H> G, In<H> I, Out<H> O, unsigned short *__restrict__ kuv, unsigned short *__restrict__ kw) {
#pragma clang fp contract(off)
  extern __shared__ unsigned char vt_smem[];
  H *hgt = (H *)vt_smem;
  for (int k = threadIdx.x; k < G.nz; k += blockDim.x) hgt[k] = I.height[k];
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.nx * G.ny) return;
  const int ix = t % G.nx, jy = t / G.nx;
  const int nz = G.nz, nuvz = G.nuvz, nwz = G.nwz;
  if (blockIdx.y == 0)
  // level index of every z level in the u/v sweep (= the eta-slope sweep: same test, same start; above
  // the column top neither of them moves idx)
  {
    const H top = O.uvzlev[G.at(ix, jy, nuvz)];
    int idx = 2;
    H zlo = O.uvzlev[G.at(ix, jy, 1)], zhi = O.uvzlev[G.at(ix, jy, 2)];
    for (int iz = 2; iz <= nz - 1; iz++) {
      const H h = hgt[iz - 1];
      if (!(h > top)) {
        H a = zlo, b = zhi;
        for (int kz = idx; kz <= nuvz; kz++) {
          if (h > a && h <= b) { idx = kz; zlo = a; zhi = b; break; }
          if (kz == nuvz) break;
          a = b;
          b = O.uvzlev[G.at(ix, jy, kz + 1)];
        }
      }
      kuv[G.at(ix, jy, iz)] = (unsigned short)idx;
    }
  }
  else
  {
    int idx = 2;
    H zlo = O.wzlev[G.at(ix, jy, 1)], zhi = O.wzlev[G.at(ix, jy, 2)];
    for (int iz = 2; iz <= nz; iz++) {
      const H h = hgt[iz - 1];
      H a = zlo, b = zhi;
      for (int kz = idx; kz <= nwz; kz++) {
        if (h > a && h <= b) { idx = kz; zlo = a; zhi = b; break; }
        if (kz == nwz) break;
        a = b;
        b = O.wzlev[G.at(ix, jy, kz + 1)];
      }
      kw[G.at(ix, jy, iz)] = (unsigned short)idx;
    }
  }
}

// vertical interpolation onto the z levels: u,v,T,q,pv,rho (:264-356) and w (:362-389)
template <typename H>
__global__ void __launch_bounds__(256) k_vt_fill(Geo<H> G, In<H> I, Out<H> O, const unsigned short *__restrict__ kuv, const unsigned short *__restrict__ kw) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x, iz = 1 + (int)blockIdx.y;
  if (t >= G.nx * G.ny) return;
  const int ix = t % G.nx, jy = t / G.nx;
  const int nz = G.nz, nuvz = G.nuvz, nwz = G.nwz;
  const size_t o = G.at(ix, jy, iz);
  const H h = I.height[iz - 1];
  if (iz == 1 || iz == nz || h > O.uvzlev[G.at(ix, jy, nuvz)]) {
    const size_t s = iz == 1 ? o : G.at(ix, jy, nuvz);   // level 1, or the top eta level (also above the column top)
    O.uu[o] = I.uuh[s]; O.vv[o] = I.vvh[s]; O.tt[o] = I.tth[s]; O.qv[o] = I.qvh[s]; O.pv[o] = I.pvh[s]; O.rho[o] = O.rhoh[s];
  } else {
    const int kz = kuv[o];
    const size_t l = G.at(ix, jy, kz - 1), u = G.at(ix, jy, kz);
    const H dz1 = h - O.uvzlev[l], dz2 = O.uvzlev[u] - h, dz = dz1 + dz2;
    O.uu[o] = (I.uuh[l] * dz2 + I.uuh[u] * dz1) / dz;
    O.vv[o] = (I.vvh[l] * dz2 + I.vvh[u] * dz1) / dz;
    O.tt[o] = (I.tth[l] * dz2 + I.tth[u] * dz1) / dz;
    O.qv[o] = (I.qvh[l] * dz2 + I.qvh[u] * dz1) / dz;
    O.pv[o] = (I.pvh[l] * dz2 + I.pvh[u] * dz1) / dz;
    O.rho[o] = (O.rhoh[l] * dz2 + O.rhoh[u] * dz1) / dz;
  }
  if (iz == 1) {
    O.ww[o] = I.wwh[o] * O.pinmconv[o];
  } else {   // iz = nz included: the sweep of the reference overwrites ww(nz) (:372-388)
    const int kz = kw[o];
    const size_t l = G.at(ix, jy, kz - 1), u = G.at(ix, jy, kz);
    const H dz1 = h - O.wzlev[l], dz2 = O.wzlev[u] - h, dz = dz1 + dz2;
    O.ww[o] = (I.wwh[l] * O.pinmconv[l] * dz2 + I.wwh[u] * O.pinmconv[u] * dz1) / dz;
  }
  (void)nwz;
}

// density gradient (:394-400) and the slope of the eta levels in windward direction with the
// resulting correction of w (:411-453; interior columns, levels 2..nz-1)
template <typename H>
__global__ void __launch_bounds__(256) k_vt_post(Geo<H> G, In<H> I, Out<H> O, const unsigned short *__restrict__ kuv) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x, iz = 1 + (int)blockIdx.y;
  if (t >= G.nx * G.ny) return;
  const int ix = t % G.nx, jy = t / G.nx;
  const int nz = G.nz;
  const size_t o = G.at(ix, jy, iz);
  {
    const int kc = iz == nz ? nz - 1 : iz;     // drhodz(nz) = drhodz(nz-1)
    H d;
    if (kc == 1) d = (O.rho[G.at(ix, jy, 2)] - O.rho[G.at(ix, jy, 1)]) / (I.height[1] - I.height[0]);
    else d = (O.rho[G.at(ix, jy, kc + 1)] - O.rho[G.at(ix, jy, kc - 1)]) / (I.height[kc] - I.height[kc - 2]);
    O.drhodz[o] = d;
  }
  if (iz < 2 || iz > nz - 1 || ix < 1 || ix > G.nx - 2 || jy < 1 || jy > G.ny - 2) return;
  const H pi180 = VK(3.14159265) / VK(180.);
  const H cosf = VK(1.) / M<H>::cos(((H)jy * G.dy + G.ylat0) * pi180);
  const H h = I.height[iz - 1];
  const int kz = kuv[o];
  const H dz1 = h - O.uvzlev[G.at(ix, jy, kz - 1)], dz2 = O.uvzlev[G.at(ix, jy, kz)] - h, dz = dz1 + dz2;
  const H dzdx1 = (O.uvzlev[G.at(ix + 1, jy, kz - 1)] - O.uvzlev[G.at(ix - 1, jy, kz - 1)]) / VK(2.);
  const H dzdx2 = (O.uvzlev[G.at(ix + 1, jy, kz)] - O.uvzlev[G.at(ix - 1, jy, kz)]) / VK(2.);
  const H dzdx = (dzdx1 * dz2 + dzdx2 * dz1) / dz;
  const H dzdy1 = (O.uvzlev[G.at(ix, jy + 1, kz - 1)] - O.uvzlev[G.at(ix, jy - 1, kz - 1)]) / VK(2.);
  const H dzdy2 = (O.uvzlev[G.at(ix, jy + 1, kz)] - O.uvzlev[G.at(ix, jy - 1, kz)]) / VK(2.);
  const H dzdy = (dzdy1 * dz2 + dzdy2 * dz1) / dz;
  O.ww[o] = O.ww[o] + (dzdx * O.uu[o] * G.dxconst * G.xres * cosf + dzdy * O.vv[o] * G.dyconst * G.yres);
}

// ---------------------------------------------------------------------------------------------------------
// Fused path (nuvz = nwz = nz, the ECMWF case; the five kernels above remain for other level counts and as the
// reference the fused kernels are tested against bit for bit).  The unfused chain moves 2.9 GB per 361x181x138 fp64
// field against 1.04 GB of compulsory traffic: four scratch arrays (uvzlev, wzlev, rhoh, pinmconv) and two index
// arrays are written and read back, rho is read three times for drhodz, uvzlev ten times for the slope.  Here the
// only scratch array is uvzlev (the one sequential quantity: a running sum over the eta levels, and the only one a
// neighbouring column needs).  Everything else is recomputed from it with the same expressions in the same order,
// so every output keeps its bits:
//   k_vt_levels  tile of 16 x 4 columns; layer thicknesses level-parallel into LDS, running sum in LDS, one coalesced store
//   k_vt_fused   the same tile with its uvzlev in LDS; each wave walks a range of z levels with the reference's running
//                level indices (initialised by bisection: uvzlev and wzlev increase with the level), interpolates
//                u, v, T, q, pv, rho (rhoh recomputed from T, q), w (wzlev, pinmconv recomputed from uvzlev), keeps
//                rho of the last two levels for drhodz, and adds the eta-slope term from the neighbour columns' uvzlev
//                (LDS inside the tile, L2 across its edge).
// Tiles are dealt to the XCDs in contiguous ranges (workgroups go round-robin over the eight XCDs), so a tile's edge
// columns are usually in the L2 that already holds its neighbour.
// ---------------------------------------------------------------------------------------------------------
#ifndef FPX_VT_TX
#define FPX_VT_TX 64
#endif
#ifndef FPX_VT_WAVES
#define FPX_VT_WAVES 8      // waves per tile in k_vt_fused: each walks nz / waves z levels
#endif
#ifndef FPX_VT_BLOCKS
#define FPX_VT_BLOCKS 1     // register budget: resident tiles per CU the compiler must allow
#endif
constexpr int kVtTx = FPX_VT_TX, kVtTy = 64 / FPX_VT_TX, kVtCols = kVtTx * kVtTy;

struct Tiles {
  int tiles_y, ntiles, tiles_per_xcd;
  __device__ __forceinline__ bool origin(int &x0, int &y0) const {
    const int tile = (int)(blockIdx.x & 7) * tiles_per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= ntiles) return false;
    x0 = (tile / tiles_y) * kVtTx;           // consecutive tiles of an XCD are neighbours in y: a tile's rows jy-1, jy+1
    y0 = (tile % tiles_y) * kVtTy;           // are the rows its predecessor and successor bring into that L2
    return true;
  }
};

template <typename H>
__global__ void __launch_bounds__(512) k_vt_levels(Geo<H> G, In<H> I, Out<H> O, Tiles T) {
#pragma clang fp contract(off)
  extern __shared__ unsigned char vt_smem[];
  H *U = (H *)vt_smem;                               // [nuvz][64]
  int x0, y0;
  if (!T.origin(x0, y0)) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int ix = x0 + (lane & (kVtTx - 1)), jy = y0 + lane / kVtTx;
  const bool on = ix < G.nx && jy < G.ny;
  const int nuvz = G.nuvz;
  const H r_air = VK(287.05), konst = VK(287.05) / VK(9.81);
  if (on) {
    const H p = I.ps[G.at2(ix, jy)];
    const int per = (nuvz + nw - 1) / nw, ka = 1 + w * per, kb = min(nuvz, ka + per - 1);
    H tvold = VK(0.), pold = VK(0.);
    if (ka <= kb) {
      if (ka <= 2) {
        tvold = I.tt2[G.at2(ix, jy)] * (VK(1.) + VK(0.378) * ew<H>(I.td2[G.at2(ix, jy)]) / p);
        pold = p;
        if (w == 0) O.rhoh[G.at2(ix, jy)] = p / (r_air * tvold);      // rhoh(1) (:214), read by k_vt_fused
      } else {
        const size_t l = G.at(ix, jy, ka - 1);
        tvold = I.tth[l] * (VK(1.) + VK(0.608) * I.qvh[l]);
        pold = I.akz[ka - 2] + I.bkz[ka - 2] * p;
      }
    }
    for (int kz = ka; kz <= kb; kz++) {
      if (kz == 1) { U[lane] = VK(0.); continue; }
      const size_t o = G.at(ix, jy, kz);
      const H pint = I.akz[kz - 1] + I.bkz[kz - 1] * p;
      const H tv = I.tth[o] * (VK(1.) + VK(0.608) * I.qvh[o]);
      const H dtv = tv - tvold;
      H inc;
      if ((dtv < 0 ? -dtv : dtv) > VK(0.2)) inc = konst * M<H>::log(pold / pint) * (tv - tvold) / M<H>::log(tv / tvold);
      else inc = konst * M<H>::log(pold / pint) * tv;
      U[(kz - 1) * kVtCols + lane] = inc;
      tvold = tv; pold = pint;
    }
  }
  __syncthreads();
  if (w == 0 && on) {
    H acc = VK(0.);
    for (int kz = 2; kz <= nuvz; kz++) {
      acc = acc + U[(kz - 1) * kVtCols + lane];
      U[(kz - 1) * kVtCols + lane] = acc;
    }
  }
  __syncthreads();
  if (on)
    for (int kz = 1 + w; kz <= nuvz; kz += nw) O.uvzlev[G.at(ix, jy, kz)] = U[(kz - 1) * kVtCols + lane];
}

template <typename H, int NW>
__global__ void __launch_bounds__(NW * 64, FPX_VT_BLOCKS) k_vt_fused(Geo<H> G, In<H> I, Out<H> O, Tiles T) {
#pragma clang fp contract(off)
  extern __shared__ unsigned char vt_smem[];
  const int nz = G.nz;                               // = nuvz = nwz on this path
  H *U = (H *)vt_smem;                               // uvzlev [nz][64]
  H *hgt = U + (size_t)nz * kVtCols, *akz = hgt + nz, *bkz = akz + nz, *akn = bkz + nz, *bkn = akn + nz;
  int x0, y0;
  if (!T.origin(x0, y0)) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lx = lane & (kVtTx - 1), ly = lane / kVtTx;
  const int ix = x0 + lx, jy = y0 + ly;
  const bool on = ix < G.nx && jy < G.ny;
  for (int k = threadIdx.x; k < nz; k += NW * 64) {
    hgt[k] = I.height[k]; akz[k] = I.akz[k]; bkz[k] = I.bkz[k]; akn[k] = I.aknew[k]; bkn[k] = I.bknew[k];
  }
  if (on) {                                          // all loads of a wave in flight before the first LDS store
    constexpr int kBatch = 12;
    for (int k0 = 1 + w; k0 <= nz; k0 += NW * kBatch) {
      H tmp[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; j++) { const int kz = k0 + j * NW; tmp[j] = O.uvzlev[G.at(ix, jy, kz <= nz ? kz : nz)]; }
#pragma unroll
      for (int j = 0; j < kBatch; j++) { const int kz = k0 + j * NW; if (kz <= nz) U[(kz - 1) * kVtCols + lane] = tmp[j]; }
    }
  }
  __syncthreads();
  const int per = (nz + NW - 1) / NW, za = 1 + w * per, zb = min(nz, za + per - 1);
  if (!on || za > zb) return;

  auto Uc = [&](int k) -> H { return U[(k - 1) * kVtCols + lane]; };           // own column, level k (1-based)
  // element indices in 32 bits (the host sends larger grids down the unfused chain)
  const unsigned col = (unsigned)ix + (unsigned)G.nxmax * (unsigned)jy, plane = (unsigned)G.nxmax * (unsigned)G.nymax;
  auto At = [&](int k) -> unsigned { return col + plane * (unsigned)(k - 1); };
  const H r_air = VK(287.05);
  const H p = I.ps[col];
  const H top = Uc(nz);
  // wzlev (:240-244) and pinmconv (:248-258) from uvzlev, as k_vt_column forms them
  const H wztop = ((nz - 1 >= 2 ? (Uc(nz) + Uc(nz - 1)) / VK(2.) : VK(0.)) + Uc(nz)) - Uc(nz - 1);
  auto Wz = [&](int k) -> H { return k == 1 ? VK(0.) : k == nz ? wztop : (Uc(k + 1) + Uc(k)) / VK(2.); };
  auto Pin = [&](int k) -> H {
    if (k == 1) return Uc(2) / ((akn[1] + bkn[1] * p) - (akn[0] + bkn[0] * p));
    if (k == nz) return (Uc(nz) - Uc(nz - 1)) / ((akn[nz - 1] + bkn[nz - 1] * p) - (akn[nz - 2] + bkn[nz - 2] * p));
    return (Uc(k + 1) - Uc(k - 1)) / ((akn[k] + bkn[k] * p) - (akn[k - 2] + bkn[k - 2] * p));
  };
  // rhoh of an eta level from T and q there (k_vt_inc); level 1 (2 m values, ew) comes from k_vt_levels
  const H rhoh1 = O.rhoh[col];
  auto Rhoh = [&](int k, H t, H q) -> H {
    const H r = (akz[k - 1] + bkz[k - 1] * p) / (r_air * (t * (VK(1.) + VK(0.608) * q)));
    return k == 1 ? rhoh1 : r;
  };
  // the running level index after the sweep has passed z level izq: the sweeps only move forward and stand still
  // where the level lies above the column (the index keeps the value of the last level that was found)
  auto locate = [&](int izq, bool wsweep) -> int {
    const H lim = wsweep ? wztop : top;
    if (izq < 2 || hgt[1] > lim) return 2;
    int lo = 2, hi = izq;                            // largest level in [2, izq] not above lim
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (hgt[mid - 1] > lim) hi = mid - 1; else lo = mid; }
    const H h = hgt[lo - 1];
    int a = 2, b = nz;                               // smallest kz with h <= level(kz)
    while (a < b) { const int mid = (a + b) >> 1; if (h <= (wsweep ? Wz(mid) : Uc(mid))) b = mid; else a = mid + 1; }
    return a;
  };
  auto sweep_uv = [&](int idx, H h) -> int {         // :294-312
    if (h > top) return idx;
    for (int k = idx; ; k++) {
      if (h > Uc(k - 1) && h <= Uc(k)) return k;
      if (k == nz) return idx;
    }
  };
  auto sweep_w = [&](int idx, H h) -> int {          // :366-378
    for (int k = idx; ; k++) {
      if (h > Wz(k - 1) && h <= Wz(k)) return k;
      if (k == nz) return idx;
    }
  };
  // Consecutive z levels use the same or the next pair of eta levels: the values of the last pair stay in registers
  // (without this every input element is fetched two to three times -- measured 1.34 GB of reads for 0.43 GB of input);
  // what is missing is loaded in one batch
  struct Lev { int k; H t, q, u, v, pv, rh; };       // eta level k: T, q, u, v, pv and rhoh there
  struct LevW { int k; H wp; };                      // wwh(k) * pinmconv(k)
  Lev cl{0, VK(0.), VK(0.), VK(0.), VK(0.), VK(0.), VK(0.)}, cu = cl;
  LevW wl{0, VK(0.)}, wu = wl;
  auto rh_at = [&](int k) -> H {
    if (k == cu.k) return cu.rh;
    if (k == cl.k) return cl.rh;
    const unsigned a = At(k);
    return Rhoh(k, I.tth[a], I.qvh[a]);
  };
  // rho alone on z level iz (for the first and last drhodz of this wave's range)
  auto rho_only = [&](int iz, int idx) -> H {
    const H h = hgt[iz - 1];
    if (iz == 1) return rh_at(1);
    if (iz == nz || h > top) return rh_at(nz);
    const H dz1 = h - Uc(idx - 1), dz2 = Uc(idx) - h, dz = dz1 + dz2;
    return (rh_at(idx - 1) * dz2 + rh_at(idx) * dz1) / dz;
  };

  const bool inner = ix >= 1 && ix <= G.nx - 2 && jy >= 1 && jy <= G.ny - 2;
  const H pi180 = VK(3.14159265) / VK(180.);
  const H cosf = VK(1.) / M<H>::cos(((H)jy * G.dy + G.ylat0) * pi180);
  // a neighbour column's uvzlev: LDS inside the tile (the neighbour exists in the grid: `inner`), global across its edge
  auto Un = [&](int dx_, int dy_, int k) -> H {
    const int nx_ = lx + dx_, ny_ = ly + dy_;
    if (nx_ >= 0 && nx_ < kVtTx && ny_ >= 0 && ny_ < kVtTy) return U[(k - 1) * kVtCols + lane + dx_ + dy_ * kVtTx];
    return O.uvzlev[At(k) + dx_ + dy_ * G.nxmax];
  };

  int idx = locate(za - 1, false), idxw = locate(za - 1, true);
  H r_m1 = za > 1 ? rho_only(za - 1, idx) : VK(0.), r_m2 = VK(0.);
  for (int iz = za; iz <= zb; iz++) {
    const unsigned o = At(iz);
    const H h = hgt[iz - 1];
    H uu, vv, rho;
    if (iz >= 2 && iz <= nz - 1) idx = sweep_uv(idx, h);
    if (iz >= 2) idxw = sweep_w(idxw, h);
    const bool copy = iz == 1 || iz == nz || h > top;        // level 1, or the top eta level (also above the column top)
    const int ka = copy ? (iz == 1 ? 1 : nz) : idx - 1, kb = copy ? ka : idx;
    const int wa = iz == 1 ? 1 : idxw - 1, wb = iz == 1 ? 1 : idxw;
    // every load this level needs is issued before the first one is used
    const bool la = ka != cu.k && ka != cl.k, lb = kb != cu.k && kb != cl.k && kb != ka;
    const bool lwa = wa != wu.k && wa != wl.k, lwb = wb != wu.k && wb != wl.k && wb != wa;
    Lev a{ka, VK(0.), VK(0.), VK(0.), VK(0.), VK(0.), VK(0.)}, b = a;
    b.k = kb;
    H wwa = VK(0.), wwb = VK(0.);
    if (la) { const unsigned g = At(ka); a.t = I.tth[g]; a.q = I.qvh[g]; a.u = I.uuh[g]; a.v = I.vvh[g]; a.pv = I.pvh[g]; }
    if (lb) { const unsigned g = At(kb); b.t = I.tth[g]; b.q = I.qvh[g]; b.u = I.uuh[g]; b.v = I.vvh[g]; b.pv = I.pvh[g]; }
    if (lwa) wwa = I.wwh[At(wa)];
    if (lwb) wwb = I.wwh[At(wb)];
    if (la) a.rh = Rhoh(ka, a.t, a.q); else a = ka == cu.k ? cu : cl;
    if (lb) b.rh = Rhoh(kb, b.t, b.q); else b = kb == ka ? a : kb == cu.k ? cu : cl;
    LevW xa{wa, VK(0.)}, xb{wb, VK(0.)};
    if (lwa) xa.wp = wwa * Pin(wa); else xa = wa == wu.k ? wu : wl;
    if (lwb) xb.wp = wwb * Pin(wb); else xb = wb == wa ? xa : wb == wu.k ? wu : wl;
    cl = a; cu = b; wl = xa; wu = xb;
    if (copy) {
      uu = a.u; vv = a.v; rho = a.rh;
      O.uu[o] = uu; O.vv[o] = vv; O.tt[o] = a.t; O.qv[o] = a.q; O.pv[o] = a.pv; O.rho[o] = rho;
    } else {
      const H dz1 = h - Uc(idx - 1), dz2 = Uc(idx) - h, dz = dz1 + dz2;
      uu = (a.u * dz2 + b.u * dz1) / dz;
      vv = (a.v * dz2 + b.v * dz1) / dz;
      rho = (a.rh * dz2 + b.rh * dz1) / dz;
      O.uu[o] = uu; O.vv[o] = vv;
      O.tt[o] = (a.t * dz2 + b.t * dz1) / dz;
      O.qv[o] = (a.q * dz2 + b.q * dz1) / dz;
      O.pv[o] = (a.pv * dz2 + b.pv * dz1) / dz;
      O.rho[o] = rho;
    }
    H ww;
    if (iz == 1) {
      ww = xa.wp;
    } else {
      const H dz1 = h - Wz(idxw - 1), dz2 = Wz(idxw) - h, dz = dz1 + dz2;
      ww = (xa.wp * dz2 + xb.wp * dz1) / dz;
    }
    if (inner && iz >= 2 && iz <= nz - 1) {          // :411-453, with the (possibly standing) index of the u/v sweep
      const int kz = idx;
      const H dz1 = h - Uc(kz - 1), dz2 = Uc(kz) - h, dz = dz1 + dz2;
      const H dzdx1 = (Un(1, 0, kz - 1) - Un(-1, 0, kz - 1)) / VK(2.);
      const H dzdx2 = (Un(1, 0, kz) - Un(-1, 0, kz)) / VK(2.);
      const H dzdx = (dzdx1 * dz2 + dzdx2 * dz1) / dz;
      const H dzdy1 = (Un(0, 1, kz - 1) - Un(0, -1, kz - 1)) / VK(2.);
      const H dzdy2 = (Un(0, 1, kz) - Un(0, -1, kz)) / VK(2.);
      const H dzdy = (dzdy1 * dz2 + dzdy2 * dz1) / dz;
      ww = ww + (dzdx * uu * G.dxconst * G.xres * cosf + dzdy * vv * G.dyconst * G.yres);
    }
    O.ww[o] = ww;
    if (iz - 1 >= za) {                              // drhodz of the level below (:394-400), now that rho(iz) is known
      const int k = iz - 1;
      const H d = k == 1 ? (rho - r_m1) / (hgt[1] - hgt[0]) : (rho - r_m2) / (hgt[k] - hgt[k - 2]);
      O.drhodz[At(k)] = d;
      if (k == nz - 1) O.drhodz[At(nz)] = d;
    }
    r_m2 = r_m1; r_m1 = rho;
  }
  if (zb < nz) {
    const int k = zb;
    const H hn = hgt[k];
    const int idn = (k + 1 >= 2 && k + 1 <= nz - 1) ? sweep_uv(idx, hn) : idx;
    const H rp = rho_only(k + 1, idn);
    const H d = k == 1 ? (rp - r_m1) / (hgt[1] - hgt[0]) : (rp - r_m2) / (hgt[k] - hgt[k - 2]);
    O.drhodz[At(k)] = d;
    if (k == nz - 1) O.drhodz[At(nz)] = d;
  }
}

// polar-stereographic winds on the rows of a polar cap, :459-470 / :530-541
template <typename H>
__global__ void __launch_bounds__(256) k_vt_polar(Geo<H> G, Out<H> O, int jy0, int jy1, int south) {
#pragma clang fp contract(off)
  const int ix = blockIdx.x * blockDim.x + threadIdx.x;
  const int jy = jy0 + (int)blockIdx.y, iz = 1 + (int)blockIdx.z;
  if (ix >= G.nx || jy > jy1 || jy < 0 || jy >= G.ny) return;
  const H ylat = G.ylat0 + (H)jy * G.dy, xlon = G.xlon0 + (H)ix * G.dx;
  const size_t o = G.at(ix, jy, iz);
  H ug, vg;
  cc2gll<H>(south ? G.southpolemap : G.northpolemap, ylat, xlon, O.uu[o], O.vv[o], ug, vg);
  O.uupol[o] = ug;
  O.vvpol[o] = vg;
}

// the pole row itself: wind from the central grid point (:473-505 / :544-580, including the
// reference's use of northpolemap for the south pole's auxiliary point, :576) and w = zonal mean
// of the next parallel summed in ix order (:508-520 / :583-597).  One wave per level: the row is
// staged in LDS with coalesced loads, one lane adds it up in the reference's order, all lanes store.
template <typename H>
__global__ void __launch_bounds__(64) k_vt_polerow(Geo<H> G, Out<H> O, int south) {
#pragma clang fp contract(off)
  extern __shared__ unsigned char vt_smem[];
  H *row = (H *)vt_smem;               // [nx] + 3 results
  const int iz = 1 + blockIdx.x, lane = threadIdx.x;
  const H pi = VK(3.14159265);
  const int jpole = south ? 0 : G.ny - 1, jnext = south ? 1 : G.ny - 2, ic = G.nx / 2 - 1;
  for (int ix = lane; ix < G.nx; ix += 64) row[ix] = O.ww[G.at(ix, jnext, iz)];
  __syncthreads();
  if (lane == 0) {
    H xlon = G.xlon0 + (H)ic * G.dx;
    H xlonr = xlon * pi / VK(180.);
    const H ucen = O.uu[G.at(ic, jpole, iz)], vcen = O.vv[G.at(ic, jpole, iz)];
    const H ffpol = M<H>::sqrt(ucen * ucen + vcen * vcen);
    H ddpol;
    if (!south) {
      if (vcen < VK(0.)) ddpol = M<H>::atan(ucen / vcen) - xlonr;
      else if (vcen > VK(0.)) ddpol = pi + M<H>::atan(ucen / vcen) - xlonr;
      else ddpol = pi / VK(2.) - xlonr;
    } else {
      if (vcen < VK(0.)) ddpol = M<H>::atan(ucen / vcen) + xlonr;
      else if (vcen > VK(0.)) ddpol = pi + M<H>::atan(ucen / vcen) + xlonr;
      else ddpol = pi / VK(2.) - xlonr;
    }
    if (ddpol < VK(0.)) ddpol = VK(2.0) * pi + ddpol;
    if (ddpol > VK(2.0) * pi) ddpol = ddpol - VK(2.0) * pi;
    xlon = VK(180.0);
    xlonr = xlon * pi / VK(180.);
    H uuaux, vvaux, up, vp;
    if (!south) { uuaux = -ffpol * M<H>::sin(xlonr + ddpol); vvaux = -ffpol * M<H>::cos(xlonr + ddpol); }
    else { uuaux = +ffpol * M<H>::sin(xlonr - ddpol); vvaux = -ffpol * M<H>::cos(xlonr - ddpol); }
    cc2gll<H>(G.northpolemap, south ? VK(-90.0) : VK(90.0), xlon, uuaux, vvaux, up, vp);
    H wdummy = VK(0.);
    for (int ix = 0; ix < G.nx; ix++) wdummy = wdummy + row[ix];
    wdummy = wdummy / (H)G.nx;
    row[G.nx] = up; row[G.nx + 1] = vp; row[G.nx + 2] = wdummy;
  }
  __syncthreads();
  const H up = row[G.nx], vp = row[G.nx + 1], wd = row[G.nx + 2];
  for (int ix = lane; ix < G.nx; ix += 64) {
    const size_t o = G.at(ix, jpole, iz);
    O.uupol[o] = up;
    O.vvpol[o] = vp;
    O.ww[o] = wd;
  }
}

#undef VK

}  // namespace vt
FPX_TU_CLOSE
}  // namespace fpx
