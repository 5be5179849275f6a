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
#include <hip/hip_runtime.h>

namespace fpx {
namespace vt {

template <typename H> struct M;
template <> struct M<float> {
  static __device__ __forceinline__ float log(float x) { return ::logf(x); }
  static __device__ __forceinline__ float pow(float x, float y) { return ::powf(x, y); }
  static __device__ __forceinline__ float cos(float x) { return ::cosf(x); }
  static __device__ __forceinline__ float sin(float x) { return ::sinf(x); }
  static __device__ __forceinline__ float atan(float x) { return ::atanf(x); }
  static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
  static __device__ __forceinline__ float fmod(float x, float y) { return ::fmodf(x, y); }
};
template <> struct M<double> {
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
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
}  // namespace fpx
