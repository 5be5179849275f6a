// Device side of fpx_verttransform_ecmwf: the reference's verttransform_ecmwf
// (src/verttransform_ecmwf.f90:118-590 -- eta levels -> terrain-following z levels, rho,
// drhodz, w conversion and eta-slope correction, polar-stereographic winds) as HIP kernels
// for gfx950.  SURVEY.md section 8 (f) item 1: the grid-sized work that otherwise serialises
// the host once per wind interval, and the 722 MB field upload it makes unnecessary.
//
// Design: the transform is column-independent except for the eta-slope stencil and the polar
// rows, so one lane owns one (ix,jy) column and sweeps it upwards with the running level
// index the reference keeps in idx(ix,jy); consecutive lanes own consecutive ix, and every
// array keeps the host's (ix,jy,level) x-fastest layout (strides nxmax, nymax), so each
// sweep step is one coalesced row access per array.  HBM-bound: about 35 array passes.
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

// heights of the eta levels, air density on them, wzlev and pinmconv: verttransform_ecmwf.f90:203-258
template <typename H>
__global__ void __launch_bounds__(256) k_vt_levels(Geo<H> G, In<H> I, Out<H> O) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.nx * G.ny) return;
  const int ix = t % G.nx, jy = t / G.nx;
  const H r_air = VK(287.05), konst = VK(287.05) / VK(9.81);
  const H p = I.ps[G.at2(ix, jy)];
  H tvold = I.tt2[G.at2(ix, jy)] * (VK(1.) + VK(0.378) * ew<H>(I.td2[G.at2(ix, jy)]) / p);
  H pold = p;
  H uvz = VK(0.);
  O.uvzlev[G.at(ix, jy, 1)] = VK(0.);
  O.wzlev[G.at(ix, jy, 1)] = VK(0.);
  O.rhoh[G.at(ix, jy, 1)] = pold / (r_air * tvold);
  for (int kz = 2; kz <= G.nuvz; kz++) {
    const H pint = I.akz[kz - 1] + I.bkz[kz - 1] * p;
    const H tv = I.tth[G.at(ix, jy, kz)] * (VK(1.) + VK(0.608) * I.qvh[G.at(ix, jy, kz)]);
    O.rhoh[G.at(ix, jy, kz)] = pint / (r_air * tv);
    const H dtv = tv - tvold;
    if ((dtv < 0 ? -dtv : dtv) > VK(0.2)) uvz = uvz + konst * M<H>::log(pold / pint) * (tv - tvold) / M<H>::log(tv / tvold);
    else uvz = uvz + konst * M<H>::log(pold / pint) * tv;
    O.uvzlev[G.at(ix, jy, kz)] = uvz;
    tvold = tv;
    pold = pint;
  }
  // wzlev :240-244 (sliding window over the column just written)
  {
    H lo = O.uvzlev[G.at(ix, jy, 2)], wprev = VK(0.);
    for (int kz = 2; kz <= G.nwz - 1; kz++) {
      const H hi = O.uvzlev[G.at(ix, jy, kz + 1)];
      wprev = (hi + lo) / VK(2.);
      O.wzlev[G.at(ix, jy, kz)] = wprev;
      lo = hi;
    }
    O.wzlev[G.at(ix, jy, G.nwz)] = wprev + O.uvzlev[G.at(ix, jy, G.nuvz)] - O.uvzlev[G.at(ix, jy, G.nuvz - 1)];
  }
  // pinmconv=(h2-h1)/(p2-p1) :248-258
  {
    const int nz = G.nz;
    O.pinmconv[G.at(ix, jy, 1)] = (O.uvzlev[G.at(ix, jy, 2)]) / ((I.aknew[1] + I.bknew[1] * p) - (I.aknew[0] + I.bknew[0] * p));
    for (int kz = 2; kz <= nz - 1; kz++)
      O.pinmconv[G.at(ix, jy, kz)] = (O.uvzlev[G.at(ix, jy, kz + 1)] - O.uvzlev[G.at(ix, jy, kz - 1)]) /
                                     ((I.aknew[kz] + I.bknew[kz] * p) - (I.aknew[kz - 2] + I.bknew[kz - 2] * p));
    O.pinmconv[G.at(ix, jy, nz)] = (O.uvzlev[G.at(ix, jy, nz)] - O.uvzlev[G.at(ix, jy, nz - 1)]) /
                                   ((I.aknew[nz - 1] + I.bknew[nz - 1] * p) - (I.aknew[nz - 2] + I.bknew[nz - 2] * p));
  }
}

// the three upward sweeps of one column: u,v,T,q,pv,rho (:264-356), w (:362-389), drhodz (:394-400)
template <typename H>
__global__ void __launch_bounds__(256) k_vt_interp(Geo<H> G, In<H> I, Out<H> O) {
#pragma clang fp contract(off)
  extern __shared__ unsigned char vt_smem[];
  H *hgt = (H *)vt_smem;
  for (int k = threadIdx.x; k < G.nz; k += blockDim.x) hgt[k] = I.height[k];
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.nx * G.ny) return;
  const int ix = t % G.nx, jy = t / G.nx;
  const int nz = G.nz, nuvz = G.nuvz, nwz = G.nwz;
  {
    const size_t b = G.at(ix, jy, 1), e = G.at(ix, jy, nz), eh = G.at(ix, jy, nuvz);
    O.uu[b] = I.uuh[b]; O.vv[b] = I.vvh[b]; O.tt[b] = I.tth[b]; O.qv[b] = I.qvh[b]; O.pv[b] = I.pvh[b]; O.rho[b] = O.rhoh[b];
    O.uu[e] = I.uuh[eh]; O.vv[e] = I.vvh[eh]; O.tt[e] = I.tth[eh]; O.qv[e] = I.qvh[eh]; O.pv[e] = I.pvh[eh]; O.rho[e] = O.rhoh[eh];
  }
  const H top = O.uvzlev[G.at(ix, jy, nuvz)];
  {
    const size_t eh = G.at(ix, jy, nuvz);
    const H utop = I.uuh[eh], vtop = I.vvh[eh], ttop = I.tth[eh], qtop = I.qvh[eh], ptop = I.pvh[eh], rtop = O.rhoh[eh];
    int idx = 2;
    H zlo = O.uvzlev[G.at(ix, jy, 1)], zhi = O.uvzlev[G.at(ix, jy, 2)];   // uvzlev(idx-1), uvzlev(idx)
    for (int iz = 2; iz <= nz - 1; iz++) {
      const H h = hgt[iz - 1];
      const size_t o = G.at(ix, jy, iz);
      if (h > top) {
        O.uu[o] = utop; O.vv[o] = vtop; O.tt[o] = ttop; O.qv[o] = qtop; O.pv[o] = ptop; O.rho[o] = rtop;
        continue;
      }
      // innuvz: first kz >= idx with uvzlev(kz-1) < h <= uvzlev(kz); idx unchanged when none
      {
        H a = zlo, b = zhi;
        for (int kz = idx; kz <= nuvz; kz++) {
          if (h > a && h <= b) { idx = kz; zlo = a; zhi = b; break; }
          if (kz == nuvz) break;
          a = b;
          b = O.uvzlev[G.at(ix, jy, kz + 1)];
        }
      }
      const int kz = idx;
      const H dz1 = h - zlo, dz2 = zhi - h, dz = dz1 + dz2;
      const size_t l = G.at(ix, jy, kz - 1), u = G.at(ix, jy, kz);
      O.uu[o] = (I.uuh[l] * dz2 + I.uuh[u] * dz1) / dz;
      O.vv[o] = (I.vvh[l] * dz2 + I.vvh[u] * dz1) / dz;
      O.tt[o] = (I.tth[l] * dz2 + I.tth[u] * dz1) / dz;
      O.qv[o] = (I.qvh[l] * dz2 + I.qvh[u] * dz1) / dz;
      O.pv[o] = (I.pvh[l] * dz2 + I.pvh[u] * dz1) / dz;
      O.rho[o] = (O.rhoh[l] * dz2 + O.rhoh[u] * dz1) / dz;
    }
  }
  // w: pressure velocity -> m/s on the z levels
  {
    O.ww[G.at(ix, jy, 1)] = I.wwh[G.at(ix, jy, 1)] * O.pinmconv[G.at(ix, jy, 1)];
    O.ww[G.at(ix, jy, nz)] = I.wwh[G.at(ix, jy, nwz)] * O.pinmconv[G.at(ix, jy, nz)];
    int idx = 2;
    H zlo = O.wzlev[G.at(ix, jy, 1)], zhi = O.wzlev[G.at(ix, jy, 2)];
    for (int iz = 2; iz <= nz; iz++) {
      const H h = hgt[iz - 1];
      {
        H a = zlo, b = zhi;
        for (int kz = idx; kz <= nwz; kz++) {
          if (h > a && h <= b) { idx = kz; zlo = a; zhi = b; break; }
          if (kz == nwz) break;
          a = b;
          b = O.wzlev[G.at(ix, jy, kz + 1)];
        }
      }
      const int kz = idx;
      const H dz1 = h - zlo, dz2 = zhi - h, dz = dz1 + dz2;
      const size_t l = G.at(ix, jy, kz - 1), u = G.at(ix, jy, kz);
      O.ww[G.at(ix, jy, iz)] = (I.wwh[l] * O.pinmconv[l] * dz2 + I.wwh[u] * O.pinmconv[u] * dz1) / dz;
    }
  }
  // density gradient (the column's own rho, just written)
  {
    H rm = O.rho[G.at(ix, jy, 1)], r0 = O.rho[G.at(ix, jy, 2)];
    H last = (r0 - rm) / (hgt[1] - hgt[0]);
    O.drhodz[G.at(ix, jy, 1)] = last;
    for (int kz = 2; kz <= nz - 1; kz++) {
      const H rp = O.rho[G.at(ix, jy, kz + 1)];
      last = (rp - rm) / (hgt[kz] - hgt[kz - 2]);
      O.drhodz[G.at(ix, jy, kz)] = last;
      rm = r0;
      r0 = rp;
    }
    O.drhodz[G.at(ix, jy, nz)] = last;
  }
}

// slope of the eta levels in windward direction and resulting correction of w, :411-453
// (interior columns; reads the neighbours' uvzlev, so it runs after k_vt_levels has finished)
template <typename H>
__global__ void __launch_bounds__(256) k_vt_slope(Geo<H> G, In<H> I, Out<H> O) {
#pragma clang fp contract(off)
  extern __shared__ unsigned char vt_smem[];
  H *hgt = (H *)vt_smem;
  for (int k = threadIdx.x; k < G.nz; k += blockDim.x) hgt[k] = I.height[k];
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int wx = G.nx - 2, wy = G.ny - 2;
  if (wx <= 0 || wy <= 0 || t >= wx * wy) return;
  const int ix = 1 + t % wx, jy = 1 + t / wx;
  const int nz = G.nz;
  const H pi180 = VK(3.14159265) / VK(180.);
  const H cosf = VK(1.) / M<H>::cos(((H)jy * G.dy + G.ylat0) * pi180);
  int idx = 2;
  H zlo = O.uvzlev[G.at(ix, jy, 1)], zhi = O.uvzlev[G.at(ix, jy, 2)];
  for (int iz = 2; iz <= nz - 1; iz++) {
    const H h = hgt[iz - 1];
    {
      H a = zlo, b = zhi;
      for (int kz = idx; kz <= nz; kz++) {   // bounded by nz, not nuvz, as in the reference (:418)
        if (h > a && h <= b) { idx = kz; zlo = a; zhi = b; break; }
        if (kz == nz) break;
        a = b;
        b = O.uvzlev[G.at(ix, jy, kz + 1)];
      }
    }
    const int kz = idx;
    const H dz1 = h - zlo, dz2 = zhi - h, dz = dz1 + dz2;
    const H dzdx1 = (O.uvzlev[G.at(ix + 1, jy, kz - 1)] - O.uvzlev[G.at(ix - 1, jy, kz - 1)]) / VK(2.);
    const H dzdx2 = (O.uvzlev[G.at(ix + 1, jy, kz)] - O.uvzlev[G.at(ix - 1, jy, kz)]) / VK(2.);
    const H dzdx = (dzdx1 * dz2 + dzdx2 * dz1) / dz;
    const H dzdy1 = (O.uvzlev[G.at(ix, jy + 1, kz - 1)] - O.uvzlev[G.at(ix, jy - 1, kz - 1)]) / VK(2.);
    const H dzdy2 = (O.uvzlev[G.at(ix, jy + 1, kz)] - O.uvzlev[G.at(ix, jy - 1, kz)]) / VK(2.);
    const H dzdy = (dzdy1 * dz2 + dzdy2 * dz1) / dz;
    const size_t o = G.at(ix, jy, iz);
    O.ww[o] = O.ww[o] + (dzdx * O.uu[o] * G.dxconst * cosf + dzdy * O.vv[o] * G.dyconst);
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
// of the next parallel summed in ix order (:508-520 / :583-597).  One lane per level.
template <typename H>
__global__ void k_vt_polerow(Geo<H> G, Out<H> O, int south) {
#pragma clang fp contract(off)
  const int iz = 1 + blockIdx.x * blockDim.x + threadIdx.x;
  if (iz > G.nz) return;
  const H pi = VK(3.14159265);
  const int jpole = south ? 0 : G.ny - 1, jnext = south ? 1 : G.ny - 2, ic = G.nx / 2 - 1;
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
  for (int ix = 0; ix < G.nx; ix++) wdummy = wdummy + O.ww[G.at(ix, jnext, iz)];
  wdummy = wdummy / (H)G.nx;
  for (int ix = 0; ix < G.nx; ix++) {
    const size_t o = G.at(ix, jpole, iz);
    O.uupol[o] = up;
    O.vvpol[o] = vp;
    O.ww[o] = wdummy;
  }
}

#undef VK

}  // namespace vt
}  // namespace fpx
