// Device side of fpx_calcpar: the boundary-layer parameters of the reference's calcpar (src/calcpar.f90:76-265,
// ECMWF branch) -- friction velocity (scalev.f90), inverse Obukhov length (obukhov.f90), mixing height and convective
// velocity scale (richardson.f90, with qvsat.f90 and ew.f90) and the thermal tropopause (calcpar.f90:199-265) -- from the
// model-level arrays fpx_verttransform_ecmwf has left on the device.  SURVEY.md section 8 (f) item 1, second half: after
// it no 2-D meteorological field but the raw surface analysis crosses PCIe.
// One lane per grid column (the level loops of a column are sequential: running hydrostatic sums, first-crossing
// searches); consecutive lanes own consecutive ix, so every level access of a wave is one contiguous row segment.
// Arithmetic in the host's real kind H with FMA contraction off; log, exp, x**y come from the device's libm.
// Not computed: the dry-deposition velocities (getvdep, calcpar.f90:174-193: land-use tables stay with the host, vdep is
// an input) and the potential vorticity (calcpv, :270).
#pragma once
#include "fpx_tu.hpp"
#include <hip/hip_runtime.h>
#include "fpx_verttransform.hpp"

namespace fpx {
FPX_TU_OPEN
namespace cp {

#define CK(x) ((H)(x))

template <typename H> __device__ __forceinline__ H m_exp(H x);
template <> __device__ __forceinline__ float m_exp<float>(float x) { return ::expf(x); }
template <> __device__ __forceinline__ double m_exp<double>(double x) { return ::exp(x); }

// qvsat.f90: f_qvsat with f_esl / f_esi
template <typename H>
__device__ __forceinline__ H f_qvsat(H p, H t) {
#pragma clang fp contract(off)
  const H rddrv = CK(287.0) / CK(461.0);
  H fespt;
  if (t >= CK(253.15)) { const H f = CK(1.0007) + CK(3.46e-8) * p; fespt = f * CK(611.21) * m_exp<H>(CK(17.502) * (t - CK(273.15)) / (t - CK(32.18))); }
  else { const H f = CK(1.0003) + CK(4.18e-8) * p; fespt = f * CK(611.15) * m_exp<H>(CK(22.452) * (t - CK(273.15)) / (t - CK(0.6))); }
  if (p - (CK(1.0) - rddrv) * fespt == CK(0.)) return CK(1.);
  return rddrv * fespt / (p - (CK(1.0) - rddrv) * fespt);
}

template <typename H>
struct Args {
  vt::Geo<H> G;
  vt::In<H> I;                 // uuh, vvh, tth, qvh (nuvz levels), ps, tt2, td2, akz, bkz on the device, host layout
  const H *surfstr, *sshf, *excessoro, *akm, *bkm;
  int lsubgrid;
  H *zlev;                     // scratch [nuvz][nymax][nxmax]
  H *ustar, *wstar, *oli, *hmix, *tropopause;   // out, host layout (0:nxmax-1,0:nymax-1)
};

template <typename H>
__global__ void __launch_bounds__(256) k_calcpar(Args<H> A) {
#pragma clang fp contract(off)
  typedef vt::M<H> M;
  const vt::Geo<H> &G = A.G;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= G.nx * G.ny) return;
  const int ix = c % G.nx, jy = c / G.nx;
  const size_t i2 = G.at2(ix, jy);
  const int nuvz = G.nuvz;
  const H r_air = CK(287.05), ga = CK(9.81), cpa = CK(1004.6), karman = CK(0.40), convke = CK(2.0);
  const H konst = r_air / ga;
  auto L = [&](const H *f, int k) { return f[G.at(ix, jy, k)]; };      // level k (1-based) of this column
  const H ps = A.I.ps[i2], tt2 = A.I.tt2[i2], td2 = A.I.td2[i2], hf = A.sshf[i2];
  // calcpar.f90:76-100
  const H ylat = G.ylat0 + (H)jy * G.dy;
  H altmin;
  if (ylat >= CK(-20.) && ylat <= CK(20.)) altmin = CK(5000.);
  else if (ylat > CK(20.) && ylat < CK(40.)) altmin = CK(2500.) + (CK(40.) - ylat) * CK(125.);
  else if (ylat > CK(-40.) && ylat < CK(-20.)) altmin = CK(2500.) + (CK(40.) + ylat) * CK(125.);
  else altmin = CK(2500.);
  const H ew_td2 = vt::ew<H>(td2);
  const H tv2 = tt2 * (CK(1.) + CK(0.378) * ew_td2 / ps);
  // 1) scalev.f90
  H ust;
  { const H rhoa = ps / (r_air * tv2); ust = M::sqrt(fabs(A.surfstr[i2]) / rhoa); }
  if (ust <= CK(1.e-8)) ust = CK(1.e-8);
  // 2) obukhov.f90, ECMWF branch, with tth(ix,jy,2,n)
  H ol;
  {
    const H rhoa = ps / (r_air * tv2);
    const H plev = (A.akm[0] + A.akm[1]) / CK(2.) + (A.bkm[0] + A.bkm[1]) / CK(2.) * ps;
    const H theta = L(A.I.tth, 2) * M::pow(CK(100000.) / plev, r_air / cpa);
    const H thetastar = hf / (rhoa * cpa * ust);
    if (fabs(thetastar) > CK(1.e-10)) ol = theta * (ust * ust) / (karman * ga * thetastar); else ol = CK(9999);
    if (ol > CK(9999.)) ol = CK(9999.);
    if (ol < CK(-9999.)) ol = CK(-9999.);
  }
  A.ustar[i2] = ust;
  A.oli[i2] = ol != CK(0.) ? CK(1.) / ol : CK(99999.);
  // 3) richardson.f90, ECMWF branch
  H hm, wst, hmixplus;
  {
    const H ric = CK(0.25), b = CK(100.), bs = CK(8.5);
    const H u2 = L(A.I.uuh, 2), v2 = L(A.I.vvh, 2);
    H excess = CK(0.);
    for (int iter = 1;; iter++) {
      H pold = ps, tvold = tv2, zold = CK(2.0);
      const H zref = zold;
      const H thetaref = tvold * M::pow(CK(100000.) / pold, r_air / cpa) + excess;
      H thetaold = thetaref, z = CK(0.), theta = CK(0.);
      int k;
      for (k = 2; k <= nuvz; k++) {
        const H pint = A.I.akz[k - 1] + A.I.bkz[k - 1] * ps;
        const H tt = L(A.I.tth, k);
        const H tv = tt * (CK(1.) + CK(0.608) * L(A.I.qvh, k));
        if (fabs(tv - tvold) > CK(0.2)) z = zold + konst * M::log(pold / pint) * (tv - tvold) / M::log(tv / tvold);
        else z = zold + konst * M::log(pold / pint) * tv;
        theta = tv * M::pow(CK(100000.) / pint, r_air / cpa);
        const H du = L(A.I.uuh, k) - u2, dv = L(A.I.vvh, k) - v2;
        H den = du * du + dv * dv + b * (ust * ust);
        if (den < CK(0.1)) den = CK(0.1);
        const H ri = ga / thetaref * (theta - thetaref) * (z - zref) / den;
        if (ri > ric && thetaold < theta) break;
        tvold = tv; pold = pint; thetaold = theta; zold = z;
      }
      if (k > nuvz) k = nuvz;
      const H uk = L(A.I.uuh, k), uk1 = L(A.I.uuh, k - 1), vk = L(A.I.vvh, k), vk1 = L(A.I.vvh, k - 1);
      H zl = zold, ul = uk1, vl = vk1, zl1 = zold, theta1 = thetaold, zl2 = zold, theta2 = thetaold;
      for (int i = 1; i <= 20; i++) {
        const H fr = (H)i / CK(20.);
        zl = zold + fr * (z - zold);
        ul = uk1 + fr * (uk - uk1);
        vl = vk1 + fr * (vk - vk1);
        const H thetal = thetaold + fr * (theta - thetaold);
        H den = (ul - u2) * (ul - u2) + (vl - v2) * (vl - v2) + b * (ust * ust);
        if (den < CK(0.1)) den = CK(0.1);
        const H ril = ga / thetaref * (thetal - thetaref) * (zl - zref) / den;
        zl2 = zl; theta2 = thetal;
        if (ril > ric) break;
        zl1 = zl; theta1 = thetal;
      }
      hm = zl;
      const H thetam = CK(0.5) * (theta1 + theta2);
      const H wspeed = M::sqrt(ul * ul + vl * vl);
      const H bvfsq = (ga / thetam) * (theta2 - theta1) / (zl2 - zl1);
      if (bvfsq <= CK(0.)) hmixplus = CK(9999.); else hmixplus = wspeed / M::sqrt(bvfsq) * convke;
      if (hf < CK(0.)) {
        wst = M::pow(-hm * ga / thetaref * hf / cpa, CK(0.333));
        excess = -bs * hf / cpa / wst;
        if (iter < 3) continue;
      } else wst = CK(0.);
      break;
    }
  }
  H subsceff = CK(0.0);
  if (A.lsubgrid == 1) { subsceff = A.excessoro[i2]; if (hmixplus < subsceff) subsceff = hmixplus; }
  hm = hm + subsceff;
  if (hm < CK(100.)) hm = CK(100.);         // hmixmin, hmixmax: par_mod.f90:77
  if (hm > CK(4500.)) hm = CK(4500.);
  A.hmix[i2] = hm; A.wstar[i2] = wst;
  // thermal tropopause, calcpar.f90:199-265
  {
    H tvold = tv2, pold = ps, zold = CK(0.);
    A.zlev[G.at(ix, jy, 1)] = CK(0.);
    int kzmin = 1;
    bool have = false;
    for (int kz = 2; kz <= nuvz; kz++) {
      const H pint = A.I.akz[kz - 1] + A.I.bkz[kz - 1] * ps;
      const H tv = L(A.I.tth, kz) * (CK(1.) + CK(0.608) * L(A.I.qvh, kz));
      H z;
      if (fabs(tv - tvold) > CK(0.2)) z = zold + konst * M::log(pold / pint) * (tv - tvold) / M::log(tv / tvold);
      else z = zold + konst * M::log(pold / pint) * tv;
      A.zlev[G.at(ix, jy, kz)] = z;
      if (!have && z >= altmin) { kzmin = kz; have = true; }
      tvold = tv; pold = pint; zold = z;
    }
    bool found = false;
    for (int kz = kzmin; kz <= nuvz && !found; kz++) {
      const H zk = A.zlev[G.at(ix, jy, kz)], tk = L(A.I.tth, kz);
      for (int lz = kz + 1; lz <= nuvz; lz++) {
        const H zz = A.zlev[G.at(ix, jy, lz)];
        if (zz - zk > CK(2000.)) {
          if ((tk - L(A.I.tth, lz)) / (zz - zk) < CK(0.002)) { A.tropopause[i2] = zk; found = true; }
          break;
        }
      }
    }
  }
}

}  // namespace cp
FPX_TU_CLOSE
}  // namespace fpx
