/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's
 * verttransform_ecmwf (eta levels -> terrain-following z levels).
 * Only tests/ and the cpu_baseline / checker leg of bench.py and tools/bench_*.py may load this; the product path
 * (flexpart_amd/) never does.
 *
 * Plain C restatement of /root/reference/src/verttransform_ecmwf.f90:118-590 (height
 * initialisation, uvzlev/wzlev, pinmconv, vertical interpolation of u,v,T,q,pv,rho, w
 * conversion, drhodz, eta-slope correction of w, polar-stereographic winds) with
 * ew.f90 and cmapf_mod.f90:cc2gll/cspanf.  The cloud diagnostics (:604-880) and prs are
 * not restated (not read by the particle path).  Operation order follows the Fortran
 * so that the result can be pinned against the flang build of the unmodified routine
 * (oracle/_ref/vtref_r4|r8 through oracle/ref_vt_driver.f90).
 *
 * Built twice: -DORC_REAL=float / -DORC_REAL=double, -ffp-contract=off.
 * Arrays are compact [level][jy][ix] (the reference's (ix,jy,level) without padding);
 * inputs and outputs travel as double and are rounded to `real` on the way in.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real;
#define K(x) ((real)(x))

static inline real r_log(real x)  { return sizeof(real) == 4 ? (real)logf((float)x)  : (real)log((double)x); }
static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_sin(real x)  { return sizeof(real) == 4 ? (real)sinf((float)x)  : (real)sin((double)x); }
static inline real r_cos(real x)  { return sizeof(real) == 4 ? (real)cosf((float)x)  : (real)cos((double)x); }
static inline real r_atan(real x) { return sizeof(real) == 4 ? (real)atanf((float)x) : (real)atan((double)x); }
static inline real r_pow(real x, real y) { return sizeof(real) == 4 ? (real)powf((float)x, (float)y) : (real)pow((double)x, (double)y); }
static inline real r_abs(real x)  { return x < 0 ? -x : x; }
static inline real r_mod(real a, real p) { return sizeof(real) == 4 ? (real)fmodf((float)a, (float)p) : (real)fmod((double)a, (double)p); }

typedef struct {
  int nx, ny, nz;             /* nz = nuvz = nwz (gridcheck_ecmwf.f90 sets them equal) */
  double dx, dy, xlon0, ylat0;
  int nglobal, sglobal;
  double northpolemap[9], southpolemap[9], switchnorthg, switchsouthg;
  int init;                   /* 1: compute height/nmixz (first call), 0: use height[] as given */
  double cos_dy, cos_ylat0;   /* grid spacing and origin used in cosf (:405): the grid's own -- on a nest dyn(l), ylat0n(l), while
                                 dx, dy above stay the MOTHER's (dxconst, dyconst) */
  double xres, yres;          /* 1 on the mother grid; xresoln(l), yresoln(l) for verttransform_nests.f90 (same algorithm
                                 on the nest's arrays, no poles, no height initialisation: :384-385) */
  const double *akz, *bkz, *aknew, *bknew;          /* [nz] */
  const double *ps, *tt2, *td2;                     /* [ny][nx] */
  const double *tth, *qvh, *uuh, *vvh, *pvh, *wwh;  /* [nz][ny][nx] */
  double *height;             /* [nz] in/out */
  int *nmixz;
  double *uu, *vv, *ww, *tt, *qv, *pv, *rho, *drhodz, *uupol, *vvpol;   /* [nz][ny][nx] */
} vto_args;

/* ew.f90:4-29 */
static real vt_ew(real x) {
  real y, a, c, d;
  y = K(373.16) / x;
  a = K(-7.90298) * (y - K(1.));
  a = a + (K(5.02808) * K(0.43429) * r_log(y));
  c = (K(1.) - (K(1.) / y)) * K(11.344);
  c = K(-1.) + r_pow(K(10.), c);
  c = K(-1.3816) * c / K(1.e7);
  d = (K(1.) - y) * K(3.49149);
  d = K(-1.) + r_pow(K(10.), d);
  d = K(8.1328) * d / K(1.e3);
  y = a + c + d;
  return K(101324.6) * r_pow(K(10.), y);
}

/* cmapf_mod.f90:494-524 */
static real vt_cspanf(real value, real begin, real end) {
  real first = begin < end ? begin : end, last = begin > end ? begin : end;
  real val = r_mod(value - first, last - first);
  return val <= K(0.) ? val + last : val + first;
}

/* cmapf_mod.f90:24-52 */
static void vt_cc2gll(const real *s, real xlat, real xlong, real ue, real vn, real *ug, real *vg) {
  const real pi = K(3.14159265358979);     /* cmapf_mod.f90:19 (its own pi, not par_mod's) */
  const real radpdg = pi / K(180.);
  double along, rot, slong, clong, xpolg, ypolg;
  along = (double)vt_cspanf(xlong - s[1], K(-180.), K(180.));
  if (xlat > K(89.985)) rot = -(double)s[0] * along + (double)xlong - 180.;
  else if (xlat < K(-89.985)) rot = -(double)s[0] * along - (double)xlong;
  else rot = -(double)s[0] * along;
  slong = sin((double)radpdg * rot);
  clong = cos((double)radpdg * rot);
  xpolg = slong * (double)s[4] + clong * (double)s[5];
  ypolg = clong * (double)s[4] - slong * (double)s[5];
  *ug = (real)(ypolg * (double)ue + xpolg * (double)vn);
  *vg = (real)(ypolg * (double)vn - xpolg * (double)ue);
}

#define A3(p, ix, jy, k) (p)[(size_t)(ix) + (size_t)nx * ((size_t)(jy) + (size_t)ny * (size_t)((k) - 1))]   /* k 1-based */
#define A2(p, ix, jy) (p)[(size_t)(ix) + (size_t)nx * (size_t)(jy)]

static real *to_real(const double *a, size_t n) {
  real *r = (real *)malloc((n ? n : 1) * sizeof(real));
  size_t i;
  for (i = 0; i < n; i++) r[i] = (real)a[i];
  return r;
}
static void to_double(double *out, const real *a, size_t n) {
  size_t i;
  if (!out) return;
  for (i = 0; i < n; i++) out[i] = (double)a[i];
}

int vto_verttransform(vto_args *I) {
  const int nx = I->nx, ny = I->ny, nz = I->nz, nuvz = I->nz, nwz = I->nz;
  const int nxmin1 = nx - 1, nymin1 = ny - 1;
  const size_t n2 = (size_t)nx * ny, n3 = n2 * nz;
  const real pi = K(3.14159265), r_earth = K(6.371e6), r_air = K(287.05), ga = K(9.81);
  const real pi180 = pi / K(180.), hmixmax = K(4500.);
  const real konst = r_air / ga;                         /* verttransform_ecmwf.f90:81 */
  const real dx = (real)I->dx, dy = (real)I->dy, xlon0 = (real)I->xlon0, ylat0 = (real)I->ylat0;
  const real dxconst = K(180.) / (dx * r_earth * pi), dyconst = K(180.) / (dy * r_earth * pi);   /* gridcheck_ecmwf.f90:311-312 */
  real *akz = to_real(I->akz, nz), *bkz = to_real(I->bkz, nz), *aknew = to_real(I->aknew, nz), *bknew = to_real(I->bknew, nz);
  real *ps = to_real(I->ps, n2), *tt2 = to_real(I->tt2, n2), *td2 = to_real(I->td2, n2);
  real *tth = to_real(I->tth, n3), *qvh = to_real(I->qvh, n3), *uuh = to_real(I->uuh, n3), *vvh = to_real(I->vvh, n3);
  real *pvh = to_real(I->pvh, n3), *wwh = to_real(I->wwh, n3);
  real *height = to_real(I->height, nz);
  real *uvzlev = (real *)calloc(n3, sizeof(real)), *wzlev = (real *)calloc(n3, sizeof(real)), *rhoh = (real *)calloc(n3, sizeof(real));
  real *pinmconv = (real *)calloc(n3, sizeof(real));
  real *uu = (real *)calloc(n3, sizeof(real)), *vv = (real *)calloc(n3, sizeof(real)), *ww = (real *)calloc(n3, sizeof(real));
  real *tt = (real *)calloc(n3, sizeof(real)), *qv = (real *)calloc(n3, sizeof(real)), *pv = (real *)calloc(n3, sizeof(real));
  real *rho = (real *)calloc(n3, sizeof(real)), *drhodz = (real *)calloc(n3, sizeof(real));
  real *uupol = (real *)calloc(n3, sizeof(real)), *vvpol = (real *)calloc(n3, sizeof(real));
  real *tvold = (real *)calloc(n2, sizeof(real)), *pold = (real *)calloc(n2, sizeof(real));
  int *idx = (int *)calloc(n2, sizeof(int));
  real northpolemap[9], southpolemap[9];
  const real switchnorthg = (real)I->switchnorthg, switchsouthg = (real)I->switchsouthg;
  const real xres = (real)I->xres, yres = (real)I->yres;
  int ix, jy, kz, iz, i;
  for (i = 0; i < 9; i++) { northpolemap[i] = (real)I->northpolemap[i]; southpolemap[i] = (real)I->southpolemap[i]; }

  /* first call: reference z profile from the first column with ps > 1000 hPa, :134-176 */
  if (I->init) {
    int ixm = -1, jym = -1;
    real tvo, po;
    for (jy = 0; jy <= nymin1 && ixm < 0; jy++)
      for (ix = 0; ix <= nxmin1; ix++)
        if (A2(ps, ix, jy) > K(100000.)) { ixm = ix; jym = jy; break; }
    if (ixm < 0) return -1;   /* the reference would use uninitialised indices */
    tvo = A2(tt2, ixm, jym) * (K(1.) + K(0.378) * vt_ew(A2(td2, ixm, jym)) / A2(ps, ixm, jym));
    po = A2(ps, ixm, jym);
    height[0] = K(0.);
    for (kz = 2; kz <= nuvz; kz++) {
      real pint = akz[kz - 1] + bkz[kz - 1] * A2(ps, ixm, jym);
      real tv = A3(tth, ixm, jym, kz) * (K(1.) + K(0.608) * A3(qvh, ixm, jym, kz));
      if (r_abs(tv - tvo) > K(0.2))
        height[kz - 1] = height[kz - 2] + konst * r_log(po / pint) * (tv - tvo) / r_log(tv / tvo);
      else
        height[kz - 1] = height[kz - 2] + konst * r_log(po / pint) * tv;
      tvo = tv;
      po = pint;
    }
    /* highest level that can be within the PBL, :181-187 */
    for (kz = 1; kz <= nz; kz++)
      if (height[kz - 1] > hmixmax) { *I->nmixz = kz; break; }
  }

  /* heights of the eta levels, :203-237 */
  for (jy = 0; jy <= nymin1; jy++)
    for (ix = 0; ix <= nxmin1; ix++) {
      A2(tvold, ix, jy) = A2(tt2, ix, jy) * (K(1.) + K(0.378) * vt_ew(A2(td2, ix, jy)) / A2(ps, ix, jy));
      A2(pold, ix, jy) = A2(ps, ix, jy);
      A3(uvzlev, ix, jy, 1) = K(0.);
      A3(wzlev, ix, jy, 1) = K(0.);
      A3(rhoh, ix, jy, 1) = A2(pold, ix, jy) / (r_air * A2(tvold, ix, jy));
    }
  for (kz = 2; kz <= nuvz; kz++)
    for (jy = 0; jy <= nymin1; jy++)
      for (ix = 0; ix <= nxmin1; ix++) {
        real pint = akz[kz - 1] + bkz[kz - 1] * A2(ps, ix, jy);
        real tv = A3(tth, ix, jy, kz) * (K(1.) + K(0.608) * A3(qvh, ix, jy, kz));
        real tvo = A2(tvold, ix, jy), po = A2(pold, ix, jy);
        A3(rhoh, ix, jy, kz) = pint / (r_air * tv);
        if (r_abs(tv - tvo) > K(0.2))
          A3(uvzlev, ix, jy, kz) = A3(uvzlev, ix, jy, kz - 1) + konst * r_log(po / pint) * (tv - tvo) / r_log(tv / tvo);
        else
          A3(uvzlev, ix, jy, kz) = A3(uvzlev, ix, jy, kz - 1) + konst * r_log(po / pint) * tv;
        A2(tvold, ix, jy) = tv;
        A2(pold, ix, jy) = pint;
      }
  /* :240-244 */
  for (kz = 2; kz <= nwz - 1; kz++)
    for (jy = 0; jy <= nymin1; jy++)
      for (ix = 0; ix <= nxmin1; ix++)
        A3(wzlev, ix, jy, kz) = (A3(uvzlev, ix, jy, kz + 1) + A3(uvzlev, ix, jy, kz)) / K(2.);
  for (jy = 0; jy <= nymin1; jy++)
    for (ix = 0; ix <= nxmin1; ix++)
      A3(wzlev, ix, jy, nwz) = A3(wzlev, ix, jy, nwz - 1) + A3(uvzlev, ix, jy, nuvz) - A3(uvzlev, ix, jy, nuvz - 1);
  /* pinmconv=(h2-h1)/(p2-p1), :248-258 */
  for (jy = 0; jy <= nymin1; jy++)
    for (ix = 0; ix <= nxmin1; ix++) {
      const real p = A2(ps, ix, jy);
      A3(pinmconv, ix, jy, 1) = (A3(uvzlev, ix, jy, 2)) / ((aknew[1] + bknew[1] * p) - (aknew[0] + bknew[0] * p));
      for (kz = 2; kz <= nz - 1; kz++)
        A3(pinmconv, ix, jy, kz) = (A3(uvzlev, ix, jy, kz + 1) - A3(uvzlev, ix, jy, kz - 1)) /
                                   ((aknew[kz] + bknew[kz] * p) - (aknew[kz - 2] + bknew[kz - 2] * p));
      A3(pinmconv, ix, jy, nz) = (A3(uvzlev, ix, jy, nz) - A3(uvzlev, ix, jy, nz - 1)) /
                                 ((aknew[nz - 1] + bknew[nz - 1] * p) - (aknew[nz - 2] + bknew[nz - 2] * p));
    }

  /* levels where u,v,t and q are given, :264-356 */
  for (jy = 0; jy <= nymin1; jy++)
    for (ix = 0; ix <= nxmin1; ix++) {
      A3(uu, ix, jy, 1) = A3(uuh, ix, jy, 1); A3(vv, ix, jy, 1) = A3(vvh, ix, jy, 1);
      A3(tt, ix, jy, 1) = A3(tth, ix, jy, 1); A3(qv, ix, jy, 1) = A3(qvh, ix, jy, 1);
      A3(pv, ix, jy, 1) = A3(pvh, ix, jy, 1); A3(rho, ix, jy, 1) = A3(rhoh, ix, jy, 1);
      A3(uu, ix, jy, nz) = A3(uuh, ix, jy, nuvz); A3(vv, ix, jy, nz) = A3(vvh, ix, jy, nuvz);
      A3(tt, ix, jy, nz) = A3(tth, ix, jy, nuvz); A3(qv, ix, jy, nz) = A3(qvh, ix, jy, nuvz);
      A3(pv, ix, jy, nz) = A3(pvh, ix, jy, nuvz); A3(rho, ix, jy, nz) = A3(rhoh, ix, jy, nuvz);
      A2(idx, ix, jy) = 2;
    }
  for (iz = 2; iz <= nz - 1; iz++) {
    for (jy = 0; jy <= nymin1; jy++)
      for (ix = 0; ix <= nxmin1; ix++) {
        if (height[iz - 1] > A3(uvzlev, ix, jy, nuvz)) {
          A3(uu, ix, jy, iz) = A3(uu, ix, jy, nz); A3(vv, ix, jy, iz) = A3(vv, ix, jy, nz);
          A3(tt, ix, jy, iz) = A3(tt, ix, jy, nz); A3(qv, ix, jy, iz) = A3(qv, ix, jy, nz);
          A3(pv, ix, jy, iz) = A3(pv, ix, jy, nz); A3(rho, ix, jy, iz) = A3(rho, ix, jy, nz);
        } else {
          for (kz = A2(idx, ix, jy); kz <= nuvz; kz++)
            if (A2(idx, ix, jy) <= kz && height[iz - 1] > A3(uvzlev, ix, jy, kz - 1) && height[iz - 1] <= A3(uvzlev, ix, jy, kz)) {
              A2(idx, ix, jy) = kz;
              break;
            }
        }
      }
    for (jy = 0; jy <= nymin1; jy++)
      for (ix = 0; ix <= nxmin1; ix++)
        if (height[iz - 1] <= A3(uvzlev, ix, jy, nuvz)) {
          real dz1, dz2, dz;
          kz = A2(idx, ix, jy);
          dz1 = height[iz - 1] - A3(uvzlev, ix, jy, kz - 1);
          dz2 = A3(uvzlev, ix, jy, kz) - height[iz - 1];
          dz = dz1 + dz2;
          A3(uu, ix, jy, iz) = (A3(uuh, ix, jy, kz - 1) * dz2 + A3(uuh, ix, jy, kz) * dz1) / dz;
          A3(vv, ix, jy, iz) = (A3(vvh, ix, jy, kz - 1) * dz2 + A3(vvh, ix, jy, kz) * dz1) / dz;
          A3(tt, ix, jy, iz) = (A3(tth, ix, jy, kz - 1) * dz2 + A3(tth, ix, jy, kz) * dz1) / dz;
          A3(qv, ix, jy, iz) = (A3(qvh, ix, jy, kz - 1) * dz2 + A3(qvh, ix, jy, kz) * dz1) / dz;
          A3(pv, ix, jy, iz) = (A3(pvh, ix, jy, kz - 1) * dz2 + A3(pvh, ix, jy, kz) * dz1) / dz;
          A3(rho, ix, jy, iz) = (A3(rhoh, ix, jy, kz - 1) * dz2 + A3(rhoh, ix, jy, kz) * dz1) / dz;
        }
  }

  /* levels where w is given, :362-389 */
  for (jy = 0; jy <= nymin1; jy++)
    for (ix = 0; ix <= nxmin1; ix++) {
      A3(ww, ix, jy, 1) = A3(wwh, ix, jy, 1) * A3(pinmconv, ix, jy, 1);
      A3(ww, ix, jy, nz) = A3(wwh, ix, jy, nwz) * A3(pinmconv, ix, jy, nz);
      A2(idx, ix, jy) = 2;
    }
  for (iz = 2; iz <= nz; iz++) {
    for (jy = 0; jy <= nymin1; jy++)
      for (ix = 0; ix <= nxmin1; ix++)
        for (kz = A2(idx, ix, jy); kz <= nwz; kz++)
          if (A2(idx, ix, jy) <= kz && height[iz - 1] > A3(wzlev, ix, jy, kz - 1) && height[iz - 1] <= A3(wzlev, ix, jy, kz)) {
            A2(idx, ix, jy) = kz;
            break;
          }
    for (jy = 0; jy <= nymin1; jy++)
      for (ix = 0; ix <= nxmin1; ix++) {
        real dz1, dz2, dz;
        kz = A2(idx, ix, jy);
        dz1 = height[iz - 1] - A3(wzlev, ix, jy, kz - 1);
        dz2 = A3(wzlev, ix, jy, kz) - height[iz - 1];
        dz = dz1 + dz2;
        A3(ww, ix, jy, iz) = (A3(wwh, ix, jy, kz - 1) * A3(pinmconv, ix, jy, kz - 1) * dz2 + A3(wwh, ix, jy, kz) * A3(pinmconv, ix, jy, kz) * dz1) / dz;
      }
  }

  /* density gradients, :394-400 */
  for (jy = 0; jy <= nymin1; jy++)
    for (ix = 0; ix <= nxmin1; ix++) {
      A3(drhodz, ix, jy, 1) = (A3(rho, ix, jy, 2) - A3(rho, ix, jy, 1)) / (height[1] - height[0]);
      for (kz = 2; kz <= nz - 1; kz++)
        A3(drhodz, ix, jy, kz) = (A3(rho, ix, jy, kz + 1) - A3(rho, ix, jy, kz - 1)) / (height[kz] - height[kz - 2]);
      A3(drhodz, ix, jy, nz) = A3(drhodz, ix, jy, nz - 1);
    }

  /* slope of the eta levels in windward direction and resulting w correction, :411-453 */
  for (jy = 0; jy <= nymin1; jy++)
    for (ix = 0; ix <= nxmin1; ix++) A2(idx, ix, jy) = 2;
  for (iz = 2; iz <= nz - 1; iz++) {
    for (jy = 1; jy <= ny - 2; jy++)
      for (ix = 1; ix <= nx - 2; ix++)
        for (kz = A2(idx, ix, jy); kz <= nz; kz++)
          if (A2(idx, ix, jy) <= kz && height[iz - 1] > A3(uvzlev, ix, jy, kz - 1) && height[iz - 1] <= A3(uvzlev, ix, jy, kz)) {
            A2(idx, ix, jy) = kz;
            break;
          }
    for (jy = 1; jy <= ny - 2; jy++) {
      const real cosf = K(1.) / r_cos(((real)jy * (real)I->cos_dy + (real)I->cos_ylat0) * pi180);
      for (ix = 1; ix <= nx - 2; ix++) {
        real dz1, dz2, dz, dzdx1, dzdx2, dzdx, dzdy1, dzdy2, dzdy;
        const int ix1 = ix - 1, jy1 = jy - 1, ixp = ix + 1, jyp = jy + 1;
        kz = A2(idx, ix, jy);
        dz1 = height[iz - 1] - A3(uvzlev, ix, jy, kz - 1);
        dz2 = A3(uvzlev, ix, jy, kz) - height[iz - 1];
        dz = dz1 + dz2;
        dzdx1 = (A3(uvzlev, ixp, jy, kz - 1) - A3(uvzlev, ix1, jy, kz - 1)) / K(2.);
        dzdx2 = (A3(uvzlev, ixp, jy, kz) - A3(uvzlev, ix1, jy, kz)) / K(2.);
        dzdx = (dzdx1 * dz2 + dzdx2 * dz1) / dz;
        dzdy1 = (A3(uvzlev, ix, jyp, kz - 1) - A3(uvzlev, ix, jy1, kz - 1)) / K(2.);
        dzdy2 = (A3(uvzlev, ix, jyp, kz) - A3(uvzlev, ix, jy1, kz)) / K(2.);
        dzdy = (dzdy1 * dz2 + dzdy2 * dz1) / dz;
        A3(ww, ix, jy, iz) = A3(ww, ix, jy, iz) + (dzdx * A3(uu, ix, jy, iz) * dxconst * xres * cosf + dzdy * A3(vv, ix, jy, iz) * dyconst * yres);
      }
    }
  }

  /* north pole: polar stereographic winds, :459-522 */
  if (I->nglobal) {
    for (iz = 1; iz <= nz; iz++) {
      real xlon, xlonr, ffpol, ddpol, uuaux, vvaux, uupolaux, vvpolaux, wdummy, ucen, vcen;
      for (jy = (int)switchnorthg - 2; jy <= nymin1; jy++) {
        const real ylat = ylat0 + (real)jy * dy;
        for (ix = 0; ix <= nxmin1; ix++) {
          xlon = xlon0 + (real)ix * dx;
          vt_cc2gll(northpolemap, ylat, xlon, A3(uu, ix, jy, iz), A3(vv, ix, jy, iz), &A3(uupol, ix, jy, iz), &A3(vvpol, ix, jy, iz));
        }
      }
      xlon = xlon0 + (real)(nx / 2 - 1) * dx;
      xlonr = xlon * pi / K(180.);
      ucen = A3(uu, nx / 2 - 1, nymin1, iz); vcen = A3(vv, nx / 2 - 1, nymin1, iz);
      ffpol = r_sqrt(ucen * ucen + vcen * vcen);
      if (vcen < K(0.)) ddpol = r_atan(ucen / vcen) - xlonr;
      else if (vcen > K(0.)) ddpol = pi + r_atan(ucen / vcen) - xlonr;
      else ddpol = pi / K(2.) - xlonr;
      if (ddpol < K(0.)) ddpol = K(2.0) * pi + ddpol;
      if (ddpol > K(2.0) * pi) ddpol = ddpol - K(2.0) * pi;
      xlon = K(180.0);
      xlonr = xlon * pi / K(180.);
      uuaux = -ffpol * r_sin(xlonr + ddpol);
      vvaux = -ffpol * r_cos(xlonr + ddpol);
      vt_cc2gll(northpolemap, K(90.0), xlon, uuaux, vvaux, &uupolaux, &vvpolaux);
      for (ix = 0; ix <= nxmin1; ix++) { A3(uupol, ix, nymin1, iz) = uupolaux; A3(vvpol, ix, nymin1, iz) = vvpolaux; }
      /* w at the pole = zonal mean of the next parallel, :508-520 */
      wdummy = K(0.);
      for (ix = 0; ix <= nxmin1; ix++) wdummy = wdummy + A3(ww, ix, ny - 2, iz);
      wdummy = wdummy / (real)nx;
      for (ix = 0; ix <= nxmin1; ix++) A3(ww, ix, nymin1, iz) = wdummy;
    }
  }
  /* south pole, :530-597 (the auxiliary point goes through northpolemap, :576) */
  if (I->sglobal) {
    for (iz = 1; iz <= nz; iz++) {
      real xlon, xlonr, ffpol, ddpol, uuaux, vvaux, uupolaux, vvpolaux, wdummy, ucen, vcen;
      for (jy = 0; jy <= (int)switchsouthg + 3; jy++) {
        const real ylat = ylat0 + (real)jy * dy;
        for (ix = 0; ix <= nxmin1; ix++) {
          xlon = xlon0 + (real)ix * dx;
          vt_cc2gll(southpolemap, ylat, xlon, A3(uu, ix, jy, iz), A3(vv, ix, jy, iz), &A3(uupol, ix, jy, iz), &A3(vvpol, ix, jy, iz));
        }
      }
      xlon = xlon0 + (real)(nx / 2 - 1) * dx;
      xlonr = xlon * pi / K(180.);
      ucen = A3(uu, nx / 2 - 1, 0, iz); vcen = A3(vv, nx / 2 - 1, 0, iz);
      ffpol = r_sqrt(ucen * ucen + vcen * vcen);
      if (vcen < K(0.)) ddpol = r_atan(ucen / vcen) + xlonr;
      else if (vcen > K(0.)) ddpol = pi + r_atan(ucen / vcen) + xlonr;
      else ddpol = pi / K(2.) - xlonr;
      if (ddpol < K(0.)) ddpol = K(2.0) * pi + ddpol;
      if (ddpol > K(2.0) * pi) ddpol = ddpol - K(2.0) * pi;
      xlon = K(180.0);
      xlonr = xlon * pi / K(180.);
      uuaux = +ffpol * r_sin(xlonr - ddpol);
      vvaux = -ffpol * r_cos(xlonr - ddpol);
      vt_cc2gll(northpolemap, K(-90.0), xlon, uuaux, vvaux, &uupolaux, &vvpolaux);
      for (ix = 0; ix <= nxmin1; ix++) { A3(uupol, ix, 0, iz) = uupolaux; A3(vvpol, ix, 0, iz) = vvpolaux; }
      wdummy = K(0.);
      for (ix = 0; ix <= nxmin1; ix++) wdummy = wdummy + A3(ww, ix, 1, iz);
      wdummy = wdummy / (real)nx;
      for (ix = 0; ix <= nxmin1; ix++) A3(ww, ix, 0, iz) = wdummy;
    }
  }

  to_double(I->height, height, nz);
  to_double(I->uu, uu, n3); to_double(I->vv, vv, n3); to_double(I->ww, ww, n3);
  to_double(I->tt, tt, n3); to_double(I->qv, qv, n3); to_double(I->pv, pv, n3);
  to_double(I->rho, rho, n3); to_double(I->drhodz, drhodz, n3);
  to_double(I->uupol, uupol, n3); to_double(I->vvpol, vvpol, n3);
  free(akz); free(bkz); free(aknew); free(bknew); free(ps); free(tt2); free(td2);
  free(tth); free(qvh); free(uuh); free(vvh); free(pvh); free(wwh); free(height);
  free(uvzlev); free(wzlev); free(rhoh); free(pinmconv);
  free(uu); free(vv); free(ww); free(tt); free(qv); free(pv); free(rho); free(drhodz); free(uupol); free(vvpol);
  free(tvold); free(pold); free(idx);
  return 0;
}
