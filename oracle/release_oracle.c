/* TEST INFRASTRUCTURE ONLY -- the checker for flexpart_amd's fpx_releaseparticles / fpx_split_particles.
 * Nothing under flexpart_amd/ links, loads or calls this file.
 *
 * Plain-C restatement of the reference's release routine and of the particle-splitting block of the time manager
 * (SURVEY section 8 f2), each part citing the lines it follows:
 *   releaseparticles.f90:63-375   (with random_mod.f90:12-42 ran1, juldate.f90, caldate.f90)
 *   timemanager.f90:473-504       (splitting)
 * `real` is the reference's default real kind (compile with -DORC_REAL=float|double); the serial semantics are kept
 * exactly: the slot search from minpart, the shared ran1 stream with four draws per particle in the order x, y, class, z,
 * xmasssave / numparticlecount / rho_rel as state that persists between calls.  Pinned against the flang build of the
 * unmodified routine (oracle/ref_rel_driver.f90 -> oracle/_ref/relref_rK) by tests/test_release.py.
 * Arrays are compact: oro [ny][nx], rho2 / tt2 [nz][ny][nx] (the literal time slot 2 the routine reads). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real;
#define K(x) ((real)(x))

typedef struct {
  /* grid */
  int nx, ny, nz, xglobal;
  double dx, dy, xlon0, ylat0;
  const double *height, *oro, *rho2, *tt2;
  /* run */
  int nspec, ldirect, lsynctime, mintime, itsplit, ind_rel, mquasilag, nclassunc, ibdate, ibtime;
  double eps_nxmax;             /* par_mod nxmax of the build: eps = nxmax/3.e5 (unused without nests) */
  /* release points (point_mod): [numpoint]; xmass [nspec][numpoint] */
  int numpoint;
  const int *ireleasestart, *ireleaseend, *npart, *kindz;
  const double *xpoint1, *xpoint2, *ypoint1, *ypoint2, *zpoint1, *zpoint2, *xmass;
  const double *point_hour, *area_hour;   /* [24][nspec] */
  const double *point_dow, *area_dow;     /* [7][nspec]  */
  /* particles, capacity maxpart; xmass1 [nspec][maxpart] */
  long maxpart, numpart;
  int numparticlecount;
  double *xtra1, *ytra1, *ztra1, *uap, *xmass1;
  int *itra1, *itramem, *itrasplit, *idt, *npoint, *nclass;
  /* state that persists between calls */
  double *xmasssave, *rho_rel;   /* [numpoint] */
  int ran1_idum, ran1_iy, ran1_iv[32];
  int status;                    /* 1: no free storage space left (the routine stops, :369-378) */
  /* one nested wind field (releaseparticles.f90:196-226,231-341): corners and resolution factors in mother grid units,
   * orography oron [nyn][nxn], rhon / ttn of the literal time slot 2 [nz][nyn][nxn] */
  int nest_on, nxn, nyn, pad_;
  double xln, yln, xrn, yrn, xresoln, yresoln;
  const double *oron, *rhon2, *ttn2;
} rlo_args;

/* juldate.f90 in the build's real kind */
static double rlo_juldate(int yyyymmdd, int hhmiss) {
  const int igreg = 15 + 31 * (10 + 12 * 1582);
  int yyyy = yyyymmdd / 10000, mm = (yyyymmdd - 10000 * yyyy) / 100, dd = yyyymmdd - 10000 * yyyy - 100 * mm;
  int hh = hhmiss / 10000, mi = (hhmiss - 10000 * hh) / 100, ss = hhmiss - 10000 * hh - 100 * mi;
  int jy, jm, julday, ja;
  if (yyyy < 0) yyyy = yyyy + 1;
  if (mm > 2) { jy = yyyy; jm = mm + 1; } else { jy = yyyy - 1; jm = mm + 13; }
  julday = (int)(K(365.25) * (real)jy) + (int)(K(30.6001) * (real)jm) + dd + 1720995;
  if (dd + 31 * (mm + 12 * yyyy) >= igreg) {
    ja = (int)(K(0.01) * (real)jy);
    julday = julday + 2 - ja + (int)(K(0.25) * (real)ja);
  }
  return (double)julday + (double)hh / 24. + (double)mi / 1440. + (double)ss / 86400.;
}
double rlo_juldate_pub(int yyyymmdd, int hhmiss) { return rlo_juldate(yyyymmdd, hhmiss); }

/* caldate.f90: only yyyymmdd is used by releaseparticles (the month decides on daylight saving time) */
static int rlo_caldate_yyyymmdd(double juldate) {
  const int igreg = 2299161;
  int julday = (int)juldate, ja, jb, jc, jd, je, jalpha, dd, mm, yyyy;
  if ((juldate - julday) * 86400. >= 86399.5) {
    juldate = juldate + juldate - julday - 86399.5 / 86400.;
    julday = (int)juldate;
  }
  if (julday >= igreg) {
    jalpha = (int)((((real)(julday - 1867216)) - K(0.25)) / K(36524.25));
    ja = julday + 1 + jalpha - (int)(K(0.25) * (real)jalpha);
  } else ja = julday;
  jb = ja + 1524;
  jc = (int)(K(6680.) + (((real)(jb - 2439870)) - K(122.1)) / K(365.25));
  jd = 365 * jc + (int)(K(0.25) * (real)jc);
  je = (int)((real)(jb - jd) / K(30.6001));
  dd = jb - jd - (int)(K(30.6001) * (real)je);
  mm = je - 1;
  if (mm > 12) mm = mm - 12;
  yyyy = jc - 4715;
  if (mm > 2) yyyy = yyyy - 1;
  if (yyyy <= 0) yyyy = yyyy - 1;
  return 10000 * yyyy + 100 * mm + dd;
}

/* random_mod.f90:12-42 */
static real rlo_ran1(rlo_args *A) {
  const int ia = 16807, im = 2147483647, iq = 127773, ir = 2836, ntab = 32, ndiv = 1 + (im - 1) / ntab;
  const real am = K(1.) / (real)im, rnmx = K(1.) - K(1.2e-7);
  int j, k;
  real r;
  if (A->ran1_idum <= 0 || A->ran1_iy == 0) {
    A->ran1_idum = -A->ran1_idum > 1 ? -A->ran1_idum : 1;
    for (j = ntab + 8; j >= 1; j--) {
      k = A->ran1_idum / iq;
      A->ran1_idum = ia * (A->ran1_idum - k * iq) - ir * k;
      if (A->ran1_idum < 0) A->ran1_idum += im;
      if (j <= ntab) A->ran1_iv[j - 1] = A->ran1_idum;
    }
    A->ran1_iy = A->ran1_iv[0];
  }
  k = A->ran1_idum / iq;
  A->ran1_idum = ia * (A->ran1_idum - k * iq) - ir * k;
  if (A->ran1_idum < 0) A->ran1_idum += im;
  j = 1 + A->ran1_iy / ndiv;
  A->ran1_iy = A->ran1_iv[j - 1];
  A->ran1_iv[j - 1] = A->ran1_idum;
  r = am * (real)A->ran1_iy;
  return r < rnmx ? r : rnmx;
}

void rlo_init(rlo_args *A) { A->ran1_idum = -7; A->ran1_iy = 0; memset(A->ran1_iv, 0, sizeof A->ran1_iv); A->status = 0; }

#define F2(f, i, j) ((real)(f)[(size_t)(i) + (size_t)A->nx * (size_t)(j)])
#define F3(f, i, j, k) ((real)(f)[(size_t)(i) + (size_t)A->nx * ((size_t)(j) + (size_t)A->ny * (size_t)((k) - 1))])
#define HGT(k) ((real)A->height[(k) - 1])

/* releaseparticles.f90:63-375 */
void rlo_releaseparticles(rlo_args *A, int itime) {
  const real eps2 = K(1.e-6), r_air = K(287.05);
  const int nspec = A->nspec;
  double julmonday, jul, jullocal, juldiff;
  int jjjjmmdd, mm, i, j, k, minpart;
  real timecorrect[16];
  julmonday = rlo_juldate(19000101, 0);                                     /* :63 */
  jul = rlo_juldate(A->ibdate, A->ibtime) + (double)itime / 86400.;         /* bdate + itime/86400 */
  jjjjmmdd = rlo_caldate_yyyymmdd(jul);
  mm = (jjjjmmdd - 10000 * (jjjjmmdd / 10000)) / 100;
  if (mm >= 4 && mm <= 9) jul = jul + 1. / 24.;                             /* :67 */
  minpart = 1;
  for (i = 1; i <= A->numpoint; i++) {
    real xlonav, average_timecorrect, rfraction, xaux, yaux, zaux;
    int nweeks, ndayofweek, nhour, numrel;
    if (!(itime >= A->ireleasestart[i - 1] && itime <= A->ireleaseend[i - 1])) continue;   /* :75-76 */
    xlonav = (real)A->xlon0 + ((real)A->xpoint2[i - 1] + (real)A->xpoint1[i - 1]) / K(2.) * (real)A->dx;
    if (xlonav < K(-180.)) xlonav = xlonav + K(360.);
    if (xlonav > K(180.)) xlonav = xlonav - K(360.);
    jullocal = jul + (double)xlonav / 360.;
    juldiff = jullocal - julmonday;
    nweeks = (int)(juldiff / 7.);
    juldiff = juldiff - (double)nweeks * 7.;
    ndayofweek = (int)juldiff + 1;
    nhour = (int)lround((juldiff - (double)(ndayofweek - 1)) * 24.);
    if (nhour == 0) { nhour = 24; ndayofweek = ndayofweek - 1; if (ndayofweek == 0) ndayofweek = 7; }
    average_timecorrect = K(0.);
    for (k = 1; k <= nspec; k++) {                                          /* :100-110 */
      if (fabs((double)((real)A->xpoint2[i - 1] - (real)A->xpoint1[i - 1])) < 1.e-4 &&
          fabs((double)((real)A->ypoint2[i - 1] - (real)A->ypoint1[i - 1])) < 1.e-4)
        timecorrect[k - 1] = (real)A->point_hour[(size_t)(nhour - 1) * nspec + (k - 1)] * (real)A->point_dow[(size_t)(ndayofweek - 1) * nspec + (k - 1)];
      else
        timecorrect[k - 1] = (real)A->area_hour[(size_t)(nhour - 1) * nspec + (k - 1)] * (real)A->area_dow[(size_t)(ndayofweek - 1) * nspec + (k - 1)];
      average_timecorrect = average_timecorrect + timecorrect[k - 1];
    }
    average_timecorrect = average_timecorrect / (real)nspec;
    if (A->ireleasestart[i - 1] != A->ireleaseend[i - 1]) {                 /* :116-128 */
      rfraction = (real)fabs((double)((real)A->npart[i - 1] * (real)A->lsynctime / (real)(A->ireleaseend[i - 1] - A->ireleasestart[i - 1])));
      if (itime == A->ireleasestart[i - 1] || itime == A->ireleaseend[i - 1]) rfraction = rfraction / K(2.);
      rfraction = rfraction * average_timecorrect;
      rfraction = rfraction + (real)A->xmasssave[i - 1];
      numrel = (int)rfraction;
      A->xmasssave[i - 1] = (double)(rfraction - (real)numrel);
    } else numrel = A->npart[i - 1];
    xaux = (real)A->xpoint2[i - 1] - (real)A->xpoint1[i - 1];
    yaux = (real)A->ypoint2[i - 1] - (real)A->ypoint1[i - 1];
    zaux = (real)A->zpoint2[i - 1] - (real)A->zpoint1[i - 1];
    for (j = 1; j <= numrel; j++) {
      long ipart;
      for (ipart = minpart; ipart <= A->maxpart; ipart++) {                 /* :133-137 */
        if (A->itra1[ipart - 1] != itime) {
          double xt, yt;
          real zt, topo, ddx, ddy, rddx, rddy, p1, p2, p3, p4;
          int ix, jy, ixp, jyp;
          xt = (double)((real)A->xpoint1[i - 1] + rlo_ran1(A) * xaux);      /* :139 */
          if (A->xglobal) {
            if (xt > (double)(real)(A->nx - 1)) xt = xt - (double)(real)(A->nx - 1);
            if (xt < 0.) xt = xt + (double)(real)(A->nx - 1);
          }
          yt = (double)((real)A->ypoint1[i - 1] + rlo_ran1(A) * yaux);      /* :146 */
          for (k = 1; k <= nspec; k++)                                      /* :156-157 */
            A->xmass1[(size_t)(k - 1) * A->maxpart + (ipart - 1)] =
                (double)((real)A->xmass[(size_t)(k - 1) * A->numpoint + (i - 1)] / (real)A->npart[i - 1] * timecorrect[k - 1] / average_timecorrect);
          {
            int nc = (int)(rlo_ran1(A) * (real)A->nclassunc) + 1;          /* :168-169 */
            A->nclass[ipart - 1] = nc < A->nclassunc ? nc : A->nclassunc;
          }
          A->numparticlecount = A->numparticlecount + 1;
          A->npoint[ipart - 1] = A->mquasilag == 0 ? i : A->numparticlecount;
          A->idt[ipart - 1] = A->mintime;
          A->itra1[ipart - 1] = itime;
          A->itramem[ipart - 1] = itime;
          A->itrasplit[ipart - 1] = itime + A->ldirect * A->itsplit;
          zt = (real)A->zpoint1[i - 1] + rlo_ran1(A) * zaux;                /* :183 */
          /* the nest we are in, :196-206; grid coordinates and weights, :211-226 */
          int ngrid = 0;
          const double *g_oro = A->oro, *g_rho = A->rho2, *g_tt = A->tt2;
          size_t gnx = (size_t)A->nx, gny = (size_t)A->ny;
          if (A->nest_on) {
            const real eps = (real)A->eps_nxmax / K(3.e5);
            if (xt > (double)((real)A->xln + eps) && xt < (double)((real)A->xrn - eps) && yt > (double)((real)A->yln + eps) && yt < (double)((real)A->yrn - eps)) ngrid = 1;
          }
          if (ngrid > 0) {
            const real xtn = (real)((xt - (double)(real)A->xln) * (double)(real)A->xresoln);
            const real ytn = (real)((yt - (double)(real)A->yln) * (double)(real)A->yresoln);
            ix = (int)xtn; jy = (int)ytn;
            ddy = ytn - (real)jy;
            ddx = xtn - (real)ix;
            g_oro = A->oron; g_rho = A->rhon2; g_tt = A->ttn2; gnx = (size_t)A->nxn; gny = (size_t)A->nyn;
          } else {
            ix = (int)xt; jy = (int)yt;
            ddy = (real)(yt - (double)(real)jy);
            ddx = (real)(xt - (double)(real)ix);
          }
#undef F2
#undef F3
#define F2(f, i, j) ((real)(f)[(size_t)(i) + gnx * (size_t)(j)])
#define F3(f, i, j, k) ((real)(f)[(size_t)(i) + gnx * ((size_t)(j) + gny * (size_t)((k) - 1))])
          ixp = ix + 1; jyp = jy + 1;
          rddx = K(1.) - ddx; rddy = K(1.) - ddy;
          p1 = rddx * rddy; p2 = ddx * rddy; p3 = rddx * ddy; p4 = ddx * ddy;
          topo = p1 * F2(g_oro, ix, jy) + p2 * F2(g_oro, ixp, jy) + p3 * F2(g_oro, ix, jyp) + p4 * F2(g_oro, ixp, jyp);
          if (A->kindz[i - 1] == 3) {                                       /* :231-273 */
            const real presspart = zt;
            real press, pressold = K(0.);
            int kz;
            for (kz = 1; kz <= A->nz; kz++) {
              const real r = p1 * F3(g_rho, ix, jy, kz) + p2 * F3(g_rho, ixp, jy, kz) + p3 * F3(g_rho, ix, jyp, kz) + p4 * F3(g_rho, ixp, jyp, kz);
              const real t = p1 * F3(g_tt, ix, jy, kz) + p2 * F3(g_tt, ixp, jy, kz) + p3 * F3(g_tt, ix, jyp, kz) + p4 * F3(g_tt, ixp, jyp, kz);
              press = r * r_air * t / K(100.);
              if (kz == 1) pressold = press;
              if (press < presspart) {
                if (kz == 1) zt = HGT(1) / K(2.);
                else {
                  const real dz1 = pressold - presspart, dz2 = presspart - press;
                  zt = (HGT(kz - 1) * dz2 + HGT(kz) * dz1) / (dz1 + dz2);
                }
                break;
              }
              pressold = press;
            }
          }
          if (A->kindz[i - 1] == 2) zt = zt - topo;                         /* :278 */
          if (zt < eps2) zt = eps2;
          if (zt > HGT(A->nz) - K(0.5)) zt = HGT(A->nz) - K(0.5);
          if (A->ind_rel == 1 || A->ind_rel == 3 || A->ind_rel == 4) {      /* :300-341 */
            int ii, indz = A->nz - 1, indzp = A->nz, n;
            real dz1, dz2, dz, rhoaux[2], rhoout;
            for (ii = 2; ii <= A->nz; ii++)
              if (HGT(ii) > zt) { indz = ii - 1; indzp = ii; break; }
            dz1 = zt - HGT(indz); dz2 = HGT(indzp) - zt; dz = K(1.) / (dz1 + dz2);
            for (n = 1; n <= 2; n++)
              rhoaux[n - 1] = p1 * F3(g_rho, ix, jy, indz + n - 1) + p2 * F3(g_rho, ixp, jy, indz + n - 1) +
                              p3 * F3(g_rho, ix, jyp, indz + n - 1) + p4 * F3(g_rho, ixp, jyp, indz + n - 1);
            rhoout = (dz2 * rhoaux[0] + dz1 * rhoaux[1]) * dz;
            A->rho_rel[i - 1] = (double)rhoout;
            for (k = 1; k <= nspec; k++) {
              double *m = &A->xmass1[(size_t)(k - 1) * A->maxpart + (ipart - 1)];
              *m = (double)((real)*m * rhoout);
            }
          }
          A->xtra1[ipart - 1] = xt; A->ytra1[ipart - 1] = yt; A->ztra1[ipart - 1] = (double)zt;
          if (ipart > A->numpart) A->numpart = ipart;                       /* :362 */
          break;
        }
      }
      if (ipart > A->maxpart) { A->status = 1; return; }                    /* :366 -> 996 */
      minpart = (int)ipart + 1;
    }
  }
}

/* timemanager.f90:473-504 */
void rlo_split(rlo_args *A, int itime) {
  long j, n;
  int ks;
  if (!(A->ldirect * itime >= A->ldirect * A->itsplit)) return;
  n = A->numpart;
  for (j = 1; j <= A->numpart; j++) {
    if (A->ldirect * itime >= A->ldirect * A->itrasplit[j - 1]) {
      if (n < A->maxpart) {
        n = n + 1;
        A->itrasplit[j - 1] = 2 * (A->itrasplit[j - 1] - A->itramem[j - 1]) + A->itramem[j - 1];
        A->itrasplit[n - 1] = A->itrasplit[j - 1];
        A->itramem[n - 1] = A->itramem[j - 1];
        A->itra1[n - 1] = A->itra1[j - 1];
        A->idt[n - 1] = A->idt[j - 1];
        A->npoint[n - 1] = A->npoint[j - 1];
        A->nclass[n - 1] = A->nclass[j - 1];
        A->xtra1[n - 1] = A->xtra1[j - 1]; A->ytra1[n - 1] = A->ytra1[j - 1]; A->ztra1[n - 1] = A->ztra1[j - 1];
        A->uap[n - 1] = A->uap[j - 1];   /* ucp, uzp, us, vs, ws, cbt are copied the same way; the test carries uap as their witness */
        for (ks = 0; ks < A->nspec; ks++) {
          double *m = &A->xmass1[(size_t)ks * A->maxpart + (j - 1)];
          *m = (double)((real)*m / K(2.));
          A->xmass1[(size_t)ks * A->maxpart + (n - 1)] = *m;
        }
      }
    }
  }
  A->numpart = n;
}
