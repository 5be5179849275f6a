/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's
 * per-particle trajectory step.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product path (flexpart_amd/)
 * never does.
 *
 * Plain C restatement of the algorithm of MeteoSwiss/flexpart's
 * timemanager -> initialize/advance path.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference/src).  It keeps the
 * reference's control flow, operation order and quirks (module-global scratch,
 * shared sequential ran3 stream, 1e6-entry Gaussian table) so that it can be
 * pinned against outputs of the real reference compiled with flang
 * (oracle/_ref/flexref_r4|r8, see oracle/build_ref.sh) -- the reference holds
 * no test vectors of its own for this path.
 *
 * Built twice: -DORC_REAL=float  (reference precision, default real = 4 bytes)
 *              -DORC_REAL=double (the -fdefault-real-8 "fp64" oracle).
 * Compile with -ffp-contract=off: the flang x86-64 build has no FMA.
 *
 * Field layout: compact C arrays [slot][level][jy][ix] (= the reference's
 * column-major (ix,jy,level,slot) without the nxmax/nymax padding).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real;
typedef float dep_real;   /* par_mod.f90:33 dep_prec = sp: deposition grids stay 4-byte in every build */
#define K(x) ((real)(x))

#define ORC_MAXSPEC 5
#define ORC_MAXRECEPTOR 20   /* par_mod.f90:201 */
#define ORC_NZMAX 256
#define ORC_MAXRAND 1000000
#define ORC_MAXNESTS 4

/* type-generic libm in the oracle's precision */
static inline real r_exp(real x)  { return sizeof(real) == 4 ? (real)expf((float)x)  : (real)exp((double)x); }
static inline real r_log(real x)  { return sizeof(real) == 4 ? (real)logf((float)x)  : (real)log((double)x); }
static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_sin(real x)  { return sizeof(real) == 4 ? (real)sinf((float)x)  : (real)sin((double)x); }
static inline real r_cos(real x)  { return sizeof(real) == 4 ? (real)cosf((float)x)  : (real)cos((double)x); }
static inline real r_erf(real x)  { return sizeof(real) == 4 ? (real)erff((float)x)  : (real)erf((double)x); }
static inline real r_pow(real x, real y) { return sizeof(real) == 4 ? (real)powf((float)x, (float)y) : (real)pow((double)x, (double)y); }
static inline real r_abs(real x)  { return x < 0 ? -x : x; }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_min(real a, real b) { return a < b ? a : b; }
static inline real r_sign(real a, real b) { real m = r_abs(a); return (b < 0 || (b == 0 && signbit((double)b))) ? -m : m; }
static inline real r_mod(real a, real p) { return sizeof(real) == 4 ? (real)fmodf((float)a, (float)p) : (real)fmod((double)a, (double)p); }
static inline double d_modulo(double a, double p) { double r = fmod(a, p); if (r != 0.0 && ((r < 0) != (p < 0))) r += p; return r; }

/* ------------------------------------------------------------------------- */
/* context = the reference's com_mod/par_mod variables the path reads         */
/* ------------------------------------------------------------------------- */
typedef struct {
  /* grid (com_mod.f90:298-299,551-560) */
  int nx, ny, nz, nxmin1, nymin1, nmixz;
  real dx, dy, xlon0, ylat0, dxconst, dyconst;
  int xglobal, nglobal, sglobal;
  real switchnorthg, switchsouthg;
  real northpolemap[9], southpolemap[9];
  real height[ORC_NZMAX];
  int memtime[2], memind[2], lwindinterv;
  /* run switches (com_mod.f90:56-77,112,144,188,589) */
  int ldirect, lsynctime, method, mintime, ifine, turbswitch, cblflag, mdomainfill, lsettling;
  int nspec, drydep, drydepspec[ORC_MAXSPEC];
  real ctl, fine, d_trop, d_strat, turbmesoscale;
  real density[ORC_MAXSPEC], dquer[ORC_MAXSPEC], vsetaver[ORC_MAXSPEC], cunningham[ORC_MAXSPEC];
  real decay[ORC_MAXSPEC];
  /* point_mod.f90:10,20: xmass(numpoint,maxspec), npart(numpoint) -- column-major, species stride = numpoint */
  int numpoint, mquasilag;
  real *xmass_pt;
  int *npart_pt;
  int lage_last;                   /* lage(nageclass)      */
  /* nests (com_mod.f90:464-541) */
  int numbnests;
  int nxn[ORC_MAXNESTS], nyn[ORC_MAXNESTS];
  real xln[ORC_MAXNESTS], yln[ORC_MAXNESTS], xrn[ORC_MAXNESTS], yrn[ORC_MAXNESTS];
  real xresoln[ORC_MAXNESTS], yresoln[ORC_MAXNESTS];
  /* fields */
  const real *uu, *vv, *ww, *rho, *drhodz, *tt, *uupol, *vvpol;
  const real *hmix, *ustar, *wstar, *oli, *tropopause, *vdep;
  const real *uun[ORC_MAXNESTS], *vvn[ORC_MAXNESTS], *wwn[ORC_MAXNESTS], *rhon[ORC_MAXNESTS],
             *drhodzn[ORC_MAXNESTS], *hmixn[ORC_MAXNESTS], *ustarn[ORC_MAXNESTS], *wstarn[ORC_MAXNESTS],
             *olin[ORC_MAXNESTS], *tropopausen[ORC_MAXNESTS], *vdepn[ORC_MAXNESTS];
  /* random table (com_mod.f90:744) */
  real rannumb[ORC_MAXRAND + 64];  /* 1-based; the reference reads past the end on rare CBL re-draws */
  long nan_count, nan_count2;

  /* ---- interpol_mod.f90:7-16 (module-global scratch, persists between particles) */
  real uprof[ORC_NZMAX + 1], vprof[ORC_NZMAX + 1], wprof[ORC_NZMAX + 1];
  real usigprof[ORC_NZMAX + 1], vsigprof[ORC_NZMAX + 1], wsigprof[ORC_NZMAX + 1];
  real rhoprof[ORC_NZMAX + 1], rhogradprof[ORC_NZMAX + 1];
  real u, v, w, usig, vsig, wsig;
  real p1, p2, p3, p4, ddx, ddy, rddx, rddy, dtt, dt1, dt2;
  int ix, jy, ixp, jyp, ngrid, indz, indzp;
  int depoindicator[ORC_MAXSPEC];
  int indzindicator[ORC_NZMAX + 1];
  /* ---- hanna_mod.f90:5-6 */
  real ust, wst, ol, h, zeta, sigu, sigv, tlu, tlv, tlw, sigw, dsigwdz, dsigw2dz;
  /* ---- random_mod.f90 saved state */
  int ran3_iff, ran3_inext, ran3_inextp, ran3_ma[56];
  int gasdev_iset; real gasdev_gset;
  int idummy_init, idummy_adv;     /* initialize.f90:64, advance.f90:120 */
  real settling;                   /* advance.f90:121 (saved) */
  /* 0 (default): the reference's exact serial semantics, including state that leaks from
     one particle to the next through module variables.  1: the two order-dependent leaks
     are replaced by the particle's own values -- what an order-independent (parallel)
     engine computes; see DESIGN.md "deviations" D1/D2. */
  int parallel_semantics;
  int turboff, interpolhmix;   /* com_mod.f90:777-778: compile-time logicals of the reference (both .false. as shipped) */
  /* optional bookkeeping for the tests (orc_set_leak_flags): which particles the two leaks touch.  flags[j] |= 1 when
     particle j takes advance.f90:550 (D1), |= 2 when initialize() runs for it with an ngrid left by its predecessor
     that selects the other wind arrays (polar vs. lat-lon) than the particle's own position would (D2). */
  unsigned char *leak_flags;
  int cur_particle;
  real eps_nxmax;                  /* par_mod nxmax of the build being mirrored (eps = nxmax/3.e5) */
  /* ---- output grid (com_mod.f90:583-586, outg_mod outheight, unc_mod gridunc/drygridunc) */
  int numxgrid, numygrid, numzgrid, maxpointspec_act, nclassunc, nageclass, maxspec_out;
  int lage[8];
  real dxout, dyout, xoutshift, youtshift;
  real outheight[ORC_NZMAX];
  int ind_samp, ioutputforeachrelease, lusekerneloutput;
  int loutnext, loutstep;
  real *gridunc;
  dep_real *drygridunc, *wetgridunc;
  /* ---- nested output grid (com_mod.f90:585-586, unc_mod.f90:24-28) and receptor points (com_mod.f90:658-663) */
  int nested_output, numxgridn, numygridn;
  real dxoutn, dyoutn, xoutshiftn, youtshiftn;
  real *griduncn;
  dep_real *drygriduncn, *wetgriduncn;
  int numreceptor;
  real xreceptor[ORC_MAXRECEPTOR], yreceptor[ORC_MAXRECEPTOR], receptorarea[ORC_MAXRECEPTOR];
  real creceptor[ORC_MAXRECEPTOR * ORC_MAXSPEC];   /* (n, ks), n fastest */
  /* ---- wet deposition (com_mod.f90:139,171-175,379-384,413-419) */
  int wetdepspec[ORC_MAXSPEC], readclouds;
  real weta_gas[ORC_MAXSPEC], wetb_gas[ORC_MAXSPEC], crain_aero[ORC_MAXSPEC], csnow_aero[ORC_MAXSPEC];
  real ccn_aero[ORC_MAXSPEC], in_aero[ORC_MAXSPEC], henry[ORC_MAXSPEC];
  const real *lsprec, *convprec, *tcc, *ctwc;   /* [slot][jy][ix] */
  const real *lsprecn, *convprecn, *tccn, *ttn;  /* nest 1: [slot][jy][ix] / [slot][iz][jy][ix], com_mod.f90:501,518 */
  const signed char *cloudsn;                    /* nest 1: [slot][iz][jy][ix], com_mod.f90:505 */
  const signed char *clouds;                    /* [slot][level][jy][ix] */
  const int *cloudsh;                           /* [slot][jy][ix] */
  long blc_count[ORC_MAXSPEC], inc_count[ORC_MAXSPEC];
  /* ---- backward runs with receptor scavenging (com_mod.f90:591,611: DRYBKDEP, WETBKDEP; readcommand.f90:320-340) */
  int drybkdep, wetbkdep;
  real *zpoint1, *zpoint2;                      /* point_mod zpoint1/zpoint2(numpoint): release height range */
} orc_ctx;

#define F3(f, i, j, k, m) ((f)[(((size_t)((m) - 1) * c->nz + (size_t)((k) - 1)) * c->ny + (size_t)(j)) * c->nx + (size_t)(i)])
#define F2(f, i, j, m)    ((f)[((size_t)((m) - 1) * c->ny + (size_t)(j)) * c->nx + (size_t)(i)])
#define FV(f, i, j, ks, m) ((f)[((((size_t)((m) - 1) * c->nspec + (size_t)((ks) - 1)) * c->ny + (size_t)(j)) * c->nx) + (size_t)(i)])
#define HGT(k) (c->height[(k) - 1])
/* nest l (1-based): compact [slot][level][jyn][ixn] */
#define N3(f, i, j, k, m, l) ((f)[(((size_t)((m) - 1) * c->nz + (size_t)((k) - 1)) * c->nyn[(l) - 1] + (size_t)(j)) * c->nxn[(l) - 1] + (size_t)(i)])
#define N2(f, i, j, m, l) ((f)[((size_t)((m) - 1) * c->nyn[(l) - 1] + (size_t)(j)) * c->nxn[(l) - 1] + (size_t)(i)])
#define NV(f, i, j, ks, m, l) ((f)[((((size_t)((m) - 1) * c->nspec + (size_t)((ks) - 1)) * c->nyn[(l) - 1] + (size_t)(j)) * c->nxn[(l) - 1]) + (size_t)(i)])

/* ------------------------------------------------------------------------- */
/* RNG: random_mod.f90                                                         */
/* ------------------------------------------------------------------------- */
/* random_mod.f90:93-139 -- Knuth subtractive generator, integer state */
static real orc_ran3(orc_ctx *c, int *idum) {
  const int mbig = 1000000000, mseed = 161803398, mz = 0;
  const real fac = K(1.) / (real)mbig;
  int i, ii, k, mj, mk;
  int *ma = c->ran3_ma;
  if (*idum < 0 || c->ran3_iff == 0) {
    c->ran3_iff = 1;
    mj = mseed - abs(*idum);
    mj = mj % mbig;
    ma[55] = mj;
    mk = 1;
    for (i = 1; i <= 54; i++) {
      ii = (21 * i) % 55;
      ma[ii] = mk;
      mk = mj - mk;
      if (mk < mz) mk = mk + mbig;
      mj = ma[ii];
    }
    for (k = 1; k <= 4; k++)
      for (i = 1; i <= 55; i++) {
        ma[i] = ma[i] - ma[1 + (i + 30) % 55];
        if (ma[i] < mz) ma[i] = ma[i] + mbig;
      }
    c->ran3_inext = 0;
    c->ran3_inextp = 31;
    *idum = 1;
  }
  c->ran3_inext++;
  if (c->ran3_inext == 56) c->ran3_inext = 1;
  c->ran3_inextp++;
  if (c->ran3_inextp == 56) c->ran3_inextp = 1;
  mj = ma[c->ran3_inext] - ma[c->ran3_inextp];
  if (mj < mz) mj = mj + mbig;
  ma[c->ran3_inext] = mj;
  return (real)mj * fac;
}

/* random_mod.f90:70-90 -- Box-Muller pair clipped to +-3 */
static void orc_gasdev1(orc_ctx *c, int *idum, real *random1, real *random2) {
  real v1, v2, r, fac;
  do {
    v1 = K(2.) * orc_ran3(c, idum) - K(1.);
    v2 = K(2.) * orc_ran3(c, idum) - K(1.);
    r = v1 * v1 + v2 * v2;
  } while (r >= K(1.0) || r == K(0.0));
  fac = r_sqrt(K(-2.) * r_log(r) / r);
  *random1 = v1 * fac;
  *random2 = v2 * fac;
  if (*random1 < K(-3.)) *random1 = K(-3.);
  if (*random2 < K(-3.)) *random2 = K(-3.);
  if (*random1 > K(3.)) *random1 = K(3.);
  if (*random2 > K(3.)) *random2 = K(3.);
}

/* random_mod.f90:45-67 */
static real orc_gasdev(orc_ctx *c, int *idum) {
  real v1, v2, r, fac;
  if (c->gasdev_iset == 0) {
    do {
      v1 = K(2.) * orc_ran3(c, idum) - K(1.);
      v2 = K(2.) * orc_ran3(c, idum) - K(1.);
      r = v1 * v1 + v2 * v2;
    } while (r >= K(1.0) || r == K(0.0));
    fac = r_sqrt(K(-2.) * r_log(r) / r);
    c->gasdev_gset = v1 * fac;
    c->gasdev_iset = 1;
    return v2 * fac;
  }
  c->gasdev_iset = 0;
  return c->gasdev_gset;
}

/* FLEXPART.f90:47,56-59 -- fill the 1e6-entry table, seed -320 */
static void orc_fill_rannumb(orc_ctx *c) {
  int idummy = -320, i;
  c->ran3_iff = 0;
  for (i = 1; i <= ORC_MAXRAND - 1; i += 2) orc_gasdev1(c, &idummy, &c->rannumb[i], &c->rannumb[i + 1]);
  orc_gasdev1(c, &idummy, &c->rannumb[ORC_MAXRAND], &c->rannumb[ORC_MAXRAND - 1]);
}

/* ------------------------------------------------------------------------- */
/* interpolation: interpol_all / misslev / wind / wind_short / vdep            */
/* ------------------------------------------------------------------------- */
/* the horizontal weights block common to interpol_all.f90:57-71,
   interpol_wind.f90:56-70, interpol_wind_short.f90:48-62 */
static void orc_hweights(orc_ctx *c, int itime, real xt, real yt) {
  c->ddx = xt - (real)c->ix;
  c->ddy = yt - (real)c->jy;
  c->rddx = K(1.) - c->ddx;
  c->rddy = K(1.) - c->ddy;
  c->p1 = c->rddx * c->rddy;
  c->p2 = c->ddx * c->rddy;
  c->p3 = c->rddx * c->ddy;
  c->p4 = c->ddx * c->ddy;
  c->dt1 = (real)(itime - c->memtime[0]);
  c->dt2 = (real)(c->memtime[1] - itime);
  c->dtt = K(1.) / (c->dt1 + c->dt2);
}

#define BIL(f, n, m) (c->p1 * F3(f, ix, jy, n, m) + c->p2 * F3(f, ixp, jy, n, m) + c->p3 * F3(f, ix, jyp, n, m) + c->p4 * F3(f, ixp, jyp, n, m))
/* Fortran evaluates  usl=usl+a+b+c+d  left to right: keep that association */
#define SUM4(acc, f, n, m) ((((acc) + F3(f, ix, jy, n, m)) + F3(f, ixp, jy, n, m)) + F3(f, ix, jyp, n, m)) + F3(f, ixp, jyp, n, m)
#define SQ4(acc, f, n, m) ((((acc) + F3(f, ix, jy, n, m) * F3(f, ix, jy, n, m)) + F3(f, ixp, jy, n, m) * F3(f, ixp, jy, n, m)) + F3(f, ix, jyp, n, m) * F3(f, ix, jyp, n, m)) + F3(f, ixp, jyp, n, m) * F3(f, ixp, jyp, n, m)

/* one level of profiles: interpol_all.f90:135-240 loop body == interpol_misslev.f90:56-159 */
static void orc_profile_level(orc_ctx *c, int n) {
  const real eps = K(1.0e-30);
  real y1[2], y2[2], y3[2], rho1[2], rhograd1[2];
  real usl = 0, vsl = 0, wsl = 0, usq = 0, vsq = 0, wsq = 0, xaux;
  int m, ix = c->ix, jy = c->jy, ixp = c->ixp, jyp = c->jyp;
  const real *fu = c->ngrid < 0 ? c->uupol : c->uu;
  const real *fv = c->ngrid < 0 ? c->vvpol : c->vv;
  for (m = 0; m < 2; m++) {
    int indexh = c->memind[m];
    y1[m] = BIL(fu, n, indexh);
    y2[m] = BIL(fv, n, indexh);
    usl = SUM4(usl, fu, n, indexh);
    vsl = SUM4(vsl, fv, n, indexh);
    usq = SQ4(usq, fu, n, indexh);
    vsq = SQ4(vsq, fv, n, indexh);
    y3[m] = BIL(c->ww, n, indexh);
    rhograd1[m] = BIL(c->drhodz, n, indexh);
    rho1[m] = BIL(c->rho, n, indexh);
    wsl = SUM4(wsl, c->ww, n, indexh);
    wsq = SQ4(wsq, c->ww, n, indexh);
  }
  c->uprof[n] = (y1[0] * c->dt2 + y1[1] * c->dt1) * c->dtt;
  c->vprof[n] = (y2[0] * c->dt2 + y2[1] * c->dt1) * c->dtt;
  c->wprof[n] = (y3[0] * c->dt2 + y3[1] * c->dt1) * c->dtt;
  c->rhoprof[n] = (rho1[0] * c->dt2 + rho1[1] * c->dt1) * c->dtt;
  c->rhogradprof[n] = (rhograd1[0] * c->dt2 + rhograd1[1] * c->dt1) * c->dtt;
  c->indzindicator[n] = 0;
  /* 8-point standard deviations: interpol_all.f90:218-238 */
  xaux = usq - usl * usl / K(8.);
  c->usigprof[n] = xaux < eps ? K(0.) : r_sqrt(xaux / K(7.));
  xaux = vsq - vsl * vsl / K(8.);
  c->vsigprof[n] = xaux < eps ? K(0.) : r_sqrt(xaux / K(7.));
  xaux = wsq - wsl * wsl / K(8.);
  c->wsigprof[n] = xaux < eps ? K(0.) : r_sqrt(xaux / K(7.));
}

/* the linear level search, e.g. interpol_all.f90:118-125 (indz keeps its old
   value if zt is above the top level -- the reference does the same) */
static void orc_find_level(orc_ctx *c, real zt, int set_indzp) {
  int i;
  for (i = 2; i <= c->nz; i++)
    if (HGT(i) > zt) {
      c->indz = i - 1;
      if (set_indzp) c->indzp = i;
      break;
    }
}

/* interpol_all.f90:57-240 */
static void orc_interpol_all(orc_ctx *c, int itime, real xt, real yt, real zt) {
  real ust1[2], wst1[2], oli1[2], oliaux;
  int m, n, ix, jy, ixp, jyp;
  orc_hweights(c, itime, xt, yt);
  ix = c->ix; jy = c->jy; ixp = c->ixp; jyp = c->jyp;
  for (m = 0; m < 2; m++) {
    int indexh = c->memind[m];
    ust1[m] = c->p1 * F2(c->ustar, ix, jy, indexh) + c->p2 * F2(c->ustar, ixp, jy, indexh) + c->p3 * F2(c->ustar, ix, jyp, indexh) + c->p4 * F2(c->ustar, ixp, jyp, indexh);
    wst1[m] = c->p1 * F2(c->wstar, ix, jy, indexh) + c->p2 * F2(c->wstar, ixp, jy, indexh) + c->p3 * F2(c->wstar, ix, jyp, indexh) + c->p4 * F2(c->wstar, ixp, jyp, indexh);
    oli1[m] = c->p1 * F2(c->oli, ix, jy, indexh) + c->p2 * F2(c->oli, ixp, jy, indexh) + c->p3 * F2(c->oli, ix, jyp, indexh) + c->p4 * F2(c->oli, ixp, jyp, indexh);
  }
  c->ust = (ust1[0] * c->dt2 + ust1[1] * c->dt1) * c->dtt;
  c->wst = (wst1[0] * c->dt2 + wst1[1] * c->dt1) * c->dtt;
  oliaux = (oli1[0] * c->dt2 + oli1[1] * c->dt1) * c->dtt;
  c->ol = oliaux != K(0.) ? K(1.) / oliaux : K(99999.);
  orc_find_level(c, zt, 1);
  for (n = c->indz; n <= c->indzp; n++) orc_profile_level(c, n);
}

/* interpol_wind.f90:56-214 (with_sigma=1) and interpol_wind_short.f90:48-140 (with_sigma=0) */
static void orc_interpol_wind(orc_ctx *c, int itime, real xt, real yt, real zt, int with_sigma) {
  const real eps = K(1.0e-30);
  real dz1, dz2, dz, u1[2], v1[2], w1[2], uh[2], vh[2], wh[2];
  real usl = 0, vsl = 0, wsl = 0, usq = 0, vsq = 0, wsq = 0, xaux;
  int m, n, ix, jy, ixp, jyp;
  const real *fu = c->ngrid < 0 ? c->uupol : c->uu;
  const real *fv = c->ngrid < 0 ? c->vvpol : c->vv;
  orc_hweights(c, itime, xt, yt);
  ix = c->ix; jy = c->jy; ixp = c->ixp; jyp = c->jyp;
  orc_find_level(c, zt, 0);
  dz = K(1.) / (HGT(c->indz + 1) - HGT(c->indz));
  dz1 = (zt - HGT(c->indz)) * dz;
  dz2 = (HGT(c->indz + 1) - zt) * dz;
  for (m = 0; m < 2; m++) {
    int indexh = c->memind[m];
    for (n = 0; n < 2; n++) {
      int indzh = c->indz + n;
      u1[n] = BIL(fu, indzh, indexh);
      v1[n] = BIL(fv, indzh, indexh);
      if (with_sigma) {
        usl = SUM4(usl, fu, indzh, indexh);
        vsl = SUM4(vsl, fv, indzh, indexh);
        usq = SQ4(usq, fu, indzh, indexh);
        vsq = SQ4(vsq, fv, indzh, indexh);
      }
      w1[n] = BIL(c->ww, indzh, indexh);
      if (with_sigma) {
        wsl = SUM4(wsl, c->ww, indzh, indexh);
        wsq = SQ4(wsq, c->ww, indzh, indexh);
      }
    }
    uh[m] = dz2 * u1[0] + dz1 * u1[1];
    vh[m] = dz2 * v1[0] + dz1 * v1[1];
    wh[m] = dz2 * w1[0] + dz1 * w1[1];
  }
  c->u = (uh[0] * c->dt2 + uh[1] * c->dt1) * c->dtt;
  c->v = (vh[0] * c->dt2 + vh[1] * c->dt1) * c->dtt;
  c->w = (wh[0] * c->dt2 + wh[1] * c->dt1) * c->dtt;
  if (with_sigma) {  /* interpol_wind.f90:194-214, 16 points */
    xaux = usq - usl * usl / K(16.);
    c->usig = xaux < eps ? K(0.) : r_sqrt(xaux / K(15.));
    xaux = vsq - vsl * vsl / K(16.);
    c->vsig = xaux < eps ? K(0.) : r_sqrt(xaux / K(15.));
    xaux = wsq - wsl * wsl / K(16.);
    c->wsig = xaux < eps ? K(0.) : r_sqrt(xaux / K(15.));
  }
}

/* interpol_vdep.f90:39-54 */
static void orc_interpol_vdep(orc_ctx *c, int level, real *vdepo) {
  real y[2];
  int m, ix = c->ix, jy = c->jy, ixp = c->ixp, jyp = c->jyp;
  for (m = 0; m < 2; m++) {
    int indexh = c->memind[m];
    y[m] = c->p1 * FV(c->vdep, ix, jy, level, indexh) + c->p2 * FV(c->vdep, ixp, jy, level, indexh) + c->p3 * FV(c->vdep, ix, jyp, level, indexh) + c->p4 * FV(c->vdep, ixp, jyp, level, indexh);
  }
  *vdepo = (y[0] * c->dt2 + y[1] * c->dt1) * c->dtt;
  c->depoindicator[level - 1] = 0;
}

/* ---- nested-grid variants: interpol_all_nests.f90, interpol_misslev_nests.f90,
   interpol_wind_nests.f90, interpol_wind_short_nests.f90, interpol_vdep_nests.f90.
   Same arithmetic on the 5-D nest arrays; no polar branch. ------------------------------ */
#define BILN(f, n, m) (c->p1 * N3(f, ix, jy, n, m, l) + c->p2 * N3(f, ixp, jy, n, m, l) + c->p3 * N3(f, ix, jyp, n, m, l) + c->p4 * N3(f, ixp, jyp, n, m, l))
#define SUM4N(acc, f, n, m) ((((acc) + N3(f, ix, jy, n, m, l)) + N3(f, ixp, jy, n, m, l)) + N3(f, ix, jyp, n, m, l)) + N3(f, ixp, jyp, n, m, l)
#define SQ4N(acc, f, n, m) ((((acc) + N3(f, ix, jy, n, m, l) * N3(f, ix, jy, n, m, l)) + N3(f, ixp, jy, n, m, l) * N3(f, ixp, jy, n, m, l)) + N3(f, ix, jyp, n, m, l) * N3(f, ix, jyp, n, m, l)) + N3(f, ixp, jyp, n, m, l) * N3(f, ixp, jyp, n, m, l)

static void orc_profile_level_nests(orc_ctx *c, int n) {   /* interpol_misslev_nests.f90:56-129 == interpol_all_nests.f90 loop body */
  const real eps = K(1.0e-30);
  real y1[2], y2[2], y3[2], rho1[2], rhograd1[2];
  real usl = 0, vsl = 0, wsl = 0, usq = 0, vsq = 0, wsq = 0, xaux;
  int m, ix = c->ix, jy = c->jy, ixp = c->ixp, jyp = c->jyp, l = c->ngrid;
  for (m = 0; m < 2; m++) {
    int indexh = c->memind[m];
    y1[m] = BILN(c->uun[l - 1], n, indexh);
    y2[m] = BILN(c->vvn[l - 1], n, indexh);
    y3[m] = BILN(c->wwn[l - 1], n, indexh);
    rhograd1[m] = BILN(c->drhodzn[l - 1], n, indexh);
    rho1[m] = BILN(c->rhon[l - 1], n, indexh);
    usl = SUM4N(usl, c->uun[l - 1], n, indexh);
    vsl = SUM4N(vsl, c->vvn[l - 1], n, indexh);
    wsl = SUM4N(wsl, c->wwn[l - 1], n, indexh);
    usq = SQ4N(usq, c->uun[l - 1], n, indexh);
    vsq = SQ4N(vsq, c->vvn[l - 1], n, indexh);
    wsq = SQ4N(wsq, c->wwn[l - 1], n, indexh);
  }
  c->uprof[n] = (y1[0] * c->dt2 + y1[1] * c->dt1) * c->dtt;
  c->vprof[n] = (y2[0] * c->dt2 + y2[1] * c->dt1) * c->dtt;
  c->wprof[n] = (y3[0] * c->dt2 + y3[1] * c->dt1) * c->dtt;
  c->rhoprof[n] = (rho1[0] * c->dt2 + rho1[1] * c->dt1) * c->dtt;
  c->rhogradprof[n] = (rhograd1[0] * c->dt2 + rhograd1[1] * c->dt1) * c->dtt;
  c->indzindicator[n] = 0;
  xaux = usq - usl * usl / K(8.);
  c->usigprof[n] = xaux < eps ? K(0.) : r_sqrt(xaux / K(7.));
  xaux = vsq - vsl * vsl / K(8.);
  c->vsigprof[n] = xaux < eps ? K(0.) : r_sqrt(xaux / K(7.));
  xaux = wsq - wsl * wsl / K(8.);
  c->wsigprof[n] = xaux < eps ? K(0.) : r_sqrt(xaux / K(7.));
}

static void orc_interpol_all_nests(orc_ctx *c, int itime, real xt, real yt, real zt) {   /* interpol_all_nests.f90:57-219 */
  real ust1[2], wst1[2], oli1[2], oliaux;
  int m, n, ix, jy, ixp, jyp, l = c->ngrid;
  orc_hweights(c, itime, xt, yt);
  ix = c->ix; jy = c->jy; ixp = c->ixp; jyp = c->jyp;
  for (m = 0; m < 2; m++) {
    int indexh = c->memind[m];
    ust1[m] = c->p1 * N2(c->ustarn[l - 1], ix, jy, indexh, l) + c->p2 * N2(c->ustarn[l - 1], ixp, jy, indexh, l) + c->p3 * N2(c->ustarn[l - 1], ix, jyp, indexh, l) + c->p4 * N2(c->ustarn[l - 1], ixp, jyp, indexh, l);
    wst1[m] = c->p1 * N2(c->wstarn[l - 1], ix, jy, indexh, l) + c->p2 * N2(c->wstarn[l - 1], ixp, jy, indexh, l) + c->p3 * N2(c->wstarn[l - 1], ix, jyp, indexh, l) + c->p4 * N2(c->wstarn[l - 1], ixp, jyp, indexh, l);
    oli1[m] = c->p1 * N2(c->olin[l - 1], ix, jy, indexh, l) + c->p2 * N2(c->olin[l - 1], ixp, jy, indexh, l) + c->p3 * N2(c->olin[l - 1], ix, jyp, indexh, l) + c->p4 * N2(c->olin[l - 1], ixp, jyp, indexh, l);
  }
  c->ust = (ust1[0] * c->dt2 + ust1[1] * c->dt1) * c->dtt;
  c->wst = (wst1[0] * c->dt2 + wst1[1] * c->dt1) * c->dtt;
  oliaux = (oli1[0] * c->dt2 + oli1[1] * c->dt1) * c->dtt;
  c->ol = oliaux != K(0.) ? K(1.) / oliaux : K(99999.);
  orc_find_level(c, zt, 1);
  for (n = c->indz; n <= c->indz + 1; n++) orc_profile_level_nests(c, n);
}

static void orc_interpol_wind_nests(orc_ctx *c, int itime, real xt, real yt, real zt, int with_sigma) {   /* interpol_wind_nests.f90 / interpol_wind_short_nests.f90 */
  const real eps = K(1.0e-30);
  real dz1, dz2, dz, u1[2], v1[2], w1[2], uh[2], vh[2], wh[2];
  real usl = 0, vsl = 0, wsl = 0, usq = 0, vsq = 0, wsq = 0, xaux;
  int m, n, ix, jy, ixp, jyp, l = c->ngrid;
  orc_hweights(c, itime, xt, yt);
  ix = c->ix; jy = c->jy; ixp = c->ixp; jyp = c->jyp;
  orc_find_level(c, zt, 0);
  dz = K(1.) / (HGT(c->indz + 1) - HGT(c->indz));
  dz1 = (zt - HGT(c->indz)) * dz;
  dz2 = (HGT(c->indz + 1) - zt) * dz;
  for (m = 0; m < 2; m++) {
    int indexh = c->memind[m];
    for (n = 0; n < 2; n++) {
      int indzh = c->indz + n;
      u1[n] = BILN(c->uun[l - 1], indzh, indexh);
      v1[n] = BILN(c->vvn[l - 1], indzh, indexh);
      w1[n] = BILN(c->wwn[l - 1], indzh, indexh);
      if (with_sigma) {
        usl = SUM4N(usl, c->uun[l - 1], indzh, indexh);
        vsl = SUM4N(vsl, c->vvn[l - 1], indzh, indexh);
        wsl = SUM4N(wsl, c->wwn[l - 1], indzh, indexh);
        usq = SQ4N(usq, c->uun[l - 1], indzh, indexh);
        vsq = SQ4N(vsq, c->vvn[l - 1], indzh, indexh);
        wsq = SQ4N(wsq, c->wwn[l - 1], indzh, indexh);
      }
    }
    uh[m] = dz2 * u1[0] + dz1 * u1[1];
    vh[m] = dz2 * v1[0] + dz1 * v1[1];
    wh[m] = dz2 * w1[0] + dz1 * w1[1];
  }
  c->u = (uh[0] * c->dt2 + uh[1] * c->dt1) * c->dtt;
  c->v = (vh[0] * c->dt2 + vh[1] * c->dt1) * c->dtt;
  c->w = (wh[0] * c->dt2 + wh[1] * c->dt1) * c->dtt;
  if (with_sigma) {
    xaux = usq - usl * usl / K(16.);
    c->usig = xaux < eps ? K(0.) : r_sqrt(xaux / K(15.));
    xaux = vsq - vsl * vsl / K(16.);
    c->vsig = xaux < eps ? K(0.) : r_sqrt(xaux / K(15.));
    xaux = wsq - wsl * wsl / K(16.);
    c->wsig = xaux < eps ? K(0.) : r_sqrt(xaux / K(15.));
  }
}

static void orc_interpol_vdep_nests(orc_ctx *c, int level, real *vdepo) {   /* interpol_vdep_nests.f90:39-53 */
  real y[2];
  int m, ix = c->ix, jy = c->jy, ixp = c->ixp, jyp = c->jyp, l = c->ngrid;
  for (m = 0; m < 2; m++) {
    int indexh = c->memind[m];
    y[m] = c->p1 * NV(c->vdepn[l - 1], ix, jy, level, indexh, l) + c->p2 * NV(c->vdepn[l - 1], ixp, jy, level, indexh, l) +
           c->p3 * NV(c->vdepn[l - 1], ix, jyp, level, indexh, l) + c->p4 * NV(c->vdepn[l - 1], ixp, jyp, level, indexh, l);
  }
  *vdepo = (y[0] * c->dt2 + y[1] * c->dt1) * c->dtt;
  c->depoindicator[level - 1] = 0;
}

/* ------------------------------------------------------------------------- */
/* turbulence parameterisation: hanna.f90 / hanna1.f90 / hanna_short.f90       */
/* ------------------------------------------------------------------------- */
/* the shared tlw branch hanna.f90:78-84 */
static real orc_tlw_unstable(orc_ctx *c, real z) {
  if (z < r_abs(c->ol)) return K(0.1) * z / (c->sigw * (K(0.55) - K(0.38) * r_abs(z / c->ol)));
  if (c->zeta < K(0.1)) return K(0.59) * z / c->sigw;
  return K(0.15) * c->h / c->sigw * (K(1.) - r_exp(K(-5) * c->zeta));
}

/* hanna.f90:41-106 */
static void orc_hanna(orc_ctx *c, real z) {
  real corr;
  if (c->h / r_abs(c->ol) < K(1.)) {                       /* neutral */
    c->ust = r_max(K(1.e-4), c->ust);
    corr = z / c->ust;
    c->sigu = K(1.e-2) + K(2.0) * c->ust * r_exp(K(-3.e-4) * corr);
    c->sigw = K(1.3) * c->ust * r_exp(K(-2.e-4) * corr);
    c->dsigwdz = K(-2.e-4) * c->sigw;
    c->sigw = c->sigw + K(1.e-2);
    c->sigv = c->sigw;
    c->tlu = K(0.5) * z / c->sigw / (K(1.) + K(1.5e-3) * corr);
    c->tlv = c->tlu;
    c->tlw = c->tlu;
  } else if (c->ol < K(0.)) {                              /* unstable */
    c->sigu = K(1.e-2) + c->ust * r_pow(K(12) - K(0.5) * c->h / c->ol, K(0.33333));
    c->sigv = c->sigu;
    c->sigw = r_sqrt(K(1.2) * (c->wst * c->wst) * (K(1.) - K(.9) * c->zeta) * r_pow(c->zeta, K(0.66666)) + (K(1.8) - K(1.4) * c->zeta) * (c->ust * c->ust)) + K(1.e-2);
    c->dsigwdz = K(0.5) / c->sigw / c->h * (K(-1.4) * (c->ust * c->ust) + (c->wst * c->wst) * (K(0.8) * r_pow(r_max(c->zeta, K(1.e-3)), K(-.33333)) - K(1.8) * r_pow(c->zeta, K(0.66666))));
    c->tlu = K(0.15) * c->h / c->sigu;
    c->tlv = c->tlu;
    c->tlw = orc_tlw_unstable(c, z);
  } else {                                                 /* stable */
    c->sigu = K(1.e-2) + K(2.) * c->ust * (K(1.) - c->zeta);
    c->sigv = K(1.e-2) + K(1.3) * c->ust * (K(1.) - c->zeta);
    c->sigw = c->sigv;
    c->dsigwdz = K(-1.3) * c->ust / c->h;
    c->tlu = K(0.15) * c->h / c->sigu * r_sqrt(c->zeta);
    c->tlv = K(0.467) * c->tlu;
    c->tlw = K(0.1) * c->h / c->sigw * r_pow(c->zeta, K(0.8));
  }
  c->tlu = r_max(K(10.), c->tlu);
  c->tlv = r_max(K(10.), c->tlv);
  c->tlw = r_max(K(30.), c->tlw);
  if (c->dsigwdz == K(0.)) c->dsigwdz = K(1.e-10);
}

/* hanna1.f90:41-129 */
static void orc_hanna1(orc_ctx *c, real z) {
  real s1, s2;
  if (c->h / r_abs(c->ol) < K(1.)) {
    c->ust = r_max(K(1.e-4), c->ust);
    c->sigu = K(2.0) * c->ust * r_exp(K(-3.e-4) * z / c->ust);
    c->sigu = r_max(c->sigu, K(1.e-5));
    c->sigv = K(1.3) * c->ust * r_exp(K(-2.e-4) * z / c->ust);
    c->sigv = r_max(c->sigv, K(1.e-5));
    c->sigw = c->sigv;
    c->dsigw2dz = K(-6.76e-4) * c->ust * r_exp(K(-4.e-4) * z / c->ust);
    c->tlu = K(0.5) * z / c->sigw / (K(1.) + K(1.5e-3) * z / c->ust);
    c->tlv = c->tlu;
    c->tlw = c->tlu;
  } else if (c->ol < K(0.)) {
    c->sigu = c->ust * r_pow(K(12) - K(0.5) * c->h / c->ol, K(0.33333));
    c->sigu = r_max(c->sigu, K(1.e-6));
    c->sigv = c->sigu;
    if (c->zeta < K(0.03)) {
      c->sigw = K(0.96) * c->wst * r_pow(K(3) * c->zeta - c->ol / c->h, K(0.33333));
      c->dsigw2dz = K(1.8432) * c->wst * c->wst / c->h * r_pow(K(3) * c->zeta - c->ol / c->h, K(-0.33333));
    } else if (c->zeta < K(0.4)) {
      s1 = K(0.96) * r_pow(K(3) * c->zeta - c->ol / c->h, K(0.33333));
      s2 = K(0.763) * r_pow(c->zeta, K(0.175));
      if (s1 < s2) {
        c->sigw = c->wst * s1;
        c->dsigw2dz = K(1.8432) * c->wst * c->wst / c->h * r_pow(K(3) * c->zeta - c->ol / c->h, K(-0.33333));
      } else {
        c->sigw = c->wst * s2;
        c->dsigw2dz = K(0.203759) * c->wst * c->wst / c->h * r_pow(c->zeta, K(-0.65));
      }
    } else if (c->zeta < K(0.96)) {
      c->sigw = K(0.722) * c->wst * r_pow(K(1) - c->zeta, K(0.207));
      c->dsigw2dz = K(-.215812) * c->wst * c->wst / c->h * r_pow(K(1) - c->zeta, K(-0.586));
    } else if (c->zeta < K(1.00)) {
      c->sigw = K(0.37) * c->wst;
      c->dsigw2dz = K(0.);
    }   /* zeta >= 1: sigw, dsigw2dz keep their previous (module) values */
    c->sigw = r_max(c->sigw, K(1.e-6));
    c->tlu = K(0.15) * c->h / c->sigu;
    c->tlv = c->tlu;
    c->tlw = orc_tlw_unstable(c, z);
  } else {
    c->sigu = K(2.) * c->ust * (K(1.) - c->zeta);
    c->sigv = K(1.3) * c->ust * (K(1.) - c->zeta);
    c->sigu = r_max(c->sigu, K(1.e-6));
    c->sigv = r_max(c->sigv, K(1.e-6));
    c->sigw = c->sigv;
    c->dsigw2dz = K(3.38) * c->ust * c->ust * (c->zeta - K(1.)) / c->h;
    c->tlu = K(0.15) * c->h / c->sigu * r_sqrt(c->zeta);
    c->tlv = K(0.467) * c->tlu;
    c->tlw = K(0.1) * c->h / c->sigw * r_pow(c->zeta, K(0.8));
  }
  c->tlu = r_max(K(10.), c->tlu);
  c->tlv = r_max(K(10.), c->tlv);
  c->tlw = r_max(K(30.), c->tlw);
}

/* hanna_short.f90:41-92 */
static void orc_hanna_short(orc_ctx *c, real z) {
  if (c->h / r_abs(c->ol) < K(1.)) {
    c->ust = r_max(K(1.e-4), c->ust);
    c->sigw = K(1.3) * r_exp(K(-2.e-4) * z / c->ust);
    c->dsigwdz = K(-2.e-4) * c->sigw;
    c->sigw = c->sigw * c->ust + K(1.e-2);
    c->tlw = K(0.5) * z / c->sigw / (K(1.) + K(1.5e-3) * z / c->ust);
  } else if (c->ol < K(0.)) {
    c->sigw = r_sqrt(K(1.2) * (c->wst * c->wst) * (K(1.) - K(.9) * c->zeta) * r_pow(c->zeta, K(0.66666)) + (K(1.8) - K(1.4) * c->zeta) * (c->ust * c->ust)) + K(1.e-2);
    c->dsigwdz = K(0.5) / c->sigw / c->h * (K(-1.4) * (c->ust * c->ust) + (c->wst * c->wst) * (K(0.8) * r_pow(r_max(c->zeta, K(1.e-3)), K(-.33333)) - K(1.8) * r_pow(c->zeta, K(0.66666))));
    c->tlw = orc_tlw_unstable(c, z);
  } else {
    c->sigw = K(1.e-2) + K(1.3) * c->ust * (K(1.) - c->zeta);
    c->dsigwdz = K(-1.3) * c->ust / c->h;
    c->tlw = K(0.1) * c->h / c->sigw * r_pow(c->zeta, K(0.8));
  }
  c->tlu = r_max(K(10.), c->tlu);
  c->tlv = r_max(K(10.), c->tlv);
  c->tlw = r_max(K(30.), c->tlw);
  if (c->dsigwdz == K(0.)) c->dsigwdz = K(1.e-10);
}

/* ------------------------------------------------------------------------- */
/* skewed CBL turbulence: cbl.f90, re_initialize_particle.f90,                 */
/* initialize_cbl_vel.f90                                                      */
/* ------------------------------------------------------------------------- */
#define PI_PAR K(3.14159265)   /* par_mod.f90:59 */

/* cbl.f90:220-234 */
static real orc_cuberoot(real x) { return r_sign(r_pow(r_abs(x), K(0.333333333)), x); }

/* cbl.f90:70-210 */
static void orc_cbl(orc_ctx *c, real wp, real zp, real ust, real wst, real h, real rhoa, real rhograd,
                    real sigmaw, real dsigmawdz, real tlw, real *ptot_o, real *Q_o, real *phi_o,
                    real *ath, real *bth, real ol, int *flagrein) {
  const real usurad2 = K(0.7071067812), usurad2p = K(0.3989422804), C0 = K(3), costluar4 = K(0.66667), eps = K(0.000001);
  real dens, ddens, fluarw, fluarw2, w3, w2, dw3, dw2, wb, wa, deltawa, deltawb, wold, wold2;
  real pa, pb, alfa, Phi, Q, ptot, timedir, transition, aperfa, aperfb;
  real z, skew, skew2, radw2, rluarw, xluarw, aluarw, bluarw, sigmawa, sigmawb, dskew, dradw2, dfluarw;
  real drluarw, dxluarw, daluarw, dbluarw, dsigmawa, dsigmawb, dwa, dwb, sigmawa2, sigmawb2;
  real a1, a3, t1;
  (void)ust;
  dens = rhoa;
  ddens = rhograd;
  timedir = (real)c->ldirect;
  z = zp / h;
  transition = K(1.);
  if (-h / ol < K(15)) transition = r_sin(((-h / ol + K(10.)) / K(10.)) * PI_PAR) / K(2.) + K(0.5);
  w2 = sigmaw * sigmaw;
  dw2 = K(2.) * sigmaw * dsigmawdz;
  alfa = K(2.) * w2 / (C0 * tlw);
  wold = timedir * wp;
  w3 = (K(1.2) * z * r_pow(K(1.) - z, K(1.5)) + eps) * (wst * wst * wst) * transition;
  dw3 = (K(1.2) * (r_pow(K(1.) - z, K(1.5)) + z * K(1.5) * r_pow(K(1.) - z, K(0.5)) * K(-1.))) * (wst * wst * wst) * (K(1.) / h) * transition;
  skew = w3 / r_pow(w2, K(1.5));
  skew2 = skew * skew;
  dskew = (dw3 * r_pow(w2, K(1.5)) - w3 * K(1.5) * r_pow(w2, K(0.5)) * dw2) / (w2 * w2 * w2);
  radw2 = r_pow(w2, K(0.5));
  dradw2 = K(0.5) * r_pow(w2, K(-0.5)) * dw2;
  fluarw = costluar4 * orc_cuberoot(skew);
  fluarw2 = fluarw * fluarw;
  if (skew != K(0)) {
    dfluarw = costluar4 * (K(1.) / K(3.)) * orc_cuberoot(r_pow(skew, K(-2.))) * dskew;
    a1 = K(1.) + fluarw2;
    a3 = K(3.) + fluarw2;
    rluarw = r_pow(a1, K(3.)) * skew2 / (r_pow(a3, K(2.)) * fluarw2);
    xluarw = r_pow(a1, K(1.5)) * skew / (a3 * fluarw);
    drluarw = (((K(3.) * (a1 * a1) * (K(2.) * fluarw * dfluarw) * skew2) + (a1 * a1 * a1) * K(2.) * skew * dskew) * r_pow(a3, K(2.)) * fluarw2 -
               (a1 * a1 * a1) * skew2 * ((K(2.) * a3 * (K(2.) * fluarw * dfluarw) * fluarw2) + (a3 * a3) * K(2.) * fluarw * dfluarw)) /
              ((r_pow(a3, K(2.)) * fluarw2) * (r_pow(a3, K(2.)) * fluarw2));
    dxluarw = (((K(1.5) * r_pow(a1, K(0.5)) * (K(2.) * fluarw * dfluarw) * skew) + r_pow(a1, K(1.5)) * dskew) * a3 * fluarw -
               r_pow(a1, K(1.5)) * skew * (K(3.) * dfluarw + K(3) * fluarw2 * dfluarw)) /
              ((a3 * fluarw) * (a3 * fluarw));
  } else {
    dfluarw = K(0.); rluarw = K(0.); drluarw = K(0.); xluarw = K(0.); dxluarw = K(0.);
  }
  aluarw = K(0.5) * (K(1.) - xluarw / r_pow(K(4.) + rluarw, K(0.5)));
  bluarw = K(1.) - aluarw;
  daluarw = K(-0.5) * ((dxluarw * r_pow(K(4.) + rluarw, K(0.5))) - (K(0.5) * xluarw * r_pow(K(4.) + rluarw, K(-0.5)) * drluarw)) / (K(4.) + rluarw);
  dbluarw = -daluarw;
  sigmawa = radw2 * r_pow(bluarw / (aluarw * (K(1.) + fluarw2)), K(0.5));
  sigmawb = radw2 * r_pow(aluarw / (bluarw * (K(1.) + fluarw2)), K(0.5));
  t1 = aluarw * (K(1.) + fluarw2);
  dsigmawa = dradw2 * r_pow(bluarw / t1, K(0.5)) +
             radw2 * ((K(0.5) * r_pow(bluarw / t1, K(-0.5))) *
                      ((dbluarw * t1 - bluarw * (daluarw * (K(1.) + fluarw2) + aluarw * K(2.) * fluarw * dfluarw)) / (t1 * t1)));
  t1 = bluarw * (K(1.) + fluarw2);
  dsigmawb = dradw2 * r_pow(aluarw / t1, K(0.5)) +
             radw2 * ((K(0.5) * r_pow(aluarw / t1, K(-0.5))) *
                      ((daluarw * t1 - aluarw * (dbluarw * (K(1.) + fluarw2) + bluarw * K(2.) * fluarw * dfluarw)) / (t1 * t1)));
  wa = fluarw * sigmawa;
  wb = fluarw * sigmawb;
  dwa = dfluarw * sigmawa + fluarw * dsigmawa;
  dwb = dfluarw * sigmawb + fluarw * dsigmawb;
  deltawa = wold - wa;
  deltawb = wold + wb;
  wold2 = wold * wold;
  sigmawa2 = sigmawa * sigmawa;
  sigmawb2 = sigmawb * sigmawb;
  if (r_abs(deltawa) > K(6.) * sigmawa && r_abs(deltawb) > K(6.) * sigmawb) *flagrein = 1;
  pa = (usurad2p * (K(1.) / sigmawa)) * r_exp(-(K(0.5) * ((deltawa / sigmawa) * (deltawa / sigmawa))));
  pb = (usurad2p * (K(1.) / sigmawb)) * r_exp(-(K(0.5) * ((deltawb / sigmawb) * (deltawb / sigmawb))));
  ptot = dens * aluarw * pa + dens * bluarw * pb;
  aperfa = deltawa * usurad2 / sigmawa;
  aperfb = deltawb * usurad2 / sigmawb;
  Phi = K(-0.5) * (aluarw * dens * dwa + dens * wa * daluarw + aluarw * wa * ddens) * r_erf(aperfa) +
        sigmawa * (aluarw * dens * dsigmawa * (wold2 / sigmawa2 + K(1.)) + sigmawa * dens * daluarw + sigmawa * ddens * aluarw +
                   aluarw * wold * dens / sigmawa2 * (sigmawa * dwa - wa * dsigmawa)) * pa +
        K(0.5) * (bluarw * dens * dwb + wb * dens * dbluarw + wb * bluarw * ddens) * r_erf(aperfb) +
        sigmawb * (bluarw * dens * dsigmawb * (wold2 / sigmawb2 + K(1.)) + sigmawb * dens * dbluarw + sigmawb * ddens * bluarw +
                   bluarw * wold * dens / sigmawb2 * (-sigmawb * dwb + wb * dsigmawb)) * pb;
  Q = timedir * ((aluarw * dens * deltawa / sigmawa2) * pa + (bluarw * dens * deltawb / sigmawb2) * pb);
  *ath = (K(1.) / ptot) * (-(C0 / K(2.)) * alfa * Q + Phi);
  *bth = r_sqrt(C0 * alfa);
  *ptot_o = ptot; *Q_o = Q; *phi_o = Phi;
}

/* the pdf set-up shared by re_initialize_particle.f90:47-70 and initialize_cbl_vel.f90:46-73 */
static void orc_cbl_pdf(orc_ctx *c, real zp, real wst, real h, real sigmaw, real ol,
                        real *aluarw, real *sigmawa, real *sigmawb, real *wa, real *wb) {
  const real costluar4 = K(0.66667), eps = K(0.000001);
  real z, transition, w2, w3, skew, skew2, radw2, fluarw, fluarw2, rluarw, xluarw, bluarw;
  (void)c;
  z = zp / h;
  transition = K(1.);
  if (-h / ol < K(15)) transition = r_sin(((-h / ol + K(10.)) / K(10.)) * PI_PAR) / K(2.) + K(0.5);
  w2 = sigmaw * sigmaw;
  w3 = ((K(1.2) * z * r_pow(K(1.) - z, K(1.5)) + eps) * (wst * wst * wst)) * transition;
  skew = w3 / r_pow(w2, K(1.5));
  skew2 = skew * skew;
  radw2 = r_sqrt(w2);
  fluarw = costluar4 * r_pow(skew, K(0.333333333333333));
  fluarw2 = fluarw * fluarw;
  rluarw = r_pow(K(1.) + fluarw2, K(3.)) * skew2 / (r_pow(K(3.) + fluarw2, K(2.)) * fluarw2);
  xluarw = r_pow(rluarw, K(0.5));
  *aluarw = K(0.5) * (K(1.) - xluarw / r_pow(K(4.) + rluarw, K(0.5)));
  bluarw = K(1.) - *aluarw;
  *sigmawa = radw2 * r_pow(bluarw / (*aluarw * (K(1.) + fluarw2)), K(0.5));
  *sigmawb = radw2 * r_pow(*aluarw / (bluarw * (K(1.) + fluarw2)), K(0.5));
  *wa = fluarw * *sigmawa;
  *wb = fluarw * *sigmawb;
}

/* re_initialize_particle.f90:44-90 */
static void orc_re_initialize_particle(orc_ctx *c, real zp, real ust, real wst, real h, real sigmaw, real *wp, int *nrand, real ol) {
  real aluarw, sigmawa, sigmawb, wa, wb, dcas1, timedir;
  (void)ust;
  *nrand = *nrand + 1;
  dcas1 = c->rannumb[*nrand];
  timedir = (real)c->ldirect;
  orc_cbl_pdf(c, zp, wst, h, sigmaw, ol, &aluarw, &sigmawa, &sigmawb, &wa, &wb);
  if (r_sign(K(1.), *wp) * timedir > 0) {          /* updraft */
    for (;;) {
      *wp = dcas1 * sigmawa + wa;
      if (*wp < 0) { *nrand = *nrand + 1; dcas1 = c->rannumb[*nrand]; continue; }
      break;
    }
    *wp = *wp * timedir;
  } else if (r_sign(K(1.), *wp) * timedir < 0) {   /* downdraft */
    for (;;) {
      *wp = dcas1 * sigmawb - wb;
      if (*wp > 0) { *nrand = *nrand + 1; dcas1 = c->rannumb[*nrand]; continue; }
      break;
    }
    *wp = *wp * timedir;
  }
}

/* initialize_cbl_vel.f90:46-83 */
static void orc_initialize_cbl_vel(orc_ctx *c, int *idum, real zp, real ust, real wst, real h, real sigmaw, real *wp, real ol) {
  real aluarw, sigmawa, sigmawb, wa, wb, dcas, dcas1, timedir;
  (void)ust;
  timedir = (real)c->ldirect;
  orc_cbl_pdf(c, zp, wst, h, sigmaw, ol, &aluarw, &sigmawa, &sigmawb, &wa, &wb);
  dcas = orc_ran3(c, idum);
  if (dcas <= aluarw) {
    dcas1 = orc_gasdev(c, idum);
    *wp = timedir * (dcas1 * sigmawa + wa);
  } else {
    dcas1 = orc_gasdev(c, idum);
    *wp = timedir * (dcas1 * sigmawb - wb);
  }
}

/* windalign.f90:36-54 */
static void orc_windalign(real u, real v, real ffap, real ffcp, real *ux, real *vy) {
  const real eps = K(1.e-30);
  real ffinv, ux1, ux2, vy1, vy2, sinphi, cosphi;
  ffinv = K(1.) / r_max(r_sqrt(u * u + v * v), eps);
  sinphi = v * ffinv;
  vy1 = sinphi * ffap;
  cosphi = u * ffinv;
  ux1 = cosphi * ffap;
  ux2 = -sinphi * ffcp;
  vy2 = cosphi * ffcp;
  *ux = ux1 + ux2;
  *vy = vy1 + vy2;
}

/* ------------------------------------------------------------------------- */
/* map projection subset: cmapf_mod.f90                                        */
/* ------------------------------------------------------------------------- */
#define CM_REARTH K(6371.2)
#define CM_ALMST1 K(.9999999)
#define CM_PI K(3.14159265358979)
#define CM_RADPDG (CM_PI / K(180.))
#define CM_DGPRAD (K(180.) / CM_PI)

/* cmapf_mod.f90:494-524 */
static real orc_cspanf(real value, real begin, real end) {
  real first = r_min(begin, end), last = r_max(begin, end), val;
  val = r_mod(value - first, last - first);
  return val <= K(0.) ? val + last : val + first;
}

/* cmapf_mod.f90:190-238 */
static real orc_cgszll(const real *s, real xlat, real xlong) {
  double slat, ymerc, efact;
  (void)xlong;
  if (xlat > K(89.985)) {
    if (s[0] > K(0.9999)) return K(2.) * s[6];
    efact = (double)r_cos(CM_RADPDG * xlat);
    if (efact <= 0.) return K(0.);
    ymerc = -log(efact / (double)(K(1.) + r_sin(CM_RADPDG * xlat)));
  } else if (xlat < K(-89.985)) {
    if (s[0] < K(-0.9999)) return K(2.) * s[6];
    efact = (double)r_cos(CM_RADPDG * xlat);
    if (efact <= 0.) return K(0.);
    ymerc = log(efact / (double)(K(1.) - r_sin(CM_RADPDG * xlat)));
  } else {
    slat = (double)r_sin(CM_RADPDG * xlat);
    ymerc = log((1. + slat) / (1. - slat)) / 2.;
  }
  return (real)((double)(s[6] * r_cos(CM_RADPDG * xlat)) * exp((double)s[0] * ymerc));
}

/* cmapf_mod.f90:310-365 */
static void orc_cnllxy(const real *s, real xlat, real xlong, real *xi, real *eta) {
  real gdlong, sndgam, csdgam, rhog1;
  double gamma, dlong, dlat, slat, mercy, gmercy;
  gamma = (double)s[0];
  dlat = (double)xlat;
  dlong = (double)orc_cspanf(xlong - s[1], K(-180.), K(180.));
  dlong = dlong * (double)CM_RADPDG;
  gdlong = (real)(gamma * dlong);
  if (r_abs(gdlong) < K(.01)) {
    gdlong = gdlong * gdlong;
    sndgam = (real)(dlong * (double)(K(1.) - K(1.) / K(6.) * gdlong * (K(1.) - K(1.) / K(20.) * gdlong * (K(1.) - K(1.) / K(42.) * gdlong))));
    csdgam = (real)(dlong * dlong * (double)K(.5) * (double)(K(1.) - K(1.) / K(12.) * gdlong * (K(1.) - K(1.) / K(30.) * gdlong * (K(1.) - K(1.) / K(56.) * gdlong))));
  } else {
    sndgam = (real)((double)r_sin(gdlong) / gamma);
    csdgam = (real)((double)(K(1.) - r_cos(gdlong)) / gamma / gamma);
  }
  slat = sin((double)CM_RADPDG * dlat);
  if (slat >= (double)CM_ALMST1 || slat <= -(double)CM_ALMST1) {
    *eta = K(1.) / s[0];
    *xi = K(0.);
    return;
  }
  mercy = .5 * log((1. + slat) / (1. - slat));
  gmercy = gamma * mercy;
  if (fabs(gmercy) < (double)K(.001)) {
    rhog1 = (real)(mercy * (1. - .5 * gmercy * (1. - (double)(K(1.) / K(3.)) * gmercy * (1. - (double)(K(1.) / K(4.)) * gmercy))));
  } else {
    rhog1 = (real)((1. - exp(-gmercy)) / gamma);
  }
  *eta = (real)((double)rhog1 + (1. - gamma * (double)rhog1) * gamma * (double)csdgam);
  *xi = (real)((1. - gamma * (double)rhog1) * (double)sndgam);
}

/* cmapf_mod.f90:295-308 */
static void orc_cll2xy(const real *s, real xlat, real xlong, real *x, real *y) {
  real xi, eta;
  orc_cnllxy(s, xlat, xlong, &xi, &eta);
  *x = s[2] + CM_REARTH / s[6] * (xi * s[4] + eta * s[5]);
  *y = s[3] + CM_REARTH / s[6] * (eta * s[4] - xi * s[5]);
}

/* cmapf_mod.f90:367-425 */
static void orc_cnxyll(const real *s, double xi, double eta, real *xlat, real *xlong) {
  double gamma, temp, arg1, arg2, ymerc, along, gxi, cgeta;
  gamma = (double)s[0];
  arg2 = 2. * eta - gamma * (xi * xi + eta * eta);
  arg1 = gamma * arg2;
  if (fabs(arg1) < (double)K(.01)) {
    temp = (arg1 / (2. - arg1)) * (arg1 / (2. - arg1));
    ymerc = arg2 / (2. - arg1) * (1. + temp * ((double)(K(1.) / K(3.)) + temp * ((double)(K(1.) / K(5.)) + temp * ((double)(K(1.) / K(7.))))));
  } else {
    ymerc = -log(1. - arg1) / 2. / gamma;
  }
  temp = exp(-fabs(ymerc));
  {
    double a = atan2((1. - temp) * (1. + temp), 2. * temp);
    *xlat = (real)(ymerc < 0 ? -fabs(a) : fabs(a));
  }
  gxi = gamma * xi;
  cgeta = 1. - gamma * eta;
  if (fabs(gxi) < (double)K(.01) * cgeta) {
    temp = (gxi / cgeta) * (gxi / cgeta);
    along = xi / cgeta * (1. - temp * ((double)(K(1.) / K(3.)) - temp * ((double)(K(1.) / K(5.)) - temp * ((double)(K(1.) / K(7.))))));
  } else {
    along = atan2(gxi, cgeta) / gamma;
  }
  *xlong = (real)((double)s[1] + (double)CM_DGPRAD * along);
  *xlat = *xlat * CM_DGPRAD;
}

/* cmapf_mod.f90:526-543 */
static void orc_cxy2ll(const real *s, real x, real y, real *xlat, real *xlong) {
  double xi0, eta0, xi, eta;
  xi0 = (double)((x - s[2]) * s[6] / CM_REARTH);
  eta0 = (double)((y - s[3]) * s[6] / CM_REARTH);
  xi = xi0 * (double)s[4] - eta0 * (double)s[5];
  eta = eta0 * (double)s[4] + xi0 * (double)s[5];
  orc_cnxyll(s, xi, eta, xlat, xlong);
  *xlong = orc_cspanf(*xlong, K(-180.), K(180.));
}

/* ------------------------------------------------------------------------- */
/* gravitational settling: get_settling.f90:52-127, dynamic_viscosity.f90:7-17 */
/* ------------------------------------------------------------------------- */
static real orc_viscosity(real t) {
  const real cc = K(120.), t_0 = K(291.15), eta_0 = K(1.827e-5);
  return eta_0 * (t_0 + cc) / (t + cc) * r_pow(t / t_0, K(1.5));
}

static void orc_get_settling(orc_ctx *c, int itime, real xt, real yt, real zt, int nsp, real *settling) {
  const real ga = K(9.81);
  real dz1, dz2, dz, rho1[2], tt1[2], temperature, airdens, vis_dyn, vis_kin, settling_old, reynolds, c_d;
  int i, n, nix, njy, indz = 1;
  (void)itime;
  nix = (int)xt;
  njy = (int)yt;
  for (i = 2; i <= c->nz; i++)
    if (HGT(i) > zt) { indz = i - 1; break; }
  dz = K(1.) / (HGT(indz + 1) - HGT(indz));
  dz1 = (zt - HGT(indz)) * dz;
  dz2 = (HGT(indz + 1) - zt) * dz;
  for (n = 0; n < 2; n++) {            /* literal time slot 1 (get_settling.f90:83-84) */
    rho1[n] = F3(c->rho, nix, njy, indz + n, 1);
    tt1[n] = F3(c->tt, nix, njy, indz + n, 1);
  }
  temperature = dz2 * tt1[0] + dz1 * tt1[1];
  airdens = dz2 * rho1[0] + dz1 * rho1[1];
  vis_dyn = orc_viscosity(temperature);
  vis_kin = vis_dyn / airdens;
  reynolds = c->dquer[nsp - 1] / K(1.e6) * r_abs(c->vsetaver[nsp - 1]) / vis_kin;
  settling_old = c->vsetaver[nsp - 1];
  for (i = 1; i <= 20; i++) {
    if (reynolds < K(1.917)) c_d = K(24.) / reynolds;
    else if (reynolds < K(500.)) c_d = K(18.5) / r_pow(reynolds, K(0.6));
    else c_d = K(0.44);
    *settling = K(-1.) * r_sqrt(K(4) * ga * c->dquer[nsp - 1] / K(1.e6) * c->density[nsp - 1] * c->cunningham[nsp - 1] / (K(3.) * c_d * airdens));
    if (r_abs((*settling - settling_old) / *settling) < K(0.01)) break;
    reynolds = c->dquer[nsp - 1] / K(1.e6) * r_abs(*settling) / vis_kin;
    settling_old = *settling;
  }
}

/* the species pick + settling add repeated at advance.f90:518-531,686-699,893-906 */
#define XMASS_PT(kp, ks) c->xmass_pt[(size_t)((ks) - 1) * c->numpoint + ((kp) - 1)]   /* xmass(kp,ks), 1-based */
static void orc_add_settling(orc_ctx *c, int itime, int nrelpoint, double xt, double yt, real zt) {
  const real eps3 = sizeof(real) == 4 ? (real)1.17549435e-38f : (real)2.2250738585072014e-308;
  int nsp;
  if (c->mdomainfill == 0 && c->lsettling) {
    for (nsp = 1; nsp <= c->nspec; nsp++)
      if (XMASS_PT(nrelpoint, nsp) > eps3) break;
    if (nsp > c->nspec) nsp = c->nspec;
    if (c->density[nsp - 1] > K(0.)) {
      orc_get_settling(c, itime, (real)xt, (real)yt, zt, nsp, &c->settling);
      c->w = c->w + c->settling;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* initialize.f90:66-217                                                       */
/* ------------------------------------------------------------------------- */
static void orc_initialize(orc_ctx *c, int itime, int *ldt, real *up, real *vp, real *wp,
                           real *usigold, real *vsigold, real *wsigold, double xt, double yt, real zt, int16_t *icbt) {
  int nrand, m;
  real dz, dz1, dz2, hh;
  *icbt = 1;
  nrand = (int)(orc_ran3(c, &c->idummy_init) * (real)(ORC_MAXRAND - 1)) + 1;
  c->ix = (int)xt;
  c->jy = (int)yt;
  c->ixp = c->ix + 1;
  c->jyp = c->jy + 1;
  hh = F2(c->hmix, c->ix, c->jy, c->memind[0]);
  for (m = 0; m < 2; m++) {
    hh = r_max(hh, F2(c->hmix, c->ix, c->jy, c->memind[m]));
    hh = r_max(hh, F2(c->hmix, c->ixp, c->jy, c->memind[m]));
    hh = r_max(hh, F2(c->hmix, c->ix, c->jyp, c->memind[m]));
    hh = r_max(hh, F2(c->hmix, c->ixp, c->jyp, c->memind[m]));
  }
  c->h = hh;
  c->zeta = zt / c->h;
  /* D2: initialize.f90 never sets ngrid -- interpol_all/interpol_wind read the value the
     previous particle's advance() left in interpol_mod. */
  {
    int own = 0;
    if (c->nglobal && yt > (double)c->switchnorthg) own = -1;
    else if (c->sglobal && yt < (double)c->switchsouthg) own = -2;
    if (c->leak_flags && ((own < 0) != (c->ngrid < 0))) c->leak_flags[c->cur_particle] |= 2;
    if (c->parallel_semantics) c->ngrid = own;
  }
  if (c->zeta <= K(1.)) {
    orc_interpol_all(c, itime, (real)xt, (real)yt, zt);
    dz1 = zt - HGT(c->indz);
    dz2 = HGT(c->indzp) - zt;
    dz = K(1.) / (dz1 + dz2);
    c->u = (dz1 * c->uprof[c->indzp] + dz2 * c->uprof[c->indz]) * dz;
    c->v = (dz1 * c->vprof[c->indzp] + dz2 * c->vprof[c->indz]) * dz;
    c->w = (dz1 * c->wprof[c->indzp] + dz2 * c->wprof[c->indz]) * dz;
    if (c->turbswitch) orc_hanna(c, zt); else orc_hanna1(c, zt);
    if (nrand + 2 > ORC_MAXRAND) nrand = 1;
    *up = c->rannumb[nrand] * c->sigu;
    *vp = c->rannumb[nrand + 1] * c->sigv;
    *wp = c->rannumb[nrand + 2];
    if (!c->turbswitch) {
      *wp = *wp * c->sigw;
    } else if (c->cblflag == 1) {
      if (-c->h / c->ol > K(5)) orc_initialize_cbl_vel(c, &c->idummy_init, zt, c->ust, c->wst, c->h, c->sigw, wp, c->ol);
      else *wp = *wp * c->sigw;
    }
    if (c->turbswitch)
      *ldt = (int)(r_min(r_min(r_min(c->tlw, c->h / r_max(K(2.) * r_abs(*wp * c->sigw), K(1.e-5))), K(0.5) / r_abs(c->dsigwdz)), K(600.)) * c->ctl);
    else
      *ldt = (int)(r_min(r_min(c->tlw, c->h / r_max(K(2.) * r_abs(*wp), K(1.e-5))), K(600.)) * c->ctl);
    if (*ldt < c->mintime) *ldt = c->mintime;
    c->usig = (c->usigprof[c->indzp] + c->usigprof[c->indz]) / K(2.);
    c->vsig = (c->vsigprof[c->indzp] + c->vsigprof[c->indz]) / K(2.);
    c->wsig = (c->wsigprof[c->indzp] + c->wsigprof[c->indz]) / K(2.);
  } else {
    orc_interpol_wind(c, itime, (real)xt, (real)yt, zt, 1);
    *ldt = abs(c->lsynctime);
    if (nrand + 1 > ORC_MAXRAND) nrand = 1;
    *up = c->rannumb[nrand] * K(0.3);
    *vp = c->rannumb[nrand + 1] * K(0.3);
    nrand = nrand + 2;
    *wp = K(0.);
    c->sigw = K(0.);
  }
  if (nrand + 2 > ORC_MAXRAND) nrand = 1;
  *usigold = c->rannumb[nrand] * c->usig;
  *vsigold = c->rannumb[nrand + 1] * c->vsig;
  *wsigold = c->rannumb[nrand + 2] * c->wsig;
}

/* ------------------------------------------------------------------------- */
/* advance.f90:133-985                                                         */
/* ------------------------------------------------------------------------- */
/* grid choice: advance.f90:161-175 (and again :841-855) */
static int orc_pick_grid(orc_ctx *c, double xt, double yt, real eps) {
  int j;
  if (c->nglobal && yt > (double)c->switchnorthg) return -1;
  if (c->sglobal && yt < (double)c->switchsouthg) return -2;
  for (j = c->numbnests; j >= 1; j--)
    if (xt > (double)(c->xln[j - 1] + eps) && xt < (double)(c->xrn[j - 1] - eps) &&
        yt > (double)(c->yln[j - 1] + eps) && yt < (double)(c->yrn[j - 1] - eps)) return j;
  return 0;
}

/* boundary conditions: advance.f90:784-813 == :956-985; returns nstop */
static int orc_boundary(orc_ctx *c, double *xt, double *yt, real *zt, real eps) {
  if (c->xglobal) {
    if (*xt >= (double)(real)c->nxmin1) *xt = *xt - (double)(real)c->nxmin1;
    if (*xt < 0.) *xt = *xt + (double)(real)c->nxmin1;
    if (*xt <= (double)eps) *xt = (double)eps;
    if (fabs(*xt - (double)(real)c->nxmin1) <= (double)eps) *xt = (double)((real)c->nxmin1 - eps);
    if (*yt < 0.) {
      *xt = d_modulo(*xt * (double)c->dx + 180., 360.) / (double)c->dx;
      *yt = -*yt;
    } else if (*yt > (double)(real)c->nymin1) {
      *xt = d_modulo(*xt * (double)c->dx + 180., 360.) / (double)c->dx;
      *yt = (double)(K(2) * (real)c->nymin1) - *yt;
    }
  }
  if (*xt < 0. || *xt >= (double)(real)c->nxmin1 || *yt < 0. || *yt > (double)(real)c->nymin1) return 3;
  if (*zt >= HGT(c->nz)) *zt = HGT(c->nz) - K(100.) * eps;
  return 0;
}

/* horizontal move by (du,dv) [m] on grid ngrid: advance.f90:750-778 == :923-951 */
static void orc_move(orc_ctx *c, double *xt, double *yt, real du, real dv, real fac) {
  const real pi180 = PI_PAR / K(180.);
  if (c->ngrid >= 0) {
    real cosfact = (real)((double)c->dxconst / cos((*yt * (double)c->dy + (double)c->ylat0) * (double)pi180));
    *xt = *xt + (double)(du * cosfact * fac);
    *yt = *yt + (double)(dv * c->dyconst * fac);
  } else {
    const real *map = c->ngrid == -1 ? c->northpolemap : c->southpolemap;
    real xlon, ylat, xpol, ypol, gridsize;
    xlon = (real)((double)c->xlon0 + *xt * (double)c->dx);
    ylat = (real)((double)c->ylat0 + *yt * (double)c->dy);
    orc_cll2xy(map, ylat, xlon, &xpol, &ypol);
    gridsize = K(1000.) * orc_cgszll(map, ylat, xlon);
    du = du / gridsize;
    dv = dv / gridsize;
    xpol = xpol + du * fac;
    ypol = ypol + dv * fac;
    orc_cxy2ll(map, xpol, ypol, &ylat, &xlon);
    *xt = (double)((xlon - c->xlon0) / c->dx);
    *yt = (double)((ylat - c->ylat0) / c->dy);
  }
}

static int orc_advance(orc_ctx *c, int itime, int nrelpoint, int *ldt, real *up, real *vp, real *wp,
                       real *usigold, real *vsigold, real *wsigold, double *xt, double *yt, real *zt,
                       real *prob, int16_t *icbt) {
  const real eps = c->eps_nxmax / K(3.e5);   /* nxmax/3.e5: advance.f90:107 with the build's par_mod nxmax */
  const real eps2 = K(1.e-9);
  const real href = K(15.);
  int itimec, i, k, nrand, loop, ngr, nix, njy, ks, mind, flagrein;
  real xts, yts, dz, dz1, dz2, ru, rv, rw, dt, ux, vy, tropop, dxsave, dysave, dawsave, dcwsave;
  real r, rs, uold, vold, wold, vdepo[ORC_MAXSPEC], rhoa, rhograd, delz = 0, dtf, rhoaux, dtftlw, uxscale, wpscale, weight;
  real ptot_lhh, Q_lhh, phi_lhh, ath, bth, old_wp_buf, del_test;
  real xtn = 0, ytn = 0;
  (void)nrelpoint;

  /* advance.f90:133-153 */
  for (i = 1; i <= c->nmixz; i++) c->indzindicator[i] = 1;
  if (c->drydep)
    for (ks = 0; ks < c->nspec; ks++) { c->depoindicator[ks] = 1; prob[ks] = K(0.); }
  dxsave = K(0.); dysave = K(0.); dawsave = K(0.); dcwsave = K(0.);
  itimec = itime;
  nrand = (int)(orc_ran3(c, &c->idummy_adv) * (real)(ORC_MAXRAND - 1)) + 1;

  /* :161-175 */
  c->ngrid = orc_pick_grid(c, *xt, *yt, eps);

  /* :191-231 */
  if (c->ngrid > 0) {
    xtn = (real)((*xt - (double)c->xln[c->ngrid - 1]) * (double)c->xresoln[c->ngrid - 1]);
    ytn = (real)((*yt - (double)c->yln[c->ngrid - 1]) * (double)c->yresoln[c->ngrid - 1]);
    c->ix = (int)xtn; c->jy = (int)ytn;
    nix = (int)lround((double)xtn); njy = (int)lround((double)ytn);
  } else {
    c->ix = (int)*xt; c->jy = (int)*yt;
    nix = (int)lround(*xt); njy = (int)lround(*yt);
  }
  c->ixp = c->ix + 1;
  c->jyp = c->jy + 1;
  c->ddx = (real)(*xt - (double)(real)c->ix);
  c->ddy = (real)(*yt - (double)(real)c->jy);
  c->rddx = K(1.) - c->ddx;
  c->rddy = K(1.) - c->ddy;
  c->p1 = c->rddx * c->rddy;
  c->p2 = c->ddx * c->rddy;
  c->p3 = c->rddx * c->ddy;
  c->p4 = c->ddx * c->ddy;
  c->dt1 = (real)(itime - c->memtime[0]);
  c->dt2 = (real)(c->memtime[1] - itime);
  c->dtt = K(1.) / (c->dt1 + c->dt2);
  if (c->jyp >= (c->ngrid > 0 ? c->nyn[c->ngrid - 1] : c->ny)) c->jyp = c->jyp - 1;   /* :228-231 (compact layouts: the row count of the grid in use) */

  /* :236-267 */
  c->h = K(0.);
  if (c->ngrid <= 0) {
    real h1[2] = {K(0.), K(0.)};
    for (k = 0; k < 2; k++) {
      int jj, ii;
      mind = c->memind[k];
      if (c->interpolhmix) {   /* :240-244 */
        h1[k] = c->p1 * F2(c->hmix, c->ix, c->jy, mind) + c->p2 * F2(c->hmix, c->ixp, c->jy, mind)
              + c->p3 * F2(c->hmix, c->ix, c->jyp, mind) + c->p4 * F2(c->hmix, c->ixp, c->jyp, mind);
      } else {
        for (jj = c->jy; jj <= c->jyp; jj++)
          for (ii = c->ix; ii <= c->ixp; ii++)
            if (F2(c->hmix, ii, jj, mind) > c->h) c->h = F2(c->hmix, ii, jj, mind);
      }
    }
    if (c->interpolhmix) c->h = (h1[0] * c->dt2 + h1[1] * c->dt1) * c->dtt;   /* :266 (on the mother grid; h1 is unset inside a nest) */
    tropop = F2(c->tropopause, nix, njy, 1);
  } else {   /* :255-263 */
    const int l = c->ngrid;
    for (k = 0; k < 2; k++) {
      int jj, ii;
      mind = c->memind[k];
      for (jj = c->jy; jj <= c->jyp; jj++)
        for (ii = c->ix; ii <= c->ixp; ii++)
          if (N2(c->hmixn[l - 1], ii, jj, mind, l) > c->h) c->h = N2(c->hmixn[l - 1], ii, jj, mind, l);
    }
    tropop = N2(c->tropopausen[l - 1], nix, njy, 1, l);
  }
  c->zeta = *zt / c->h;

  /* :276 PBL branch */
  if (c->zeta <= K(1.)) {
    loop = 0;
  L100:
    loop = loop + 1;
    if (c->method == 1) {
      int rem = abs(c->lsynctime - itimec + itime);
      *ldt = *ldt < rem ? *ldt : rem;
      itimec = itimec + *ldt * c->ldirect;
    } else {
      *ldt = abs(c->lsynctime);
      itimec = itime + c->lsynctime;
    }
    dt = (real)*ldt;
    c->zeta = *zt / c->h;

    if (loop == 1) {
      if (c->ngrid <= 0) {
        xts = (real)*xt;
        yts = (real)*yt;
        orc_interpol_all(c, itime, xts, yts, *zt);
      } else {
        orc_interpol_all_nests(c, itime, xtn, ytn, *zt);
      }
    } else {
      for (i = 2; i <= c->nz; i++)
        if (HGT(i) > *zt) { c->indz = i - 1; c->indzp = i; break; }
      for (i = c->indz; i <= c->indzp; i++)
        if (c->indzindicator[i]) { if (c->ngrid <= 0) orc_profile_level(c, i); else orc_profile_level_nests(c, i); }
    }

    /* :342-350 */
    dz = K(1.) / (HGT(c->indzp) - HGT(c->indz));
    dz1 = (*zt - HGT(c->indz)) * dz;
    dz2 = (HGT(c->indzp) - *zt) * dz;
    c->u = dz1 * c->uprof[c->indzp] + dz2 * c->uprof[c->indz];
    c->v = dz1 * c->vprof[c->indzp] + dz2 * c->vprof[c->indz];
    c->w = dz1 * c->wprof[c->indzp] + dz2 * c->wprof[c->indz];
    rhoa = dz1 * c->rhoprof[c->indzp] + dz2 * c->rhoprof[c->indz];
    rhograd = dz1 * c->rhogradprof[c->indzp] + dz2 * c->rhogradprof[c->indz];

    /* :357-361 */
    if (c->turbswitch) orc_hanna(c, *zt); else orc_hanna1(c, *zt);

    /* :371-384 horizontal Langevin */
    if (nrand + 1 > ORC_MAXRAND) nrand = 1;
    if (dt / c->tlu < K(.5)) {
      *up = (K(1.) - dt / c->tlu) * *up + c->rannumb[nrand] * c->sigu * r_sqrt(K(2.) * dt / c->tlu);
    } else {
      ru = r_exp(-dt / c->tlu);
      *up = ru * *up + c->rannumb[nrand] * c->sigu * r_sqrt(K(1.) - ru * ru);
    }
    if (dt / c->tlv < K(.5)) {
      *vp = (K(1.) - dt / c->tlv) * *vp + c->rannumb[nrand + 1] * c->sigv * r_sqrt(K(2.) * dt / c->tlv);
    } else {
      rv = r_exp(-dt / c->tlv);
      *vp = rv * *vp + c->rannumb[nrand + 1] * c->sigv * r_sqrt(K(1.) - rv * rv);
    }
    nrand = nrand + 2;

    /* :387-391 */
    if (nrand + c->ifine > ORC_MAXRAND) nrand = 1;
    rhoaux = rhograd / rhoa;
    dtf = dt * c->fine;
    dtftlw = dtf / c->tlw;

    /* :396-498 vertical Langevin, ifine sub-steps */
    for (i = 1; i <= c->ifine; i++) {
      if (c->turbswitch) {
        if (dtftlw < K(.5)) {
          if (c->cblflag == 1) {
            if (-c->h / c->ol > K(5)) {
              flagrein = 0;
              nrand = nrand + 1;
              old_wp_buf = *wp;
              orc_cbl(c, *wp, *zt, c->ust, c->wst, c->h, rhoa, rhograd, c->sigw, c->dsigwdz, c->tlw, &ptot_lhh, &Q_lhh, &phi_lhh, &ath, &bth, c->ol, &flagrein);
              *wp = (*wp + ath * dtf + bth * c->rannumb[nrand] * r_sqrt(dtf)) * (real)*icbt;
              delz = *wp * dtf;
              if (flagrein == 1) {
                orc_re_initialize_particle(c, *zt, c->ust, c->wst, c->h, c->sigw, &old_wp_buf, &nrand, c->ol);
                *wp = old_wp_buf;
                delz = *wp * dtf;
                c->nan_count++;
              }
            } else {
              nrand = nrand + 1;
              old_wp_buf = *wp;
              ath = -*wp / c->tlw + c->sigw * c->dsigwdz + *wp * *wp / c->sigw * c->dsigwdz + c->sigw * c->sigw / rhoa * rhograd;
              bth = c->sigw * c->rannumb[nrand] * r_sqrt(K(2.) * dtftlw);
              *wp = (*wp + ath * dtf + bth) * (real)*icbt;
              delz = *wp * dtf;
              del_test = (K(1.) - *wp) / *wp;
              if (isnan((double)*wp) || isnan((double)del_test)) {
                nrand = nrand + 1;
                *wp = c->sigw * c->rannumb[nrand];
                delz = *wp * dtf;
                c->nan_count2++;
              }
            }
          } else {
            *wp = ((K(1.) - dtftlw) * *wp + c->rannumb[nrand + i] * r_sqrt(K(2.) * dtftlw) + dtf * (c->dsigwdz + rhoaux * c->sigw)) * (real)*icbt;
            delz = *wp * c->sigw * dtf;
          }
        } else {
          rw = r_exp(-dtftlw);
          *wp = (rw * *wp + c->rannumb[nrand + i] * r_sqrt(K(1.) - rw * rw) + c->tlw * (K(1.) - rw) * (c->dsigwdz + rhoaux * c->sigw)) * (real)*icbt;
          delz = *wp * c->sigw * dtf;
        }
      } else {
        rw = r_exp(-dtftlw);
        *wp = (rw * *wp + c->rannumb[nrand + i] * r_sqrt(K(1.) - rw * rw) * c->sigw + c->tlw * (K(1.) - rw) * (c->dsigw2dz + rhoaux * (c->sigw * c->sigw))) * (real)*icbt;
        delz = *wp * dtf;
      }
      if (c->turboff) {   /* :464-470 */
        *up = K(0.);
        *vp = K(0.);
        *wp = K(0.);
        delz = K(0.);
      }

      /* :476-491 */
      if (r_abs(delz) > c->h) delz = r_mod(delz, c->h);
      if (delz < -*zt) {
        *icbt = -1;
        *zt = -*zt - delz;
      } else if (delz > (c->h - *zt)) {
        *icbt = -1;
        *zt = -*zt - delz + K(2.) * c->h;
      } else {
        *icbt = 1;
        *zt = *zt + delz;
      }
      if (i != c->ifine) {
        c->zeta = *zt / c->h;
        orc_hanna_short(c, *zt);
      }
    }
    if (c->cblflag != 1) nrand = nrand + i;   /* i == ifine+1 here, as in the reference (:499) */

    /* :504-510 */
    if (c->turbswitch)
      *ldt = (int)(r_min(r_min(c->tlw, c->h / r_max(K(2.) * r_abs(*wp * c->sigw), K(1.e-5))), K(0.5) / r_abs(c->dsigwdz)) * c->ctl);
    else
      *ldt = (int)(r_min(c->tlw, c->h / r_max(K(2.) * r_abs(*wp), K(1.e-5))) * c->ctl);
    if (*ldt < c->mintime) *ldt = c->mintime;

    /* :518-531 */
    orc_add_settling(c, itime, nrelpoint, *xt, *yt, *zt);

    /* :539-547 */
    dxsave = dxsave + c->u * dt;
    dysave = dysave + c->v * dt;
    dawsave = dawsave + *up * dt;
    dcwsave = dcwsave + *vp * dt;
    *zt = *zt + c->w * dt * (real)c->ldirect;
    if (*zt >= HGT(c->nz)) *zt = HGT(c->nz) - K(100.) * eps;

    /* :549-552 */
    if (*zt > c->h) {
      if (itimec == itime + c->lsynctime) {
        /* D1: the reference jumps to 99 with usig/vsig/wsig left over from an earlier
           particle (advance.f90:550 skips :604-606). */
        if (c->leak_flags) c->leak_flags[c->cur_particle] |= 1;
        if (c->parallel_semantics) {
          c->usig = K(0.5) * (c->usigprof[c->indzp] + c->usigprof[c->indz]);
          c->vsig = K(0.5) * (c->vsigprof[c->indzp] + c->vsigprof[c->indz]);
          c->wsig = K(0.5) * (c->wsigprof[c->indzp] + c->wsigprof[c->indz]);
        }
        goto L99;
      }
      goto L700;
    }

    /* :582-599 dry deposition probability */
    if (c->drydep && *zt < K(2.) * href) {
      for (ks = 1; ks <= c->nspec; ks++) {
        if (c->drydepspec[ks - 1]) {
          if (c->depoindicator[ks - 1]) { if (c->ngrid <= 0) orc_interpol_vdep(c, ks, &vdepo[ks - 1]); else orc_interpol_vdep_nests(c, ks, &vdepo[ks - 1]); }
          prob[ks - 1] = K(1.) + (prob[ks - 1] - K(1.)) * r_exp(-vdepo[ks - 1] * r_abs(dt) / (K(2.) * href));
        }
      }
    }

    if (*zt < K(0.)) *zt = r_min(c->h - eps2, K(-1.) * *zt);   /* :601 */

    if (itimec == itime + c->lsynctime) {   /* :603-608 */
      c->usig = K(0.5) * (c->usigprof[c->indzp] + c->usigprof[c->indz]);
      c->vsig = K(0.5) * (c->vsigprof[c->indzp] + c->vsigprof[c->indz]);
      c->wsig = K(0.5) * (c->wsigprof[c->indzp] + c->wsigprof[c->indz]);
      goto L99;
    }
    goto L100;
  }

L700:
  /* :629-636 */
  if (c->ngrid <= 0) {
    xts = (real)*xt;
    yts = (real)*yt;
    orc_interpol_wind(c, itime, xts, yts, *zt, 1);
  } else {
    orc_interpol_wind_nests(c, itime, xtn, ytn, *zt, 1);
  }

  /* :647-673 */
  *ldt = abs(c->lsynctime - itimec + itime);
  dt = (real)*ldt;
  if (*zt < tropop) {
    uxscale = r_sqrt(K(2.) * c->d_trop / dt);
    if (nrand + 1 > ORC_MAXRAND) nrand = 1;
    ux = c->rannumb[nrand] * uxscale;
    vy = c->rannumb[nrand + 1] * uxscale;
    nrand = nrand + 2;
    *wp = K(0.);
  } else if (*zt < tropop + K(1000.)) {
    weight = (*zt - tropop) / K(1000.);
    uxscale = r_sqrt(K(2.) * c->d_trop / dt * (K(1.) - weight));
    if (nrand + 2 > ORC_MAXRAND) nrand = 1;
    ux = c->rannumb[nrand] * uxscale;
    vy = c->rannumb[nrand + 1] * uxscale;
    wpscale = r_sqrt(K(2.) * c->d_strat / dt * weight);
    *wp = c->rannumb[nrand + 2] * wpscale + c->d_strat / K(1000.);
    nrand = nrand + 3;
  } else {
    if (nrand > ORC_MAXRAND) nrand = 1;
    ux = K(0.);
    vy = K(0.);
    wpscale = r_sqrt(K(2.) * c->d_strat / dt);
    *wp = c->rannumb[nrand] * wpscale;
    nrand = nrand + 1;
  }

  if (c->turboff) {   /* :675-679 */
    ux = K(0.);
    vy = K(0.);
    *wp = K(0.);
  }

  /* :686-699 */
  orc_add_settling(c, itime, nrelpoint, *xt, *yt, *zt);

  /* :705-708 */
  dxsave = dxsave + (c->u + ux) * dt;
  dysave = dysave + (c->v + vy) * dt;
  *zt = *zt + (c->w + *wp) * dt * (real)c->ldirect;
  if (*zt < K(0.)) *zt = r_min(c->h - eps2, K(-1.) * *zt);

L99:
  /* :728-739 mesoscale fluctuations */
  r = r_exp(K(-2.) * (real)abs(c->lsynctime) / (real)c->lwindinterv);
  rs = r_sqrt(K(1.) - r * r);
  if (nrand + 2 > ORC_MAXRAND) nrand = 1;
  *usigold = r * *usigold + rs * c->rannumb[nrand] * c->usig * c->turbmesoscale;
  *vsigold = r * *vsigold + rs * c->rannumb[nrand + 1] * c->vsig * c->turbmesoscale;
  *wsigold = r * *wsigold + rs * c->rannumb[nrand + 2] * c->wsig * c->turbmesoscale;
  dxsave = dxsave + *usigold * (real)c->lsynctime;
  dysave = dysave + *vsigold * (real)c->lsynctime;
  *zt = *zt + *wsigold * (real)c->lsynctime;
  if (*zt < K(0.)) *zt = K(-1.) * *zt;

  /* :747-778 */
  orc_windalign(dxsave, dysave, dawsave, dcwsave, &ux, &vy);
  dxsave = dxsave + ux;
  dysave = dysave + vy;
  orc_move(c, xt, yt, dxsave, dysave, (real)c->ldirect);

  /* :784-813 */
  if (orc_boundary(c, xt, yt, zt, eps)) return 3;

  /* :829-857 Petterssen gates */
  if (*ldt != abs(c->lsynctime)) return 0;
  if (abs(itime + *ldt * c->ldirect) > abs(c->memtime[1])) return 0;
  ngr = orc_pick_grid(c, *xt, *yt, eps);
  if (ngr != c->ngrid) return 0;

  /* :862-872 */
  if (c->ngrid > 0) {
    xtn = (real)((*xt - (double)c->xln[c->ngrid - 1]) * (double)c->xresoln[c->ngrid - 1]);
    ytn = (real)((*yt - (double)c->yln[c->ngrid - 1]) * (double)c->yresoln[c->ngrid - 1]);
    c->ix = (int)xtn; c->jy = (int)ytn;
  } else {
    c->ix = (int)*xt; c->jy = (int)*yt;
  }
  c->ixp = c->ix + 1;
  c->jyp = c->jy + 1;

  /* :878-891 */
  uold = c->u; vold = c->v; wold = c->w;
  if (c->ngrid <= 0) {
    xts = (real)*xt;
    yts = (real)*yt;
    orc_interpol_wind(c, itime + *ldt * c->ldirect, xts, yts, *zt, 0);
  } else {
    orc_interpol_wind_nests(c, itime + *ldt * c->ldirect, xtn, ytn, *zt, 0);
  }

  /* :893-906 */
  orc_add_settling(c, itime + *ldt, nrelpoint, *xt, *yt, *zt);

  /* :913-951 */
  c->u = (c->u - uold) / K(2.);
  c->v = (c->v - vold) / K(2.);
  c->w = (c->w - wold) / K(2.);
  *zt = *zt + c->w * (real)(*ldt * c->ldirect);
  if (*zt < K(0.)) *zt = r_min(c->h - eps2, K(-1.) * *zt);
  orc_move(c, xt, yt, c->u, c->v, (real)(*ldt * c->ldirect));

  /* :956-985 */
  if (orc_boundary(c, xt, yt, zt, eps)) return 3;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* grid sampling: conccalc.f90:50-295 (mother output grid), drydepokernel.f90   */
/* ------------------------------------------------------------------------- */
#define ORC_GIDXN(ix, jy, kz, ks, kp, nc, na) ((size_t)(ix) + (size_t)c->numxgridn * ((size_t)(jy) + (size_t)c->numygridn * ((size_t)((kz) - 1) + (size_t)c->numzgrid * ((size_t)((ks) - 1) + (size_t)c->maxspec_out * ((size_t)((kp) - 1) + (size_t)c->maxpointspec_act * ((size_t)((nc) - 1) + (size_t)c->nclassunc * (size_t)((na) - 1)))))))
#define ORC_DIDXN(ix, jy, ks, kp, nc, na) ((size_t)(ix) + (size_t)c->numxgridn * ((size_t)(jy) + (size_t)c->numygridn * ((size_t)((ks) - 1) + (size_t)c->maxspec_out * ((size_t)((kp) - 1) + (size_t)c->maxpointspec_act * ((size_t)((nc) - 1) + (size_t)c->nclassunc * (size_t)((na) - 1))))))
#define GIDX(ix, jy, kz, ks, kp, nc, na) ((size_t)(ix) + (size_t)c->numxgrid * ((size_t)(jy) + (size_t)c->numygrid * ((size_t)((kz) - 1) + (size_t)c->numzgrid * ((size_t)((ks) - 1) + (size_t)c->maxspec_out * ((size_t)((kp) - 1) + (size_t)c->maxpointspec_act * ((size_t)((nc) - 1) + (size_t)c->nclassunc * (size_t)((na) - 1)))))))
#define DIDX(ix, jy, ks, kp, nc, na) ((size_t)(ix) + (size_t)c->numxgrid * ((size_t)(jy) + (size_t)c->numygrid * ((size_t)((ks) - 1) + (size_t)c->maxspec_out * ((size_t)((kp) - 1) + (size_t)c->maxpointspec_act * ((size_t)((nc) - 1) + (size_t)c->nclassunc * (size_t)((na) - 1))))))

static int orc_ageclass(orc_ctx *c, int itage) {   /* conccalc.f90:54-58, timemanager.f90:545-548 */
  int nage;
  for (nage = 1; nage <= c->nageclass; nage++)
    if (itage < c->lage[nage - 1]) break;
  return nage;   /* nageclass+1 when older than every class, exactly as the Fortran loop leaves it */
}

/* conccalc.f90:50-295 */
void orc_conccalc(orc_ctx *c, int itime, double weight_d, int npart, const double *xtra1, const double *ytra1,
                  const real *ztra1, const int *itra1, const int *itramem, const int *npoint, const int *nclass,
                  const real *xmass1, const real *xscav_frac1) {
  const real weight = (real)weight_d;
  /* DRYBKDEP / WETBKDEP: every contribution carries the factor max(xscav_frac1(i,ks), 0.), conccalc.f90:177-181,226-230 ... */
  const int bk = (c->drybkdep || c->wetbkdep) && xscav_frac1;
#define SCAV(ks_) (bk ? r_max(xscav_frac1[(size_t)((ks_) - 1) * npart + i], K(0.)) : K(1.))
  int i, ks;
  for (i = 0; i < npart; i++) {
    int itage, nage, ix, jy, ixp, jyp, kz, nrelpointer, il, ind, indz = 1, indzp = 2;
    real rhoi = K(1.), xl, yl, ddx, ddy, wx, wy, w;
    if (itra1[i] != itime) continue;
    itage = abs(itra1[i] - itramem[i]);
    nage = orc_ageclass(c, itage);
    if (nage > c->nageclass) continue;   /* guard, as in the kernels above: no plane for this age */
    if (c->ind_samp == -1) {   /* :80-122 */
      real rddx, rddy, p1, p2, p3, p4, dz1, dz2, dz, rhoprof[2];
      ix = (int)xtra1[i]; jy = (int)ytra1[i];
      ixp = ix + 1; jyp = jy + 1;
      ddx = (real)(xtra1[i] - (double)(real)ix);
      ddy = (real)(ytra1[i] - (double)(real)jy);
      rddx = K(1.) - ddx; rddy = K(1.) - ddy;
      p1 = rddx * rddy; p2 = ddx * rddy; p3 = rddx * ddy; p4 = ddx * ddy;
      if (jyp >= c->ny) jyp = jyp - 1;
      for (il = 2; il <= c->nz; il++)
        if (HGT(il) > ztra1[i]) { indz = il - 1; indzp = il; break; }
      dz1 = ztra1[i] - HGT(indz);
      dz2 = HGT(indzp) - ztra1[i];
      dz = K(1.) / (dz1 + dz2);
      for (ind = indz; ind <= indzp; ind++)   /* the literal slot 2 of :118-120 is kept */
        rhoprof[ind - indz] = p1 * F3(c->rho, ix, jy, ind, c->memind[1]) + p2 * F3(c->rho, ixp, jy, ind, 2) +
                              p3 * F3(c->rho, ix, jyp, ind, 2) + p4 * F3(c->rho, ixp, jyp, ind, 2);
      rhoi = (dz1 * rhoprof[1] + dz2 * rhoprof[0]) * dz;
    }
    nrelpointer = (c->ioutputforeachrelease == 0 || c->mdomainfill == 1) ? 1 : npoint[i];
    for (kz = 1; kz <= c->numzgrid; kz++)
      if (c->outheight[kz - 1] > ztra1[i]) break;
    if (kz > c->numzgrid) continue;
    xl = (real)((xtra1[i] * (double)c->dx + (double)c->xoutshift) / (double)c->dxout);
    yl = (real)((ytra1[i] * (double)c->dy + (double)c->youtshift) / (double)c->dyout);
    ix = (int)xl; if (xl < K(0.)) ix = ix - 1;
    jy = (int)yl; if (yl < K(0.)) jy = jy - 1;
    if (!c->lusekerneloutput || itage < 10800 || xl < K(0.5) || yl < K(0.5) ||
        xl > (real)(c->numxgrid - 1) - K(0.5) || yl > (real)(c->numygrid - 1) - K(0.5)) {
      if (ix >= 0 && jy >= 0 && ix <= c->numxgrid - 1 && jy <= c->numygrid - 1)
        for (ks = 1; ks <= c->nspec; ks++)
          c->gridunc[GIDX(ix, jy, kz, ks, nrelpointer, nclass[i], nage)] += xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * SCAV(ks);
    } else {
      ddx = xl - (real)ix;
      ddy = yl - (real)jy;
      if (ddx > K(0.5)) { ixp = ix + 1; wx = K(1.5) - ddx; } else { ixp = ix - 1; wx = K(0.5) + ddx; }
      if (ddy > K(0.5)) { jyp = jy + 1; wy = K(1.5) - ddy; } else { jyp = jy - 1; wy = K(0.5) + ddy; }
      if (ix >= 0 && ix <= c->numxgrid - 1) {
        if (jy >= 0 && jy <= c->numygrid - 1) {
          w = wx * wy;
          for (ks = 1; ks <= c->nspec; ks++) c->gridunc[GIDX(ix, jy, kz, ks, nrelpointer, nclass[i], nage)] += (bk ? xmass1[(size_t)(ks - 1) * npart + i] / rhoi * w * weight * SCAV(ks) : xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * w);
        }
        if (jyp >= 0 && jyp <= c->numygrid - 1) {
          w = wx * (K(1.) - wy);
          for (ks = 1; ks <= c->nspec; ks++) c->gridunc[GIDX(ix, jyp, kz, ks, nrelpointer, nclass[i], nage)] += xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * w * SCAV(ks);
        }
      }
      if (ixp >= 0 && ixp <= c->numxgrid - 1) {
        if (jyp >= 0 && jyp <= c->numygrid - 1) {
          w = (K(1.) - wx) * (K(1.) - wy);
          for (ks = 1; ks <= c->nspec; ks++) c->gridunc[GIDX(ixp, jyp, kz, ks, nrelpointer, nclass[i], nage)] += (bk ? xmass1[(size_t)(ks - 1) * npart + i] / rhoi * w * weight * SCAV(ks) : xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * w);
        }
        if (jy >= 0 && jy <= c->numygrid - 1) {
          w = (K(1.) - wx) * wy;
          for (ks = 1; ks <= c->nspec; ks++) c->gridunc[GIDX(ixp, jy, kz, ks, nrelpointer, nclass[i], nage)] += xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * w * SCAV(ks);
        }
      }
    }
    if (c->nested_output == 1) {   /* conccalc.f90:301-441, the same attribution on the nested output grid */
      xl = (real)((xtra1[i] * (double)c->dx + (double)c->xoutshiftn) / (double)c->dxoutn);
      yl = (real)((ytra1[i] * (double)c->dy + (double)c->youtshiftn) / (double)c->dyoutn);
      ix = (int)xl; if (xl < K(0.)) ix = ix - 1;
      jy = (int)yl; if (yl < K(0.)) jy = jy - 1;
      if (itage < 10800 || xl < K(0.5) || yl < K(0.5) || xl > (real)(c->numxgridn - 1) - K(0.5) ||
          yl > (real)(c->numygridn - 1) - K(0.5) || !c->lusekerneloutput) {
        if (ix >= 0 && jy >= 0 && ix <= c->numxgridn - 1 && jy <= c->numygridn - 1)
          for (ks = 1; ks <= c->nspec; ks++)
            c->griduncn[ORC_GIDXN(ix, jy, kz, ks, nrelpointer, nclass[i], nage)] += xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * SCAV(ks);
      } else {
        ddx = xl - (real)ix;
        ddy = yl - (real)jy;
        if (ddx > K(0.5)) { ixp = ix + 1; wx = K(1.5) - ddx; } else { ixp = ix - 1; wx = K(0.5) + ddx; }
        if (ddy > K(0.5)) { jyp = jy + 1; wy = K(1.5) - ddy; } else { jyp = jy - 1; wy = K(0.5) + ddy; }
        if (ix >= 0 && ix <= c->numxgridn - 1) {
          if (jy >= 0 && jy <= c->numygridn - 1) {
            w = wx * wy;
            for (ks = 1; ks <= c->nspec; ks++) c->griduncn[ORC_GIDXN(ix, jy, kz, ks, nrelpointer, nclass[i], nage)] += xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * w * SCAV(ks);
          }
          if (jyp >= 0 && jyp <= c->numygridn - 1) {
            w = wx * (K(1.) - wy);
            for (ks = 1; ks <= c->nspec; ks++) c->griduncn[ORC_GIDXN(ix, jyp, kz, ks, nrelpointer, nclass[i], nage)] += xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * w * SCAV(ks);
          }
        }
        if (ixp >= 0 && ixp <= c->numxgridn - 1) {
          if (jyp >= 0 && jyp <= c->numygridn - 1) {
            w = (K(1.) - wx) * (K(1.) - wy);
            for (ks = 1; ks <= c->nspec; ks++) c->griduncn[ORC_GIDXN(ixp, jyp, kz, ks, nrelpointer, nclass[i], nage)] += xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * w * SCAV(ks);
          }
          if (jy >= 0 && jy <= c->numygridn - 1) {
            w = (K(1.) - wx) * wy;
            for (ks = 1; ks <= c->nspec; ks++) c->griduncn[ORC_GIDXN(ixp, jy, kz, ks, nrelpointer, nclass[i], nage)] += xmass1[(size_t)(ks - 1) * npart + i] / rhoi * weight * w * SCAV(ks);
          }
        }
      }
    }
  }
  /* 2. concentrations at receptor points, parabolic kernel: conccalc.f90:451-498 */
  {
    const real factor = K(.596831), hxmax = K(6.0), hymax = K(4.0), hzmax = K(150.);
    int n;
    for (n = 0; n < c->numreceptor; n++) {
      real cc[ORC_MAXSPEC];
      for (ks = 0; ks < c->nspec; ks++) cc[ks] = K(0.);
      for (i = 0; i < npart; i++) {
        int itage;
        real hz, zd, hx, xd, hy, yd, h, r2;
        if (itra1[i] != itime) continue;
        itage = abs(itra1[i] - itramem[i]);
        hz = r_min(K(50.) + K(0.3) * r_sqrt((real)itage), hzmax);
        zd = ztra1[i] / hz;
        if (zd > K(1.)) continue;
        hx = r_min((K(0.29) + K(2.222e-3) * r_sqrt((real)itage)) * c->dx + (real)itage * K(1.2e-5), hxmax);
        xd = (real)((xtra1[i] - (double)c->xreceptor[n]) / (double)hx);
        if (xd * xd > K(1.)) continue;
        hy = r_min((K(0.18) + K(1.389e-3) * r_sqrt((real)itage)) * c->dy + (real)itage * K(7.5e-6), hymax);
        yd = (real)((ytra1[i] - (double)c->yreceptor[n]) / (double)hy);
        if (yd * yd > K(1.)) continue;
        h = hx * hy * hz;
        r2 = xd * xd + yd * yd + zd * zd;
        if (r2 < K(1.)) {
          const real xkern = factor * (K(1.) - r2);
          for (ks = 0; ks < c->nspec; ks++) cc[ks] = cc[ks] + xmass1[(size_t)ks * npart + i] * xkern / h;
        }
      }
      for (ks = 0; ks < c->nspec; ks++)
        c->creceptor[n + ORC_MAXRECEPTOR * ks] = c->creceptor[n + ORC_MAXRECEPTOR * ks] + K(2.) * weight * cc[ks] / c->receptorarea[n];
    }
  }
}

/* drydepokernel.f90:41-116 */
static void orc_drydepokernel(orc_ctx *c, int nunc, const dep_real *deposit, real x, real y, int nage, int kp) {
  if (nage > c->nageclass) return;   /* guard: the reference would write past the last age plane (particle older than lage(nageclass)) */
  real xl, yl, ddx, ddy, wx, wy, w;
  int ix, jy, ixp, jyp, ks;
  xl = (x * c->dx + c->xoutshift) / c->dxout;
  yl = (y * c->dy + c->youtshift) / c->dyout;
  ix = (int)xl;
  jy = (int)yl;
  ddx = xl - (real)ix;
  ddy = yl - (real)jy;
  if (ddx > K(0.5)) { ixp = ix + 1; wx = K(1.5) - ddx; } else { ixp = ix - 1; wx = K(0.5) + ddx; }
  if (ddy > K(0.5)) { jyp = jy + 1; wy = K(1.5) - ddy; } else { jyp = jy - 1; wy = K(0.5) + ddy; }
  for (ks = 1; ks <= c->nspec; ks++) {
    if (!(fabsf(deposit[ks - 1]) > 0 && c->drydepspec[ks - 1])) continue;
    if (!c->lusekerneloutput) {
      if (ix >= 0 && jy >= 0 && ix <= c->numxgrid - 1 && jy <= c->numygrid - 1) c->drygridunc[DIDX(ix, jy, ks, kp, nunc, nage)] = (dep_real)(c->drygridunc[DIDX(ix, jy, ks, kp, nunc, nage)] + deposit[ks - 1]);
      continue;
    }
    if (ix >= 0 && jy >= 0 && ix <= c->numxgrid - 1 && jy <= c->numygrid - 1) { w = wx * wy; c->drygridunc[DIDX(ix, jy, ks, kp, nunc, nage)] = (dep_real)((real)c->drygridunc[DIDX(ix, jy, ks, kp, nunc, nage)] + (real)deposit[ks - 1] * w); }
    if (ixp >= 0 && jyp >= 0 && ixp <= c->numxgrid - 1 && jyp <= c->numygrid - 1) { w = (K(1.) - wx) * (K(1.) - wy); c->drygridunc[DIDX(ixp, jyp, ks, kp, nunc, nage)] = (dep_real)((real)c->drygridunc[DIDX(ixp, jyp, ks, kp, nunc, nage)] + (real)deposit[ks - 1] * w); }
    if (ixp >= 0 && jy >= 0 && ixp <= c->numxgrid - 1 && jy <= c->numygrid - 1) { w = (K(1.) - wx) * wy; c->drygridunc[DIDX(ixp, jy, ks, kp, nunc, nage)] = (dep_real)((real)c->drygridunc[DIDX(ixp, jy, ks, kp, nunc, nage)] + (real)deposit[ks - 1] * w); }
    if (ix >= 0 && jyp >= 0 && ix <= c->numxgrid - 1 && jyp <= c->numygrid - 1) { w = wx * (K(1.) - wy); c->drygridunc[DIDX(ix, jyp, ks, kp, nunc, nage)] = (dep_real)((real)c->drygridunc[DIDX(ix, jyp, ks, kp, nunc, nage)] + (real)deposit[ks - 1] * w); }
  }
}

/* drydepokernel_nest.f90:38-100: always the uniform kernel (no lusekerneloutput branch), int() truncation */
static void orc_drydepokernel_nest(orc_ctx *c, int nunc, const dep_real *deposit, real x, real y, int nage, int kp) {
  if (nage > c->nageclass) return;   /* guard: the reference would write past the last age plane (particle older than lage(nageclass)) */
  real xl, yl, ddx, ddy, wx, wy, w;
  int ix, jy, ixp, jyp, ks;
  xl = (x * c->dx + c->xoutshiftn) / c->dxoutn;
  yl = (y * c->dy + c->youtshiftn) / c->dyoutn;
  ix = (int)xl;
  jy = (int)yl;
  ddx = xl - (real)ix;
  ddy = yl - (real)jy;
  if (ddx > K(0.5)) { ixp = ix + 1; wx = K(1.5) - ddx; } else { ixp = ix - 1; wx = K(0.5) + ddx; }
  if (ddy > K(0.5)) { jyp = jy + 1; wy = K(1.5) - ddy; } else { jyp = jy - 1; wy = K(0.5) + ddy; }
#define DNADD(i, j, ww) c->drygriduncn[ORC_DIDXN(i, j, ks, kp, nunc, nage)] = (dep_real)((real)c->drygriduncn[ORC_DIDXN(i, j, ks, kp, nunc, nage)] + (real)deposit[ks - 1] * (ww))
  for (ks = 1; ks <= c->nspec; ks++) {
    if (!(c->drydepspec[ks - 1] && fabsf(deposit[ks - 1]) > 0)) continue;
    if (ix >= 0 && jy >= 0 && ix <= c->numxgridn - 1 && jy <= c->numygridn - 1) { w = wx * wy; DNADD(ix, jy, w); }
    if (ixp >= 0 && jyp >= 0 && ixp <= c->numxgridn - 1 && jyp <= c->numygridn - 1) { w = (K(1.) - wx) * (K(1.) - wy); DNADD(ixp, jyp, w); }
    if (ixp >= 0 && jy >= 0 && ixp <= c->numxgridn - 1 && jy <= c->numygridn - 1) { w = (K(1.) - wx) * wy; DNADD(ixp, jy, w); }
    if (ix >= 0 && jyp >= 0 && ix <= c->numxgridn - 1 && jyp <= c->numygridn - 1) { w = wx * (K(1.) - wy); DNADD(ix, jyp, w); }
  }
#undef DNADD
}

/* ------------------------------------------------------------------------- */
/* wet deposition: wetdepo.f90:58-151, get_wetscav.f90:78-314,                  */
/* interpol_rain.f90:68-130, wetdepokernel.f90:38-108 (mother grid)             */
/* ------------------------------------------------------------------------- */
static void orc_interpol_rain(orc_ctx *c, const real *yy1, const real *yy2, const real *yy3, int iwftouse,
                              real xt, real yt, real *yint1, real *yint2, real *yint3) {
  int ix, jy, ixp, jyp;
  real ddx, ddy, rddx, rddy, p1, p2, p3, p4;
  if (xt >= (real)(c->nx - 1)) xt = (real)(c->nx - 1) - K(0.00001);
  if (yt >= (real)(c->ny - 1)) yt = (real)(c->ny - 1) - K(0.00001);
  ix = (int)xt; jy = (int)yt; ixp = ix + 1; jyp = jy + 1;
  ddx = xt - (real)ix; ddy = yt - (real)jy;
  rddx = K(1.) - ddx; rddy = K(1.) - ddy;
  p1 = rddx * rddy; p2 = ddx * rddy; p3 = rddx * ddy; p4 = ddx * ddy;
  *yint1 = p1 * F2(yy1, ix, jy, iwftouse) + p2 * F2(yy1, ixp, jy, iwftouse) + p3 * F2(yy1, ix, jyp, iwftouse) + p4 * F2(yy1, ixp, jyp, iwftouse);
  *yint2 = p1 * F2(yy2, ix, jy, iwftouse) + p2 * F2(yy2, ixp, jy, iwftouse) + p3 * F2(yy2, ix, jyp, iwftouse) + p4 * F2(yy2, ixp, jyp, iwftouse);
  *yint3 = p1 * F2(yy3, ix, jy, iwftouse) + p2 * F2(yy3, ixp, jy, iwftouse) + p3 * F2(yy3, ix, jyp, iwftouse) + p4 * F2(yy3, ixp, jyp, iwftouse);
}

/* interpol_rain_nests.f90:68-142 */
static void orc_interpol_rain_nests(orc_ctx *c, const real *yy1, const real *yy2, const real *yy3, int ngrid, int iwftouse,
                                    real xt, real yt, real *yint1, real *yint2, real *yint3) {
  int ix, jy, ixp, jyp;
  real ddx, ddy, rddx, rddy, p1, p2, p3, p4;
  if (xt >= (real)(c->nxn[ngrid - 1] - 1)) xt = (real)(c->nxn[ngrid - 1] - 1) - K(0.00001);
  if (yt >= (real)(c->nyn[ngrid - 1] - 1)) yt = (real)(c->nyn[ngrid - 1] - 1) - K(0.00001);
  ix = (int)xt; jy = (int)yt; ixp = ix + 1; jyp = jy + 1;
  ddx = xt - (real)ix; ddy = yt - (real)jy;
  rddx = K(1.) - ddx; rddy = K(1.) - ddy;
  p1 = rddx * rddy; p2 = ddx * rddy; p3 = rddx * ddy; p4 = ddx * ddy;
  *yint1 = p1 * N2(yy1, ix, jy, iwftouse, ngrid) + p2 * N2(yy1, ixp, jy, iwftouse, ngrid) + p3 * N2(yy1, ix, jyp, iwftouse, ngrid) + p4 * N2(yy1, ixp, jyp, iwftouse, ngrid);
  *yint2 = p1 * N2(yy2, ix, jy, iwftouse, ngrid) + p2 * N2(yy2, ixp, jy, iwftouse, ngrid) + p3 * N2(yy2, ix, jyp, iwftouse, ngrid) + p4 * N2(yy2, ixp, jyp, iwftouse, ngrid);
  *yint3 = p1 * N2(yy3, ix, jy, iwftouse, ngrid) + p2 * N2(yy3, ixp, jy, iwftouse, ngrid) + p3 * N2(yy3, ix, jyp, iwftouse, ngrid) + p4 * N2(yy3, ixp, jyp, iwftouse, ngrid);
}

static real r_pow10(real x) { return r_pow(K(10.), x); }
static real r_log10(real x) { return sizeof(real) == 4 ? (real)log10f((float)x) : (real)log10((double)x); }

/* get_wetscav.f90:78-314 */
static real orc_get_wetscav(orc_ctx *c, int itime, int ltsample, double xtra1, double ytra1, real ztra1, int ks, real *grfraction) {
  static const real lfr[5] = {K(0.5), K(0.65), K(0.8), K(0.9), K(0.95)};
  static const real cfr[5] = {K(0.4), K(0.55), K(0.7), K(0.8), K(0.9)};
  static const real bclr[6] = {K(274.35758), K(332839.59273), K(226656.57259), K(58005.91340), K(6588.38582), K(0.244984)};
  static const real bcls[6] = {K(22.7), K(0.0), K(0.0), K(1321.0), K(381.0), K(0.0)};
  const real incloud_ratio = K(6.2), r_air = K(287.05);
  real wetscav = K(0.), lsp, convp, cc, prec1, act_temp, S_i, cl, cle, frac_act, liq_frac, ice_frac, dquer_m;
  int ix, jy, hz = 1, il, interp_time, n, i, j, clouds_v, ngrid = 0;
  real xtn = K(0.), ytn = K(0.);
  /* nesting level, :82-90: the plain nest bounds, without the eps margin advance.f90:167-173 applies */
  for (j = c->numbnests; j >= 1; j--)
    if (xtra1 > (double)c->xln[j - 1] && xtra1 < (double)c->xrn[j - 1] && ytra1 > (double)c->yln[j - 1] && ytra1 < (double)c->yrn[j - 1]) { ngrid = j; break; }
  if (ngrid > 0 && !c->lsprecn) ngrid = 0;   /* scenario without nest precipitation fields: mother grid only (test set-ups) */
  if (ngrid > 0) {   /* :97-101 */
    xtn = (real)((xtra1 - (double)c->xln[ngrid - 1]) * (double)c->xresoln[ngrid - 1]);
    ytn = (real)((ytra1 - (double)c->yln[ngrid - 1]) * (double)c->yresoln[ngrid - 1]);
    ix = (int)xtn; jy = (int)ytn;
  } else {
    ix = (int)xtra1; jy = (int)ytra1;
  }
  interp_time = (int)lround((double)((real)itime - K(0.5) * (real)ltsample));   /* nint(itime-0.5*ltsample) */
  n = c->memind[1];
  if (abs(c->memtime[0] - interp_time) < abs(c->memtime[1] - interp_time)) n = c->memind[0];
  if (ngrid == 0) orc_interpol_rain(c, c->lsprec, c->convprec, c->tcc, n, (real)xtra1, (real)ytra1, &lsp, &convp, &cc);
  else orc_interpol_rain_nests(c, c->lsprecn, c->convprecn, c->tccn, ngrid, n, xtn, ytn, &lsp, &convp, &cc);
  if (lsp < K(0.01) && convp < K(0.01)) return wetscav;
  for (il = 2; il <= c->nz; il++)
    if (HGT(il) > ztra1) { hz = il - 1; break; }
  if (ngrid == 0) clouds_v = (int)c->clouds[(((size_t)(n - 1) * c->nz + (size_t)(hz - 1)) * c->ny + (size_t)jy) * c->nx + (size_t)ix];
  else clouds_v = (int)N3(c->cloudsn, ix, jy, hz, n, ngrid);
  if (clouds_v <= 1) return wetscav;
  if (lsp > K(20.)) i = 5; else if (lsp > K(8.)) i = 4; else if (lsp > K(3.)) i = 3; else if (lsp > K(1.)) i = 2; else i = 1;
  if (convp > K(20.)) j = 5; else if (convp > K(8.)) j = 4; else if (convp > K(3.)) j = 3; else if (convp > K(1.)) j = 2; else j = 1;
  grfraction[0] = r_max(K(0.05), cc * (lsp * lfr[i - 1] + convp * cfr[j - 1]) / (lsp + convp));
  prec1 = (lsp + convp) / grfraction[0];
  act_temp = ngrid > 0 ? N3(c->ttn, ix, jy, hz, n, ngrid) : F3(c->tt, ix, jy, hz, n);   /* :197-201 */
  if (clouds_v >= 4) {   /* below cloud */
    if (c->dquer[ks] <= K(0.) && (c->weta_gas[ks] > K(0.) || c->wetb_gas[ks] > K(0.))) {
      c->blc_count[ks]++;
      wetscav = c->weta_gas[ks] * r_pow(prec1, c->wetb_gas[ks]);
    } else if (c->dquer[ks] > K(0.) && (c->crain_aero[ks] > K(0.) || c->csnow_aero[ks] > K(0.))) {
      real l10;
      c->blc_count[ks]++;
      dquer_m = r_min(K(10.), c->dquer[ks]) / K(1000000.);
      l10 = r_log10(dquer_m);
      if (act_temp >= K(273.) && c->crain_aero[ks] > K(0.))
        wetscav = c->crain_aero[ks] * r_pow10(bclr[0] + (bclr[1] * (K(1.) / ((l10 * l10) * (l10 * l10)))) + (bclr[2] * (K(1.) / (l10 * (l10 * l10)))) +
                                              (bclr[3] * (K(1.) / (l10 * l10))) + (bclr[4] * (K(1.) / l10)) + bclr[5] * r_pow(prec1, K(0.5)));
      else if (act_temp < K(273.) && c->csnow_aero[ks] > K(0.))
        wetscav = c->csnow_aero[ks] * r_pow10(bcls[0] + (bcls[1] * (K(1.) / ((l10 * l10) * (l10 * l10)))) + (bcls[2] * (K(1.) / (l10 * (l10 * l10)))) +
                                              (bcls[3] * (K(1.) / (l10 * l10))) + (bcls[4] * (K(1.) / l10)) + bcls[5] * r_pow(prec1, K(0.5)));
    }
  }
  if (clouds_v < 4) {   /* in cloud */
    if ((c->ccn_aero[ks] > K(0.) || c->in_aero[ks] > K(0.)) || (c->henry[ks] > K(0.) && c->dquer[ks] <= K(0.))) {
      c->inc_count[ks]++;
      if (c->ccn_aero[ks] < K(0.)) c->ccn_aero[ks] = K(0.);
      if (c->in_aero[ks] < K(0.)) c->in_aero[ks] = K(0.);
      if (ngrid == 0 && c->readclouds) cl = F2(c->ctwc, ix, jy, n) * (grfraction[0] / cc);
      else cl = K(1E6) * K(2E-7) * r_pow(prec1, K(0.36));
      if (act_temp <= K(253.)) { liq_frac = K(0); ice_frac = K(1); }
      else if (act_temp >= K(273.)) { liq_frac = K(1); ice_frac = K(0); }
      else {
        real t = (act_temp - K(273.)) / (K(273.) - K(253.));
        ice_frac = t * t;
        liq_frac = r_max(K(0.), K(1.) - ice_frac);
      }
      frac_act = liq_frac * c->ccn_aero[ks] + ice_frac * c->in_aero[ks];
      if (c->dquer[ks] > K(0.)) S_i = frac_act / cl;
      else {
        cle = (K(1) - cl) / (c->henry[ks] * (r_air / K(3500.)) * act_temp) + cl;
        S_i = K(1) / cle;
      }
      wetscav = incloud_ratio * S_i * (prec1 / K(3.6E6));
    }
  }
  return wetscav;
}

/* wetdepokernel.f90:38-108 */
static void orc_wetdepokernel(orc_ctx *c, int nunc, const real *deposit, real x, real y, int nage, int kp) {
  if (nage > c->nageclass) return;   /* guard: the reference would write past the last age plane (particle older than lage(nageclass)) */
  real xl, yl, ddx, ddy, wx, wy, w;
  int ix, jy, ixp, jyp, ks;
  xl = (x * c->dx + c->xoutshift) / c->dxout;
  yl = (y * c->dy + c->youtshift) / c->dyout;
  ix = (int)xl; jy = (int)yl;
  ddx = xl - (real)ix; ddy = yl - (real)jy;
  if (ddx > K(0.5)) { ixp = ix + 1; wx = K(1.5) - ddx; } else { ixp = ix - 1; wx = K(0.5) + ddx; }
  if (ddy > K(0.5)) { jyp = jy + 1; wy = K(1.5) - ddy; } else { jyp = jy - 1; wy = K(0.5) + ddy; }
#define WADD(i, j, ww) c->wetgridunc[DIDX(i, j, ks, kp, nunc, nage)] = (dep_real)((real)c->wetgridunc[DIDX(i, j, ks, kp, nunc, nage)] + (ww))
  for (ks = 1; ks <= c->nspec; ks++) {
    if (!c->lusekerneloutput) {
      if (ix >= 0 && jy >= 0 && ix <= c->numxgrid - 1 && jy <= c->numygrid - 1) WADD(ix, jy, deposit[ks - 1]);
      continue;
    }
    if (ix >= 0 && jy >= 0 && ix <= c->numxgrid - 1 && jy <= c->numygrid - 1) { w = wx * wy; WADD(ix, jy, deposit[ks - 1] * w); }
    if (ixp >= 0 && jyp >= 0 && ixp <= c->numxgrid - 1 && jyp <= c->numygrid - 1) { w = (K(1.) - wx) * (K(1.) - wy); WADD(ixp, jyp, deposit[ks - 1] * w); }
    if (ixp >= 0 && jy >= 0 && ixp <= c->numxgrid - 1 && jy <= c->numygrid - 1) { w = (K(1.) - wx) * wy; WADD(ixp, jy, deposit[ks - 1] * w); }
    if (ix >= 0 && jyp >= 0 && ix <= c->numxgrid - 1 && jyp <= c->numygrid - 1) { w = wx * (K(1.) - wy); WADD(ix, jyp, deposit[ks - 1] * w); }
  }
#undef WADD
}

/* wetdepokernel_nest.f90:38-107: floor() instead of int(), always the uniform kernel */
static void orc_wetdepokernel_nest(orc_ctx *c, int nunc, const real *deposit, real x, real y, int nage, int kp) {
  if (nage > c->nageclass) return;   /* guard: the reference would write past the last age plane (particle older than lage(nageclass)) */
  real xl, yl, ddx, ddy, wx, wy, w;
  int ix, jy, ixp, jyp, ks;
  xl = (x * c->dx + c->xoutshiftn) / c->dxoutn;
  yl = (y * c->dy + c->youtshiftn) / c->dyoutn;
  ix = (int)floor((double)xl); jy = (int)floor((double)yl);
  ddx = xl - (real)ix; ddy = yl - (real)jy;
  if (ddx > K(0.5)) { ixp = ix + 1; wx = K(1.5) - ddx; } else { ixp = ix - 1; wx = K(0.5) + ddx; }
  if (ddy > K(0.5)) { jyp = jy + 1; wy = K(1.5) - ddy; } else { jyp = jy - 1; wy = K(0.5) + ddy; }
#define WNADD(i, j, ww) c->wetgriduncn[ORC_DIDXN(i, j, ks, kp, nunc, nage)] = (dep_real)((real)c->wetgriduncn[ORC_DIDXN(i, j, ks, kp, nunc, nage)] + (ww))
  for (ks = 1; ks <= c->nspec; ks++) {
    if (ix >= 0 && jy >= 0 && ix <= c->numxgridn - 1 && jy <= c->numygridn - 1) { w = wx * wy; WNADD(ix, jy, deposit[ks - 1] * w); }
    if (ixp >= 0 && jyp >= 0 && ixp <= c->numxgridn - 1 && jyp <= c->numygridn - 1) { w = (K(1.) - wx) * (K(1.) - wy); WNADD(ixp, jyp, deposit[ks - 1] * w); }
    if (ixp >= 0 && jy >= 0 && ixp <= c->numxgridn - 1 && jy <= c->numygridn - 1) { w = (K(1.) - wx) * wy; WNADD(ixp, jy, deposit[ks - 1] * w); }
    if (ix >= 0 && jyp >= 0 && ix <= c->numxgridn - 1 && jyp <= c->numygridn - 1) { w = wx * (K(1.) - wy); WNADD(ix, jyp, deposit[ks - 1] * w); }
  }
#undef WNADD
}

/* wetdepo.f90:58-151 */
void orc_wetdepo(orc_ctx *c, int itime, int ltsample, int loutnext, int npart, const double *xtra1, const double *ytra1,
                 const real *ztra1, const int *itra1, const int *itramem, const int *npoint, const int *nclass, real *xmass1) {
  const real smallnum = sizeof(real) == 4 ? (real)1.17549435e-38f : (real)2.2250738585072014e-308;
  int jpart, ks, ldeltat;
  if (itime <= loutnext) ldeltat = itime - (loutnext - c->loutstep); else ldeltat = itime - loutnext;
  for (jpart = 0; jpart < npart; jpart++) {
    int itage, nage, kp = 1;
    real wetdeposit[ORC_MAXSPEC], grfraction[3], restmass;
    if (itra1[jpart] == -999999999) continue;
    if (c->ldirect == 1) { if (itra1[jpart] > itime) continue; } else { if (itra1[jpart] < itime) continue; }
    itage = abs(itra1[jpart] - itramem[jpart]);
    nage = orc_ageclass(c, itage);
    for (ks = 0; ks < ORC_MAXSPEC; ks++) wetdeposit[ks] = K(0.);
    for (ks = 0; ks < c->nspec; ks++) {
      real wetscav;
      if (!c->wetdepspec[ks]) continue;
      grfraction[0] = K(0.);
      wetscav = orc_get_wetscav(c, itime, ltsample, xtra1[jpart], ytra1[jpart], ztra1[jpart], ks, grfraction);
      if (wetscav > K(0.)) wetdeposit[ks] = xmass1[(size_t)ks * npart + jpart] * (K(1.) - r_exp(-wetscav * (real)abs(ltsample))) * grfraction[0];
      else wetdeposit[ks] = K(0.);
      restmass = xmass1[(size_t)ks * npart + jpart] - wetdeposit[ks];
      kp = c->ioutputforeachrelease == 1 ? npoint[jpart] : 1;
      if (restmass > smallnum) xmass1[(size_t)ks * npart + jpart] = restmass; else xmass1[(size_t)ks * npart + jpart] = K(0.);
      if (c->decay[ks] > K(0.)) wetdeposit[ks] = wetdeposit[ks] * r_exp((real)abs(ldeltat) * c->decay[ks]);
    }
    if (c->ldirect == 1 && c->wetgridunc) orc_wetdepokernel(c, nclass[jpart], wetdeposit, (real)xtra1[jpart], (real)ytra1[jpart], nage, kp);
    if (c->ldirect == 1 && c->nested_output == 1) orc_wetdepokernel_nest(c, nclass[jpart], wetdeposit, (real)xtra1[jpart], (real)ytra1[jpart], nage, kp);   /* wetdepo.f90:142 */
  }
}

void orc_set_wet(orc_ctx *c, const int *wetdepspec, const double *weta_gas, const double *wetb_gas, const double *crain_aero,
                 const double *csnow_aero, const double *ccn_aero, const double *in_aero, const double *henry, int readclouds,
                 const real *lsprec, const real *convprec, const real *tcc, const signed char *clouds, const int *cloudsh,
                 const real *ctwc) {
  int i;
  for (i = 0; i < c->nspec && i < ORC_MAXSPEC; i++) {
    c->wetdepspec[i] = wetdepspec[i];
    c->weta_gas[i] = (real)weta_gas[i]; c->wetb_gas[i] = (real)wetb_gas[i]; c->crain_aero[i] = (real)crain_aero[i];
    c->csnow_aero[i] = (real)csnow_aero[i]; c->ccn_aero[i] = (real)ccn_aero[i]; c->in_aero[i] = (real)in_aero[i];
    c->henry[i] = (real)henry[i];
  }
  c->readclouds = readclouds;
  c->lsprec = lsprec; c->convprec = convprec; c->tcc = tcc; c->clouds = clouds; c->cloudsh = cloudsh; c->ctwc = ctwc;
}
const dep_real *orc_wetgridunc(orc_ctx *c) { return c->wetgridunc; }
/* precipitation / cloud / temperature fields of nest 1 (readclouds_nest = .false.) */
void orc_set_wet_nest(orc_ctx *c, const real *lsprecn, const real *convprecn, const real *tccn, const signed char *cloudsn, const real *ttn) {
  c->lsprecn = lsprecn; c->convprecn = convprecn; c->tccn = tccn; c->cloudsn = cloudsn; c->ttn = ttn;
}

void orc_set_outgrid(orc_ctx *c, int numxgrid, int numygrid, int numzgrid, double dxout, double dyout, double outlon0,
                     double outlat0, const double *outheight, int maxpointspec_act, int nclassunc, int nageclass,
                     const int *lage, int ind_samp, int ioutputforeachrelease, int lusekerneloutput, int maxspec_out) {
  int k;
  size_t n2, n3;
  c->numxgrid = numxgrid; c->numygrid = numygrid; c->numzgrid = numzgrid;
  c->dxout = (real)dxout; c->dyout = (real)dyout;
  c->xoutshift = c->xlon0 - (real)outlon0;   /* readoutgrid.f90:199-200 */
  c->youtshift = c->ylat0 - (real)outlat0;
  for (k = 0; k < numzgrid; k++) c->outheight[k] = (real)outheight[k];
  c->maxpointspec_act = maxpointspec_act; c->nclassunc = nclassunc; c->nageclass = nageclass;
  for (k = 0; k < nageclass && k < 8; k++) c->lage[k] = lage[k];
  c->ind_samp = ind_samp; c->ioutputforeachrelease = ioutputforeachrelease; c->lusekerneloutput = lusekerneloutput;
  c->maxspec_out = maxspec_out;
  n2 = (size_t)numxgrid * numygrid * maxspec_out * maxpointspec_act * nclassunc * nageclass;
  n3 = n2 * numzgrid;
  free(c->gridunc); free(c->drygridunc);
  c->gridunc = (real *)calloc(n3, sizeof(real));
  c->drygridunc = (dep_real *)calloc(n2, sizeof(dep_real));
  free(c->wetgridunc);
  c->wetgridunc = (dep_real *)calloc(n2, sizeof(dep_real));
}
/* nested output grid, readoutgrid_nest.f90 + outgrid_init_nest.f90 (after orc_set_outgrid) */
void orc_set_outgrid_nest(orc_ctx *c, int numxgridn, int numygridn, double dxoutn, double dyoutn, double outlon0n, double outlat0n) {
  size_t n2, n3;
  c->nested_output = 1;
  c->numxgridn = numxgridn; c->numygridn = numygridn;
  c->dxoutn = (real)dxoutn; c->dyoutn = (real)dyoutn;
  c->xoutshiftn = c->xlon0 - (real)outlon0n;
  c->youtshiftn = c->ylat0 - (real)outlat0n;
  n2 = (size_t)numxgridn * numygridn * c->maxspec_out * c->maxpointspec_act * c->nclassunc * c->nageclass;
  n3 = n2 * c->numzgrid;
  free(c->griduncn); free(c->drygriduncn); free(c->wetgriduncn);
  c->griduncn = (real *)calloc(n3, sizeof(real));
  c->drygriduncn = (dep_real *)calloc(n2, sizeof(dep_real));
  c->wetgriduncn = (dep_real *)calloc(n2, sizeof(dep_real));
}
const real *orc_griduncn(orc_ctx *c) { return c->griduncn; }
const dep_real *orc_drygriduncn(orc_ctx *c) { return c->drygriduncn; }
const dep_real *orc_wetgriduncn(orc_ctx *c) { return c->wetgriduncn; }
/* receptor points in grid coordinates (readreceptors.f90:88-92) */
void orc_set_receptors(orc_ctx *c, int n, const double *x, const double *y, const double *area) {
  int i;
  c->numreceptor = n < ORC_MAXRECEPTOR ? n : ORC_MAXRECEPTOR;
  for (i = 0; i < c->numreceptor; i++) { c->xreceptor[i] = (real)x[i]; c->yreceptor[i] = (real)y[i]; c->receptorarea[i] = (real)area[i]; }
  for (i = 0; i < ORC_MAXRECEPTOR * ORC_MAXSPEC; i++) c->creceptor[i] = K(0.);
}
/* creceptor(n, ks) -> out[n + numreceptor*ks] */
void orc_get_receptors(orc_ctx *c, double *out) {
  int n, ks;
  for (ks = 0; ks < c->nspec; ks++)
    for (n = 0; n < c->numreceptor; n++) out[n + c->numreceptor * ks] = (double)c->creceptor[n + ORC_MAXRECEPTOR * ks];
}
void orc_set_output_times(orc_ctx *c, int loutnext, int loutstep) { c->loutnext = loutnext; c->loutstep = loutstep; }
/* what concoutput.f90:719-720 (concoutput_nest.f90) does after writing: gridunc, griduncn and creceptor start again
   from zero, the deposition grids keep accumulating */
void orc_clear_gridunc(orc_ctx *c) {
  size_t n2 = (size_t)c->numxgrid * c->numygrid * c->maxspec_out * c->maxpointspec_act * c->nclassunc * c->nageclass;
  int i;
  if (c->gridunc) memset(c->gridunc, 0, sizeof(real) * n2 * c->numzgrid);
  if (c->nested_output && c->griduncn)
    memset(c->griduncn, 0, sizeof(real) * (size_t)c->numxgridn * c->numygridn * c->maxspec_out * c->maxpointspec_act * c->nclassunc * c->nageclass * c->numzgrid);
  for (i = 0; i < ORC_MAXRECEPTOR * ORC_MAXSPEC; i++) c->creceptor[i] = K(0.);
}
const real *orc_gridunc(orc_ctx *c) { return c->gridunc; }
const dep_real *orc_drygridunc(orc_ctx *c) { return c->drygridunc; }

/* ------------------------------------------------------------------------- */
/* public C entry points (ctypes)                                              */
/* ------------------------------------------------------------------------- */
orc_ctx *orc_create(void) {
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  c->idummy_init = -7;
  c->idummy_adv = -7;
  c->ldirect = 1;
  c->numpoint = 0; c->xmass_pt = NULL; c->npart_pt = NULL; c->mquasilag = 0;
  c->lage_last = 999999999;
  c->eps_nxmax = K(361);
  orc_fill_rannumb(c);
  return c;
}
void orc_destroy(orc_ctx *c) { free(c->xmass_pt); free(c->npart_pt); free(c->zpoint1); free(c->zpoint2); free(c); }
int orc_real_size(void) { return (int)sizeof(real); }
const real *orc_rannumb(orc_ctx *c) { return &c->rannumb[1]; }

void orc_set_grid(orc_ctx *c, int nx, int ny, int nz, double dx, double dy, double xlon0, double ylat0,
                  int xglobal, int nglobal, int sglobal, const double *height, int nmixz) {
  int k;
  const real r_earth = K(6.371e6);
  c->nx = nx; c->ny = ny; c->nz = nz; c->nxmin1 = nx - 1; c->nymin1 = ny - 1; c->nmixz = nmixz;
  c->dx = (real)dx; c->dy = (real)dy; c->xlon0 = (real)xlon0; c->ylat0 = (real)ylat0;
  c->dxconst = K(180.) / (c->dx * r_earth * PI_PAR);   /* gridcheck_ecmwf.f90:311-312 */
  c->dyconst = K(180.) / (c->dy * r_earth * PI_PAR);
  c->xglobal = xglobal; c->nglobal = nglobal; c->sglobal = sglobal;
  c->switchnorthg = nglobal ? (K(75.) - c->ylat0) / c->dy : K(999999.);   /* gridcheck_ecmwf.f90:348,362 */
  c->switchsouthg = sglobal ? (K(-75.) - c->ylat0) / c->dy : K(999999.);
  for (k = 0; k < nz; k++) c->height[k] = (real)height[k];
}
/* cmapf_mod.f90:780-814 */
static void orc_stlmbr(real *s, real tnglat, real xlong) {
  real xi, eta;
  s[0] = r_sin(CM_RADPDG * tnglat);
  s[1] = orc_cspanf(xlong, K(-180.), K(180.));
  s[2] = K(0.); s[3] = K(0.); s[4] = K(1.); s[5] = K(0.);
  s[6] = CM_REARTH;
  orc_cnllxy(s, K(89.), xlong, &xi, &eta);
  s[7] = K(2.) * eta - s[0] * eta * eta;
  orc_cnllxy(s, K(-89.), xlong, &xi, &eta);
  s[8] = K(2.) * eta - s[0] * eta * eta;
}
/* cmapf_mod.f90:603-633 */
static void orc_stcm2p(real *s, real x1, real y1, real xlat1, real xlong1, real x2, real y2, real xlat2, real xlong2) {
  real x1a, y1a, x2a, y2a, den, dena;
  int k;
  for (k = 2; k < 6; k++) s[k] = K(0.);
  s[4] = K(1.);
  s[6] = K(1.);
  orc_cll2xy(s, xlat1, xlong1, &x1a, &y1a);
  orc_cll2xy(s, xlat2, xlong2, &x2a, &y2a);
  den = r_sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
  dena = r_sqrt((x1a - x2a) * (x1a - x2a) + (y1a - y2a) * (y1a - y2a));
  s[4] = ((x1a - x2a) * (x1 - x2) + (y1a - y2a) * (y1 - y2)) / den / dena;
  s[5] = ((y1a - y2a) * (x1 - x2) - (x1a - x2a) * (y1 - y2)) / den / dena;
  s[6] = s[6] * dena / den;
  orc_cll2xy(s, xlat1, xlong1, &x1a, &y1a);
  s[2] = s[2] + x1 - x1a;
  s[3] = s[3] + y1 - y1a;
}
/* the polar-cap maps as gridcheck_ecmwf.f90:341-366 sets them up */
void orc_make_polemaps(orc_ctx *c) {
  const real switchnorth = K(75.), switchsouth = K(-75.);
  real sizesouth = K(6.) * (switchsouth + K(90.)) / c->dy;
  real sizenorth = K(6.) * (K(90.) - switchnorth) / c->dy;
  orc_stlmbr(c->southpolemap, K(-90.), K(0.));
  orc_stcm2p(c->southpolemap, K(0.), K(0.), switchsouth, K(0.), sizesouth, sizesouth, switchsouth, K(180.));
  orc_stlmbr(c->northpolemap, K(90.), K(0.));
  orc_stcm2p(c->northpolemap, K(0.), K(0.), switchnorth, K(0.), sizenorth, sizenorth, switchnorth, K(180.));
}
void orc_set_dy_for_polemaps(orc_ctx *c, double dy) { c->dy = (real)dy; }
void orc_get_polemaps(orc_ctx *c, double *north, double *south) {
  int i;
  for (i = 0; i < 9; i++) { north[i] = (double)c->northpolemap[i]; south[i] = (double)c->southpolemap[i]; }
}
void orc_set_polemaps(orc_ctx *c, const double *north, const double *south) {
  int i;
  for (i = 0; i < 9; i++) { c->northpolemap[i] = (real)north[i]; c->southpolemap[i] = (real)south[i]; }
}
void orc_set_time(orc_ctx *c, int memtime1, int memtime2, int memind1, int memind2) {
  c->memtime[0] = memtime1; c->memtime[1] = memtime2; c->memind[0] = memind1; c->memind[1] = memind2;
  c->lwindinterv = abs(memtime2 - memtime1);
}
void orc_set_switches(orc_ctx *c, int ldirect, int lsynctime, int method, int mintime, double ctl, int ifine,
                      int turbswitch, int cblflag, int mdomainfill, int lsettling, int nspec, int drydep,
                      const int *drydepspec, double d_trop, double d_strat, double turbmesoscale) {
  int i;
  c->ldirect = ldirect; c->lsynctime = lsynctime; c->method = method; c->mintime = mintime;
  c->ctl = (real)ctl; c->ifine = ifine; c->fine = K(1.) / (real)ifine;
  c->turbswitch = turbswitch; c->cblflag = cblflag; c->mdomainfill = mdomainfill; c->lsettling = lsettling;
  c->nspec = nspec; c->drydep = drydep;
  for (i = 0; i < nspec && i < ORC_MAXSPEC; i++) c->drydepspec[i] = drydepspec ? drydepspec[i] : 0;
  c->d_trop = (real)d_trop; c->d_strat = (real)d_strat; c->turbmesoscale = (real)turbmesoscale;
}
void orc_set_species(orc_ctx *c, const double *density, const double *dquer, const double *vsetaver,
                     const double *cunningham, const double *decay, int lage_last, int mquasilag) {
  int i;
  for (i = 0; i < c->nspec && i < ORC_MAXSPEC; i++) {
    c->density[i] = (real)density[i]; c->dquer[i] = (real)dquer[i]; c->vsetaver[i] = (real)vsetaver[i];
    c->cunningham[i] = (real)cunningham[i]; c->decay[i] = (real)decay[i];
  }
  c->lage_last = lage_last; c->mquasilag = mquasilag;
}
/* point_mod xmass(numpoint,maxspec) (given [nspec][numpoint]) and npart(numpoint), readreleases.f90:239-243,419-421 */
void orc_set_release_points(orc_ctx *c, int numpoint, const double *xmass, const int *npart) {
  int i;
  free(c->xmass_pt); free(c->npart_pt);
  c->numpoint = numpoint;
  c->xmass_pt = (real *)malloc(sizeof(real) * (size_t)numpoint * c->nspec);
  c->npart_pt = (int *)malloc(sizeof(int) * (size_t)numpoint);
  for (i = 0; i < numpoint * c->nspec; i++) c->xmass_pt[i] = (real)xmass[i];
  for (i = 0; i < numpoint; i++) c->npart_pt[i] = npart[i];
}
/* field pointers (arrays stay owned by the caller, in the oracle's precision) */
void orc_set_fields(orc_ctx *c, const real *uu, const real *vv, const real *ww, const real *rho, const real *drhodz,
                    const real *tt, const real *uupol, const real *vvpol, const real *hmix, const real *ustar,
                    const real *wstar, const real *oli, const real *tropopause, const real *vdep) {
  c->uu = uu; c->vv = vv; c->ww = ww; c->rho = rho; c->drhodz = drhodz; c->tt = tt; c->uupol = uupol; c->vvpol = vvpol;
  c->hmix = hmix; c->ustar = ustar; c->wstar = wstar; c->oli = oli; c->tropopause = tropopause; c->vdep = vdep;
}

/* One synchronisation step over all particles in index order: the particle
   loop of timemanager.f90:531-712 (initialize if new :553-555, advance :609-611,
   epilogue :630-708 without the deposition-grid kernels).  prob_out (npart*nspec,
   species-major) receives advance's dry-deposition probabilities or may be NULL. */
/* get_vdep_prob.f90:4-145 -- "probability" (in fact the deposition velocity, :136-139) of dry deposition at the
   receptor, used by backward runs with DRYBKDEP.  The routine sets ngrid, ix, jy, ixp, jyp but NOT the horizontal
   weights p1..p4 and the time weights dt1, dt2, dtt that interpol_vdep[_nests] reads: those are whatever interpol_mod
   holds -- left by initialize() of the same particle (timemanager.f90:553-556 runs just before; mother-grid weights
   also for a particle inside a nest, as initialize() knows no nests). */
static void orc_get_vdep_prob(orc_ctx *c, int itime, double xt, double yt, real zt, real *prob) {
  const real eps = c->eps_nxmax / K(3.e5);
  const real href = K(15.);
  int ks;
  real xtn, ytn, vdepo[ORC_MAXSPEC];
  (void)itime;
  if (c->drydep)
    for (ks = 0; ks < c->nspec; ks++) { c->depoindicator[ks] = 1; prob[ks] = K(0.); }
  c->ngrid = orc_pick_grid(c, xt, yt, eps);   /* :52-69 */
  if (c->ngrid > 0) {                          /* :84-99 */
    xtn = (real)((xt - (double)c->xln[c->ngrid - 1]) * (double)c->xresoln[c->ngrid - 1]);
    ytn = (real)((yt - (double)c->yln[c->ngrid - 1]) * (double)c->yresoln[c->ngrid - 1]);
    c->ix = (int)xtn; c->jy = (int)ytn;
  } else {
    c->ix = (int)xt; c->jy = (int)yt;
  }
  c->ixp = c->ix + 1;
  c->jyp = c->jy + 1;
  if (c->drydep && zt < K(2.) * href) {        /* :105-127 */
    for (ks = 0; ks < c->nspec; ks++) {
      if (c->drydepspec[ks]) {
        if (c->depoindicator[ks]) {
          if (c->ngrid <= 0) orc_interpol_vdep(c, ks + 1, &vdepo[ks]);
          else orc_interpol_vdep_nests(c, ks + 1, &vdepo[ks]);
        }
        prob[ks] = vdepo[ks];
      }
    }
  }
}

void orc_set_bkdep(orc_ctx *c, int drybkdep, int wetbkdep, int numpoint, const double *zpoint1, const double *zpoint2) {
  int i;
  c->drybkdep = drybkdep; c->wetbkdep = wetbkdep;
  free(c->zpoint1); free(c->zpoint2);
  c->zpoint1 = (real *)calloc((size_t)(numpoint > 0 ? numpoint : 1), sizeof(real));
  c->zpoint2 = (real *)calloc((size_t)(numpoint > 0 ? numpoint : 1), sizeof(real));
  for (i = 0; i < numpoint; i++) { c->zpoint1[i] = (real)zpoint1[i]; c->zpoint2[i] = (real)zpoint2[i]; }
}

/* the receptor block of timemanager.f90:564-598: once per particle, before it is moved for the first time */
static void orc_receptor_scavenging(orc_ctx *c, int itime, int j, int npart, double xt, double yt, real zt, int npoint,
                                    real *xmass1, real *xscav_frac1) {
  int ks;
  if (c->drybkdep) {
    for (ks = 0; ks < c->nspec; ks++) {
      if (xscav_frac1[(size_t)ks * npart + j] < K(0.)) {
        real prob_rec[ORC_MAXSPEC];
        orc_get_vdep_prob(c, itime, xt, yt, zt, prob_rec);
        if (c->drydepspec[ks]) xscav_frac1[(size_t)ks * npart + j] = prob_rec[ks];
        else { xmass1[(size_t)ks * npart + j] = K(0.); xscav_frac1[(size_t)ks * npart + j] = K(0.); }
      }
    }
  }
  if (c->wetbkdep) {
    for (ks = 0; ks < c->nspec; ks++) {
      if (xscav_frac1[(size_t)ks * npart + j] < K(0.)) {
        real grfraction[3] = {K(0.), K(0.), K(0.)};
        const real wetscav = orc_get_wetscav(c, itime, c->lsynctime, xt, yt, zt, ks, grfraction);
        if (wetscav > K(0.))
          xscav_frac1[(size_t)ks * npart + j] = wetscav * (c->zpoint2[npoint - 1] - c->zpoint1[npoint - 1]) * grfraction[0];
        else { xmass1[(size_t)ks * npart + j] = K(0.); xscav_frac1[(size_t)ks * npart + j] = K(0.); }
      }
    }
  }
}

long orc_step(orc_ctx *c, int itime, int npart, double *xtra1, double *ytra1, real *ztra1,
              real *uap, real *ucp, real *uzp, real *us, real *vs, real *ws,
              int *idt, int *itra1, const int *itramem, const int *npoint, int16_t *cbt,
              real *xmass1, real *prob_out, const int *nclass_arr, real *xscav_frac1) {
  const real minmass = K(0.0001);
  long nadv = 0;
  int j, ks, nstop;
  real prob[ORC_MAXSPEC];
  for (j = 0; j < npart; j++) {
    if (itra1[j] != itime) continue;
    c->cur_particle = j;
    if (itramem[j] == itime || itime == 0)
      orc_initialize(c, itime, &idt[j], &uap[j], &ucp[j], &uzp[j], &us[j], &vs[j], &ws[j], xtra1[j], ytra1[j], ztra1[j], &cbt[j]);
    if ((c->drybkdep || c->wetbkdep) && xscav_frac1 && xmass1)   /* timemanager.f90:564-598 */
      orc_receptor_scavenging(c, itime, j, npart, xtra1[j], ytra1[j], ztra1[j], npoint ? npoint[j] : 1, xmass1, xscav_frac1);
    for (ks = 0; ks < ORC_MAXSPEC; ks++) prob[ks] = K(0.);
    nstop = orc_advance(c, itime, npoint ? npoint[j] : 1, &idt[j], &uap[j], &ucp[j], &uzp[j], &us[j], &vs[j], &ws[j],
                        &xtra1[j], &ytra1[j], &ztra1[j], prob, &cbt[j]);
    nadv++;
    if (prob_out)
      for (ks = 0; ks < c->nspec; ks++) prob_out[(size_t)ks * npart + j] = prob[ks];
    if (nstop > 1) {
      itra1[j] = -999999999;
    } else {
      real xmassfract = K(0.), decfact;
      dep_real drydeposit[ORC_MAXSPEC];
      /* age class and release pointer taken at the start of the step, timemanager.f90:539-548 */
      const int nage = orc_ageclass(c, abs(itime - itramem[j]));
      const int kp = (c->ioutputforeachrelease == 1 && npoint) ? npoint[j] : 1;
      /* timemanager.f90:513-517 */
      const int ldeltat = itime < c->loutnext ? itime - (c->loutnext - c->loutstep) : itime - c->loutnext;
      itra1[j] = itime + c->lsynctime;
      for (ks = 0; ks < c->nspec; ks++) {
        if (c->decay[ks] > K(0.)) decfact = r_exp(-(real)abs(c->lsynctime) * c->decay[ks]);
        else decfact = K(1.);
        drydeposit[ks] = 0.f;
        if (xmass1) {
          if (c->drydepspec[ks]) {
            drydeposit[ks] = (dep_real)(xmass1[(size_t)ks * npart + j] * prob[ks] * decfact);
            xmass1[(size_t)ks * npart + j] = xmass1[(size_t)ks * npart + j] * (K(1.) - prob[ks]) * decfact;
            if (c->decay[ks] > K(0.)) drydeposit[ks] = (dep_real)((real)drydeposit[ks] * r_exp((real)abs(ldeltat) * c->decay[ks]));
          } else xmass1[(size_t)ks * npart + j] = xmass1[(size_t)ks * npart + j] * decfact;
          if (c->mdomainfill == 0 && c->mquasilag == 0) {   /* timemanager.f90:663-666 */
            const int kr = npoint ? npoint[j] : 1;
            if (XMASS_PT(kr, ks + 1) > K(0.)) xmassfract = r_max(xmassfract, (real)c->npart_pt[kr - 1] * xmass1[(size_t)ks * npart + j] / XMASS_PT(kr, ks + 1));
          } else xmassfract = K(1.0);
        } else xmassfract = K(1.0);
      }
      if (xmassfract < minmass) itra1[j] = -999999999;
      /* timemanager.f90:690-696 (forward runs only) */
      if (c->drydep && c->ldirect == 1 && c->drygridunc && xmass1)
        orc_drydepokernel(c, nclass_arr ? nclass_arr[j] : 1, drydeposit, (real)xtra1[j], (real)ytra1[j], nage, kp);
      if (c->drydep && c->ldirect == 1 && c->nested_output == 1 && xmass1)   /* timemanager.f90:694-696 */
        orc_drydepokernel_nest(c, nclass_arr ? nclass_arr[j] : 1, drydeposit, (real)xtra1[j], (real)ytra1[j], nage, kp);
      if (abs(itra1[j] - itramem[j]) >= c->lage_last) itra1[j] = -999999999;
    }
  }
  return nadv;
}

void orc_set_eps_nxmax(orc_ctx *c, int nxmax) { c->eps_nxmax = (real)nxmax; }
/* one nest (index 1): geometry as gridcheck_nests.f90:362-378, compact arrays [slot][level][jyn][ixn] */
void orc_set_nest(orc_ctx *c, int nxn, int nyn, double dxn, double dyn, double xlon0n, double ylat0n,
                  const real *uun, const real *vvn, const real *wwn, const real *rhon, const real *drhodzn,
                  const real *hmixn, const real *ustarn, const real *wstarn, const real *olin, const real *tropopausen,
                  const real *vdepn) {
  real xaux1 = (real)xlon0n, yaux1 = (real)ylat0n, xaux2, yaux2;
  c->numbnests = 1;
  c->nxn[0] = nxn; c->nyn[0] = nyn;
  xaux2 = xaux1 + (real)(nxn - 1) * (real)dxn;
  yaux2 = yaux1 + (real)(nyn - 1) * (real)dyn;
  c->xresoln[0] = c->dx / (real)dxn;
  c->yresoln[0] = c->dy / (real)dyn;
  c->xln[0] = (xaux1 - c->xlon0) / c->dx;
  c->xrn[0] = (xaux2 - c->xlon0) / c->dx;
  c->yln[0] = (yaux1 - c->ylat0) / c->dy;
  c->yrn[0] = (yaux2 - c->ylat0) / c->dy;
  c->uun[0] = uun; c->vvn[0] = vvn; c->wwn[0] = wwn; c->rhon[0] = rhon; c->drhodzn[0] = drhodzn;
  c->hmixn[0] = hmixn; c->ustarn[0] = ustarn; c->wstarn[0] = wstarn; c->olin[0] = olin; c->tropopausen[0] = tropopausen;
  c->vdepn[0] = vdepn;
}
void orc_set_parallel_semantics(orc_ctx *c, int on) { c->parallel_semantics = on; }
void orc_set_com_parameters(orc_ctx *c, int turboff, int interpolhmix) { c->turboff = turboff; c->interpolhmix = interpolhmix; }
void orc_set_leak_flags(orc_ctx *c, unsigned char *flags) { c->leak_flags = flags; }

long orc_nan_count(orc_ctx *c, int which) { return which == 2 ? c->nan_count2 : c->nan_count; }

/* expose the shared ran3 stream (tests of the host-side replica in the product) */
double orc_ran3_next(orc_ctx *c, int *idum) { return (double)orc_ran3(c, idum); }
/* the module's one ran3 state (random_mod.f90:46-52, 93-139) out of / into the context, in the layout of convect_oracle.c's
   cvo_args.state_words -- words[1..55] = ma, [56] inext, [57] inextp, [58] iff -- so that a test can hand the stream from
   advance to redist and back, as the two routines share it in the reference */
void orc_get_ran3_state(orc_ctx *c, int *words) {
  int i;
  for (i = 1; i <= 55; i++) words[i] = c->ran3_ma[i];
  words[56] = c->ran3_inext; words[57] = c->ran3_inextp; words[58] = c->ran3_iff;
}
void orc_set_ran3_state(orc_ctx *c, const int *words) {
  int i;
  for (i = 1; i <= 55; i++) c->ran3_ma[i] = words[i];
  c->ran3_inext = words[56]; c->ran3_inextp = words[57]; c->ran3_iff = words[58];
}
void orc_reset_rng(orc_ctx *c) { c->ran3_iff = 0; c->idummy_init = -7; c->idummy_adv = -7; c->gasdev_iset = 0; c->gasdev_gset = 0; }
