/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's readpartpositions
 * (the warm start from the particle dump partposit_end).  Only tests/ and the cpu_baseline / checker leg of tools/bench_*.py may
 * load this; the product path (flexpart_amd/) never does.
 *
 * Plain C restatement of /root/reference/src/readpartpositions.f90:115-148 (the part after the
 * header file: parsing the dump written by partoutput.f90, lon/lat -> grid coordinates, release
 * time re-based on the new run's start, idt/itra1/nclass/itrasplit) with juldate.f90 and
 * random_mod.f90:ran1.  Pinned against the flang build of the unmodified routine
 * (oracle/_ref/rpref_r4|r8 through oracle/ref_rp_driver.f90).
 * Built twice: -DORC_REAL=float / -DORC_REAL=double, -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real;
#define K(x) ((real)(x))

/* juldate.f90 */
double rpo_juldate(int yyyymmdd, int hhmiss) {
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

/* random_mod.f90:12-42 */
typedef struct { int iv[32]; int iy; } rpo_ran1_state;
static real rpo_ran1(rpo_ran1_state *S, int *idum) {
  const int ia = 16807, im = 2147483647, iq = 127773, ir = 2836, ntab = 32, ndiv = 1 + (2147483647 - 1) / 32;
  const real am = K(1.) / (real)im, eps = K(1.2e-7), rnmx = K(1.) - eps;
  int j, k;
  real r;
  if (*idum <= 0 || S->iy == 0) {
    *idum = -*idum > 1 ? -*idum : 1;
    for (j = ntab + 8; j >= 1; j--) {
      k = *idum / iq;
      *idum = ia * (*idum - k * iq) - ir * k;
      if (*idum < 0) *idum = *idum + im;
      if (j <= ntab) S->iv[j - 1] = *idum;
    }
    S->iy = S->iv[0];
  }
  k = *idum / iq;
  *idum = ia * (*idum - k * iq) - ir * k;
  if (*idum < 0) *idum = *idum + im;
  j = 1 + S->iy / ndiv;
  S->iy = S->iv[j - 1];
  S->iv[j - 1] = *idum;
  r = am * (real)S->iy;
  return r < rnmx ? r : rnmx;
}

typedef struct {
  int nspec, ldirect, mintime, itsplit, nclassunc;
  int ibdatein, ibtimein;       /* from the header file */
  double bdate;                 /* start of this run (julian) */
  double dx, dy, xlon0, ylat0;
  long maxpart;
  /* outputs */
  long numpart;
  int numparticlecount, itimein, status;   /* status 1: julin differs from bdate (the reference stops) */
  double *xtra1, *ytra1, *ztra1, *xmass1;  /* xmass1 [nspec][maxpart] */
  int *npoint, *itramem, *nclass, *idt, *itra1, *itrasplit;
} rpo_args;

/* file: the bytes of partposit_end; returns 0, or -1 on a malformed file */
int rpo_readpartpositions(rpo_args *A, const unsigned char *file, long nbytes) {
  const int nspec = A->nspec;
  const int reclen = 8 + (10 + nspec) * (int)sizeof(real);
  const real dx = (real)A->dx, dy = (real)A->dy, xlon0 = (real)A->xlon0, ylat0 = (real)A->ylat0;
  const unsigned char *p = file, *end = file + nbytes;
  long i = 0, k;
  int ks, idummy = -8;
  rpo_ran1_state S;
  double julin;
  memset(&S, 0, sizeof S);
  A->numparticlecount = 0; A->status = 0; A->itimein = 0;
  while (p < end) {                                            /* 100 read(unitpartin,end=99) itimein */
    int32_t l;
    if (end - p < 12) return -1;
    memcpy(&l, p, 4); if (l != 4) return -1;
    memcpy(&A->itimein, p + 4, 4); p += 12;
    i = 0;
    for (;;) {                                                   /* 200 */
      real xlonin, ylatin, v;
      int32_t iv;
      if (end - p < reclen + 8) return -1;
      memcpy(&l, p, 4); if (l != reclen) return -1;
      if (i >= A->maxpart) return -1;
      p += 4;
      memcpy(&iv, p, 4); p += 4; A->npoint[i] = iv;
      memcpy(&xlonin, p, sizeof(real)); p += sizeof(real);
      memcpy(&ylatin, p, sizeof(real)); p += sizeof(real);
      memcpy(&v, p, sizeof(real)); p += sizeof(real); A->ztra1[i] = (double)v;
      memcpy(&iv, p, 4); p += 4; A->itramem[i] = iv;
      p += 7 * sizeof(real);                                     /* topo,pvi,qvi,rhoi,hmixi,tri,tti */
      for (ks = 0; ks < nspec; ks++) { memcpy(&v, p, sizeof(real)); p += sizeof(real); A->xmass1[(size_t)ks * A->maxpart + i] = (double)v; }
      p += 4;
      i++;
      if (xlonin == K(-9999.9)) break;                           /* goto 100 */
      A->xtra1[i - 1] = (double)((xlonin - xlon0) / dx);
      A->ytra1[i - 1] = (double)((ylatin - ylat0) / dy);
      if (A->npoint[i - 1] > A->numparticlecount) A->numparticlecount = A->npoint[i - 1];
    }
  }
  A->numpart = i - 1;                                            /* 99 */
  julin = rpo_juldate(A->ibdatein, A->ibtimein) + (double)A->itimein / 86400.;
  if (fabs(julin - A->bdate) > 1.e-5) { A->status = 1; return 0; }
  for (k = 0; k < A->numpart; k++) {
    const double julpartin = rpo_juldate(A->ibdatein, A->ibtimein) + (double)A->itramem[k] / 86400.;
    const real r = rpo_ran1(&S, &idummy);
    int nc = (int)(r * (real)A->nclassunc) + 1;
    A->nclass[k] = nc < A->nclassunc ? nc : A->nclassunc;
    A->idt[k] = A->mintime;
    A->itra1[k] = 0;
    A->itramem[k] = (int)lround((julpartin - A->bdate) * (double)K(86400.));
    A->itrasplit[k] = A->ldirect * A->itsplit;
  }
  return 0;
}
