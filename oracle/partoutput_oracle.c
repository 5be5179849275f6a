/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's partoutput
 * (the binary particle dump partposit_*).  Only tests/ and the cpu_baseline / checker leg of tools/bench_*.py may load this;
 * the product path (flexpart_amd/) never does.
 *
 * Plain C restatement of /root/reference/src/partoutput.f90:63-190: for every particle due at
 * itime the bilinear+time+vertical interpolation of oro, pv, qv, tt, rho, hmix, tropopause to the
 * particle position, and the Fortran sequential-unformatted record stream (4-byte length
 * markers before and after every record, as flang/gfortran write them) the routine produces.
 * Pinned byte for byte against the file written by the flang build of the unmodified routine
 * (oracle/_ref/poref_r4|r8 through oracle/ref_po_driver.f90).
 *
 * Built twice: -DORC_REAL=float / -DORC_REAL=double, -ffp-contract=off.
 * Fields are compact [slot][level][jy][ix]; elements beyond nx, ny read as 0 (the zero padding of
 * the reference's nxmax/nymax arrays in the driver).
 */
#include <stdint.h>
#include <string.h>

#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real;
#define K(x) ((real)(x))

typedef struct {
  int nx, ny, nz, nymax, nspec, itime;
  int memtime[2], memind[2];
  double dx, dy, xlon0, ylat0;
  const double *height;                          /* [nz] */
  const double *oro;                             /* [ny][nx] */
  const double *pv, *qv, *tt, *rho;              /* [2][nz][ny][nx] */
  const double *hmix, *tropopause;               /* [2][ny][nx] */
  long numpart;
  const double *xtra1, *ytra1, *ztra1;           /* [numpart] */
  const int *itra1, *itramem, *npoint;
  const double *xmass1;                          /* [nspec][numpart] */
} poo_args;

static unsigned char *put(unsigned char *p, const void *src, size_t n) { memcpy(p, src, n); return p + n; }

static real f3(const poo_args *A, const double *f, int ix, int jy, int k /*1-based*/, int slot /*1-based*/) {
  if (ix >= A->nx || jy >= A->ny) return K(0.);
  return (real)f[(size_t)ix + (size_t)A->nx * ((size_t)jy + (size_t)A->ny * ((size_t)(k - 1) + (size_t)A->nz * (size_t)(slot - 1)))];
}
static real f2(const poo_args *A, const double *f, int ix, int jy, int slot) {
  if (ix >= A->nx || jy >= A->ny) return K(0.);
  return (real)f[(size_t)ix + (size_t)A->nx * ((size_t)jy + (size_t)A->ny * (size_t)(slot - 1))];
}

/* returns the number of bytes written to out (the file image), or -1 if cap is too small */
long poo_partoutput(const poo_args *A, unsigned char *out, long cap) {
  const int nspec = A->nspec;
  const int32_t reclen = (int32_t)(2 * 4 + (10 + nspec) * (int)sizeof(real));
  const real dx = (real)A->dx, dy = (real)A->dy, xlon0 = (real)A->xlon0, ylat0 = (real)A->ylat0;
  const real dt1 = (real)(A->itime - A->memtime[0]), dt2 = (real)(A->memtime[1] - A->itime);   /* partoutput.f90:69-71 */
  const real dtt = K(1.) / (dt1 + dt2);
  unsigned char *p = out;
  long i;
  int indz = 1, indzp = 2;
  {
    const int32_t four = 4, it = A->itime;
    if (cap < 12) return -1;
    p = put(p, &four, 4); p = put(p, &it, 4); p = put(p, &four, 4);            /* write(unitpartout) itime, :90 */
  }
  for (i = 0; i < A->numpart; i++) {
    real xlon, ylat, zt, ddx, ddy, rddx, rddy, p1, p2, p3, p4, topo, dz1, dz2, dz;
    real pvprof[2], qvprof[2], ttprof[2], rhoprof[2], pvi, qvi, tti, rhoi, hm[2], tr[2], hmixi, tri;
    int ix, jy, ixp, jyp, il, ind, m, ks;
    int32_t iv;
    if (A->itra1[i] != A->itime) continue;                                      /* :98 */
    if (p - out + reclen + 8 > cap) return -1;
    xlon = (real)((double)xlon0 + A->xtra1[i] * (double)dx);
    ylat = (real)((double)ylat0 + A->ytra1[i] * (double)dy);
    zt = (real)A->ztra1[i];
    ix = (int)A->xtra1[i]; jy = (int)A->ytra1[i];
    ixp = ix + 1; jyp = jy + 1;
    ddx = (real)(A->xtra1[i] - (double)(real)ix);
    ddy = (real)(A->ytra1[i] - (double)(real)jy);
    rddx = K(1.) - ddx; rddy = K(1.) - ddy;
    p1 = rddx * rddy; p2 = ddx * rddy; p3 = rddx * ddy; p4 = ddx * ddy;
    if (jyp >= A->nymax) jyp = jyp - 1;                                        /* :119-121 */
    topo = p1 * f2(A, A->oro, ix, jy, 1) + p2 * f2(A, A->oro, ixp, jy, 1) + p3 * f2(A, A->oro, ix, jyp, 1) + p4 * f2(A, A->oro, ixp, jyp, 1);
    for (il = 2; il <= A->nz; il++)                                            /* :131-138; keeps the last indices if none */
      if ((real)A->height[il - 1] > zt) { indz = il - 1; indzp = il; break; }
    dz1 = zt - (real)A->height[indz - 1];
    dz2 = (real)A->height[indzp - 1] - zt;
    dz = K(1.) / (dz1 + dz2);
    for (ind = indz; ind <= indzp; ind++) {
      real pv1[2], qv1[2], tt1[2], rho1[2];
      for (m = 0; m < 2; m++) {
        const int h = A->memind[m];
        pv1[m] = p1 * f3(A, A->pv, ix, jy, ind, h) + p2 * f3(A, A->pv, ixp, jy, ind, h) + p3 * f3(A, A->pv, ix, jyp, ind, h) + p4 * f3(A, A->pv, ixp, jyp, ind, h);
        qv1[m] = p1 * f3(A, A->qv, ix, jy, ind, h) + p2 * f3(A, A->qv, ixp, jy, ind, h) + p3 * f3(A, A->qv, ix, jyp, ind, h) + p4 * f3(A, A->qv, ixp, jyp, ind, h);
        tt1[m] = p1 * f3(A, A->tt, ix, jy, ind, h) + p2 * f3(A, A->tt, ixp, jy, ind, h) + p3 * f3(A, A->tt, ix, jyp, ind, h) + p4 * f3(A, A->tt, ixp, jyp, ind, h);
        rho1[m] = p1 * f3(A, A->rho, ix, jy, ind, h) + p2 * f3(A, A->rho, ixp, jy, ind, h) + p3 * f3(A, A->rho, ix, jyp, ind, h) + p4 * f3(A, A->rho, ixp, jyp, ind, h);
      }
      pvprof[ind - indz] = (pv1[0] * dt2 + pv1[1] * dt1) * dtt;
      qvprof[ind - indz] = (qv1[0] * dt2 + qv1[1] * dt1) * dtt;
      ttprof[ind - indz] = (tt1[0] * dt2 + tt1[1] * dt1) * dtt;
      rhoprof[ind - indz] = (rho1[0] * dt2 + rho1[1] * dt1) * dtt;
    }
    pvi = (dz1 * pvprof[1] + dz2 * pvprof[0]) * dz;
    qvi = (dz1 * qvprof[1] + dz2 * qvprof[0]) * dz;
    tti = (dz1 * ttprof[1] + dz2 * ttprof[0]) * dz;
    rhoi = (dz1 * rhoprof[1] + dz2 * rhoprof[0]) * dz;
    for (m = 0; m < 2; m++) {
      const int h = A->memind[m];
      tr[m] = p1 * f2(A, A->tropopause, ix, jy, h) + p2 * f2(A, A->tropopause, ixp, jy, h) + p3 * f2(A, A->tropopause, ix, jyp, h) + p4 * f2(A, A->tropopause, ixp, jyp, h);
      hm[m] = p1 * f2(A, A->hmix, ix, jy, h) + p2 * f2(A, A->hmix, ixp, jy, h) + p3 * f2(A, A->hmix, ix, jyp, h) + p4 * f2(A, A->hmix, ixp, jyp, h);
    }
    hmixi = (hm[0] * dt2 + hm[1] * dt1) * dtt;
    tri = (tr[0] * dt2 + tr[1] * dt1) * dtt;
    /* :177-179 */
    p = put(p, &reclen, 4);
    iv = A->npoint[i]; p = put(p, &iv, 4);
    p = put(p, &xlon, sizeof(real)); p = put(p, &ylat, sizeof(real)); p = put(p, &zt, sizeof(real));
    iv = A->itramem[i]; p = put(p, &iv, 4);
    p = put(p, &topo, sizeof(real)); p = put(p, &pvi, sizeof(real)); p = put(p, &qvi, sizeof(real)); p = put(p, &rhoi, sizeof(real));
    p = put(p, &hmixi, sizeof(real)); p = put(p, &tri, sizeof(real)); p = put(p, &tti, sizeof(real));
    for (ks = 0; ks < nspec; ks++) { real xm = (real)A->xmass1[(size_t)ks * A->numpart + i]; p = put(p, &xm, sizeof(real)); }
    p = put(p, &reclen, 4);
  }
  {   /* the closing record, :182-184 */
    const int32_t m5 = -99999;
    const real m4 = K(-9999.9);
    int j;
    if (p - out + reclen + 8 > cap) return -1;
    p = put(p, &reclen, 4);
    p = put(p, &m5, 4);
    for (j = 0; j < 3; j++) p = put(p, &m4, sizeof(real));
    p = put(p, &m5, 4);
    for (j = 0; j < 7 + nspec; j++) p = put(p, &m4, sizeof(real));
    p = put(p, &reclen, 4);
  }
  return (long)(p - out);
}
