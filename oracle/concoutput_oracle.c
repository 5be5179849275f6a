/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the part of the reference's concoutput that
 * writes the sparse concentration files grid_conc_<date><time>_<species>.
 * Only tests/ and the cpu_baseline / checker leg of tools/bench_*.py may load this; the product path (flexpart_amd/) never does.
 *
 * Plain C restatement of /root/reference/src/concoutput.f90:226-228 (factor3d), :296-345 (class mean of
 * wetgridunc, drygridunc, gridunc; mean_mod.f90:mean_sp) and :349-447 (the three run-length compressed
 * dumps: start index of every run of non-zero cells, values with a sign that flips from run to run), for a
 * forward run (ldirect=1), iout=1, maxpointspec_act=1, nageclass=1, in the reference's own real kind
 * (float: with -fdefault-real-8 concoutput.f90 does not compile).  Pinned byte for byte against the files
 * written by the flang build of the unmodified routine (oracle/_ref/coref_r4, oracle/ref_co_driver.f90).
 * Compile with -ffp-contract=off.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
  int nxg, nyg, nzg, nspec, nclassunc, wetdep, drydep, itime;
  float outnum;
  const float *area;        /* [nyg][nxg] */
  const float *volume;      /* [nzg][nyg][nxg] */
  const float *gridunc;     /* [nclassunc][nspec][nzg][nyg][nxg] */
  const float *wetgridunc;  /* [nclassunc][nspec][nyg][nxg] */
  const float *drygridunc;
  /* mixing-ratio files (iout = 2, 3): met grid, z levels, air density of slot memind(2), molar weights */
  int nx, ny, nz;
  float dx, dy, xlon0, ylat0, dxout, dyout, outlon0, outlat0;
  const float *height;      /* [nz] */
  const float *outheight;   /* [nzg] */
  const float *rho;         /* [nz][ny][nx] */
  const float *weightmolar; /* [nspec] */
} coo_args;

static unsigned char *put(unsigned char *p, const void *s, size_t n) { memcpy(p, s, n); return p + n; }
static unsigned char *rec_i(unsigned char *p, const int32_t *v, int32_t n) {
  const int32_t l = 4 * n;
  p = put(p, &l, 4); p = put(p, v, (size_t)l); return put(p, &l, 4);
}
static unsigned char *rec_f(unsigned char *p, const float *v, int32_t n) {
  const int32_t l = 4 * n;
  p = put(p, &l, 4); p = put(p, v, (size_t)l); return put(p, &l, 4);
}

/* mean_mod.f90:mean_sp times nclassunc (concoutput.f90:323-325) */
static float class_sum_mean(const float *x, size_t stride, int n) {
  float xl = 0.f;
  int i;
  for (i = 0; i < n; i++) xl = xl + x[(size_t)i * stride];
  return (xl / (float)n) * (float)n;
}

/* one run-length compressed dump (:349-372): values v[i] in dump order, offset = index of the first cell */
static unsigned char *dump(unsigned char *p, const float *val, const float *scale, long ncell, int idx0, int32_t *wi, float *wr, float tot_mu, int conc) {
  /* conc: 0 deposition (scale = area), 1 concentration (scale = factor3d), 2 handled by dump_ppt below */
  const float smallnum = FLT_MIN;      /* tiny(0.0) */
  int32_t ci = 0, cr = 0;
  float sp_fact = -1.f;
  int sp_zer = 1;
  long i;
  for (i = 0; i < ncell; i++) {
    if (val[i] > smallnum) {
      if (sp_zer) { wi[ci++] = (int32_t)(i + idx0); sp_zer = 0; sp_fact = sp_fact * (-1.f); }
      wr[cr++] = conc ? sp_fact * val[i] * scale[i] / tot_mu : sp_fact * 1.e12f * val[i] / scale[i];
    } else sp_zer = 1;
  }
  p = rec_i(p, &ci, 1); p = rec_i(p, wi, ci);
  p = rec_i(p, &cr, 1); p = rec_f(p, wr, cr);
  return p;
}

/* file image of grid_conc_*_<ks+1>; work arrays wi, wr, g of nxg*nyg*nzg elements; returns the byte count */
long coo_concoutput(const coo_args *A, int ks, unsigned char *out, int32_t *wi, float *wr, float *g, float *f3) {
  const long n2 = (long)A->nxg * A->nyg, n3 = n2 * A->nzg;
  const size_t cls2 = (size_t)A->nspec * n2, cls3 = (size_t)A->nspec * n3;
  unsigned char *p = out;
  const int32_t it = A->itime, zero = 0;
  long i;
  for (i = 0; i < n3; i++) f3[i] = 1.e12f / A->volume[i] / A->outnum;          /* :226, ldirect = 1 */
  p = rec_i(p, &it, 1);                                                          /* write(unitoutgrid) itime */
  /* wet deposition */
  if (A->wetdep) {
    for (i = 0; i < n2; i++) g[i] = class_sum_mean(A->wetgridunc + (size_t)ks * n2 + i, cls2, A->nclassunc);
    p = dump(p, g, A->area, n2, 0, wi, wr, 1.f, 0);
  } else { p = rec_i(p, &zero, 1); p = rec_i(p, wi, 0); p = rec_i(p, &zero, 1); p = rec_f(p, wr, 0); }
  if (A->drydep) {
    for (i = 0; i < n2; i++) g[i] = class_sum_mean(A->drygridunc + (size_t)ks * n2 + i, cls2, A->nclassunc);
    p = dump(p, g, A->area, n2, 0, wi, wr, 1.f, 0);
  } else { p = rec_i(p, &zero, 1); p = rec_i(p, wi, 0); p = rec_i(p, &zero, 1); p = rec_f(p, wr, 0); }
  for (i = 0; i < n3; i++) g[i] = class_sum_mean(A->gridunc + (size_t)ks * n3 + i, cls3, A->nclassunc);
  p = dump(p, g, f3, n3, (int)n2 /* kz is 1-based in the index, :425 */, wi, wr, 1.f /* tot_mu, ldirect = 1 */, 1);
  return (long)(p - out);
}

/* densityoutgrid, concoutput.f90:176-205: air density at the centre of every output cell, nearest met column,
 * linear between the two z levels around the mid height of the output layer */
static void density_outgrid(const coo_args *A, float *dens) {
  int kz, jy, ix, kzz;
  for (kz = 1; kz <= A->nzg; kz++) {
    float halfheight, dz1, dz2, dz;
    if (kz == 1) halfheight = A->outheight[0] / 2.f;
    else halfheight = (A->outheight[kz - 1] + A->outheight[kz - 2]) / 2.f;
    for (kzz = 2; kzz <= A->nz; kzz++)
      if (A->height[kzz - 2] < halfheight && A->height[kzz - 1] > halfheight) break;
    kzz = kzz < A->nz ? kzz : A->nz;
    kzz = kzz > 2 ? kzz : 2;
    dz1 = halfheight - A->height[kzz - 2];
    dz2 = A->height[kzz - 1] - halfheight;
    dz = dz1 + dz2;
    for (jy = 0; jy < A->nyg; jy++)
      for (ix = 0; ix < A->nxg; ix++) {
        float xl = A->outlon0 + (float)ix * A->dxout, yl = A->outlat0 + (float)jy * A->dyout;
        int iix, jjy;
        xl = (xl - A->xlon0) / A->dx;
        yl = (yl - A->ylat0) / A->dy;
        iix = (int)lroundf(xl); jjy = (int)lroundf(yl);
        iix = iix < A->nx - 1 ? iix : A->nx - 1; iix = iix > 0 ? iix : 0;
        jjy = jjy < A->ny - 1 ? jjy : A->ny - 1; jjy = jjy > 0 ? jjy : 0;
        dens[(size_t)ix + (size_t)A->nxg * (jy + (size_t)A->nyg * (kz - 1))] =
            (A->rho[(size_t)iix + (size_t)A->nx * (jjy + (size_t)A->ny * (kzz - 1))] * dz1 +
             A->rho[(size_t)iix + (size_t)A->nx * (jjy + (size_t)A->ny * (kzz - 2))] * dz2) / dz;
      }
  }
}

/* file image of grid_pptv_*_<ks+1> (:482-590): the deposition dumps as in grid_conc, the 3-D dump as mixing ratio */
long coo_pptvoutput(const coo_args *A, int ks, unsigned char *out, int32_t *wi, float *wr, float *g, float *dens) {
  const float smallnum = FLT_MIN, weightair = 28.97f;
  const long n2 = (long)A->nxg * A->nyg, n3 = n2 * A->nzg;
  const size_t cls2 = (size_t)A->nspec * n2, cls3 = (size_t)A->nspec * n3;
  unsigned char *p = out;
  const int32_t it = A->itime, zero = 0;
  int32_t ci = 0, cr = 0;
  float sp_fact = -1.f;
  int sp_zer = 1;
  long i;
  density_outgrid(A, dens);
  p = rec_i(p, &it, 1);
  if (A->wetdep) {
    for (i = 0; i < n2; i++) g[i] = class_sum_mean(A->wetgridunc + (size_t)ks * n2 + i, cls2, A->nclassunc);
    p = dump(p, g, A->area, n2, 0, wi, wr, 1.f, 0);
  } else { p = rec_i(p, &zero, 1); p = rec_i(p, wi, 0); p = rec_i(p, &zero, 1); p = rec_f(p, wr, 0); }
  if (A->drydep) {
    for (i = 0; i < n2; i++) g[i] = class_sum_mean(A->drygridunc + (size_t)ks * n2 + i, cls2, A->nclassunc);
    p = dump(p, g, A->area, n2, 0, wi, wr, 1.f, 0);
  } else { p = rec_i(p, &zero, 1); p = rec_i(p, wi, 0); p = rec_i(p, &zero, 1); p = rec_f(p, wr, 0); }
  for (i = 0; i < n3; i++) {
    const float v = class_sum_mean(A->gridunc + (size_t)ks * n3 + i, cls3, A->nclassunc);
    if (v > smallnum) {
      if (sp_zer) { wi[ci++] = (int32_t)(i + n2); sp_zer = 0; sp_fact = sp_fact * (-1.f); }
      wr[cr++] = sp_fact * 1.e12f * v / A->volume[i] / A->outnum * weightair / A->weightmolar[ks] / dens[i];
    } else sp_zer = 1;
  }
  p = rec_i(p, &ci, 1); p = rec_i(p, wi, ci);
  p = rec_i(p, &cr, 1); p = rec_f(p, wr, cr);
  return (long)(p - out);
}
