// Device side of fpx_convmix: the reference's convective mixing of particles --
// convmix.f90:61-196 (particles by grid column), calcmatrix.f90:56-137 (the redistribution matrix of a column, ECMWF
// branch), convect43c.f90 (Emanuel's scheme CONVECT + TLIFT as the reference ships it), redist.f90:49-236 (the
// displacement of a particle) -- as HIP kernels for gfx950.
//
// Design.  The scheme is a long computation per grid column (level loops with data-dependent bounds, O(levels^2)
// matrices) and only columns that hold particles are computed:
//   k_conv_mark      one lane per particle: its column (nint of the grid coordinates), a flag per column
//   (scan)           the columns that hold particles, in grid order
//   k_conv_column_a  one lane per such column: profiles at the particle time, CONVECT up to its early exits
//   (scan)           the columns that go on
//   k_conv_prelude, k_conv_rows, k_conv_cols, k_conv_flux, k_conv_matrix
//                    the mixing part of CONVECT, the matrix fmassfrac and the heights of the half levels: the part that is
//                    sequential along the sounding with a lane per column, the matrices with a lane per (column, level)
//                    (k_conv_column_b: all of it with one lane per column -- the check of the level kernels)
//   k_conv_redist    one lane per particle of a convective column: level search, one uniform random number, the walk
//                    along its matrix row, new height or the compensating subsidence.
// All per-column arrays (26 vectors, 3 matrices) live in HBM scratch interleaved in groups of 64 columns (struct Scr), so
// that the lanes of a wave, which walk the same loops, touch one contiguous segment per access.  The matrices exist for
// batches of surviving columns that fit the scratch budget.
// Arithmetic is in the host's real kind H (the reference computes in its default real) with FMA contraction off and
// in the reference's order of operations; only libm (exp, log, pow) can differ from the CPU result.
#pragma once
#include "fpx_tu.hpp"
#include <hip/hip_runtime.h>

#include "fpx_calcpar.hpp"
#include "fpx_verttransform.hpp"

namespace fpx {
FPX_TU_OPEN
namespace conv {

using vt::M;
#define HK(x) ((H)(x))
#define R_ABS(a) ((a) < 0 ? -(a) : (a))
#define R_MAX(a, b) ((a) > (b) ? (a) : (b))
#define R_MIN(a, b) ((a) < (b) ? (a) : (b))
#define I_MAX(a, b) ((a) > (b) ? (a) : (b))
#define I_MIN(a, b) ((a) < (b) ? (a) : (b))

// the per-column arrays (1-based index i in 0..nv-1 is used as in the Fortran; element 0 is unused)
enum Vec { V_fup, V_fdown, V_m, V_tvp, V_tv, V_ep, V_clw, V_sigp, V_tp, V_cpn, V_lv,
           V_h, V_hp, V_gz, V_hm, V_nent, V_tconv, V_qconv, V_qsconv, V_pconv_hpa, V_phconv_hpa, V_sub, V_pconv,
           V_phconv, V_dpr, V_uvzlev, V_COUNT };
enum Mat { M_fmass, M_ment, M_sij, M_COUNT };
constexpr int M_fmassfrac = M_fmass;      // calcmatrix scales fmass into fmassfrac in place

// Scratch layout: columns are interleaved in groups of kGroup = 64 (one wave): element e of column c lives at
// [(c / 64) * elems_per_column * 64 + e * 64 + c % 64].  The lanes of a wave, which walk the same loops, touch one 256 / 512
// byte segment per access, and all arrays of a group lie within a few MB (29 MB for the three fp64 matrices of 138 levels)
// -- interleaving over the whole batch instead put consecutive levels 32 MB apart and every access on another page.
constexpr int kGroup = 64;
template <typename H>
struct Scr {
  H *vb;         // this column's element 0 of the vectors  [group][V_COUNT][nv][64]: every column that holds particles
  H *mb;         // this column's element 0 of the matrices [group][M_COUNT][nv][nv][64], column-major A(i,j) -> [j][i]: only
                 // the columns that reach the mixing computation
  int nv, ml;    // ml: this column's lane in its matrix group
  __device__ Scr(H *v, H *mat, int /*B*/, int c, int nv_, int /*Bm*/, int cm)
      : vb(v + (size_t)(c / kGroup) * V_COUNT * nv_ * kGroup + (c % kGroup)),
        mb(mat ? mat + (size_t)(cm / kGroup) * M_COUNT * nv_ * nv_ * kGroup + (cm % kGroup) : nullptr), nv(nv_), ml(cm % kGroup) {}
  // fmassfrac in walk-major form (k_conv_matrix_walk, _walk_t): the group's FMASS block re-used as 64 matrices [column][levold][nv], so
  // that the entries a particle walks along -- row levold of a forward run, column levold of a backward one -- are consecutive in memory
  __device__ H *fm_walk_group() const { return mb - ml + (size_t)M_fmass * nv * nv * kGroup; }
  __device__ H *fm_walk() const { return fm_walk_group() + (size_t)ml * nv * nv; }
};
// what the first part of CONVECT (up to its early exits) hands to the second
template <typename H>
struct CvState { int nk, icb, inb, iflag; H plcl, cbmf; };
#define VV(name, i) Sx.vb[((size_t)V_##name * Sx.nv + (size_t)(i)) * kGroup]
#define MM(name, i, j) Sx.mb[(((size_t)M_##name * Sx.nv + (size_t)(j)) * Sx.nv + (size_t)(i)) * kGroup]

__host__ __device__ inline size_t group_round(size_t n) { return (n + kGroup - 1) / kGroup * kGroup; }   // columns of whole groups
template <typename H>
__host__ __device__ inline size_t vec_elems_per_column(int nv) { return (size_t)V_COUNT * nv; }
template <typename H>
__host__ __device__ inline size_t mat_elems_per_column(int nv) { return (size_t)M_COUNT * nv * nv; }

template <typename H>
__device__ void tlift(const Scr<H> &Sx, int icb, int nk, int nl, int kk) {
#pragma clang fp contract(off)
  const H cpd = HK(1005.7), cpv = HK(1870.0), cl = HK(2500.0), rv = HK(461.5), rd = HK(287.04), lv0 = HK(2.501e6);
  const H cpvmcl = cl - cpv, eps0 = rd / rv, epsi = HK(1.) / eps0;
  H ah0, ahg, alv, cpinv, cpp, denom, es, qg, rg, s, tc, tg;
  int i, j, nsb, nst;
  ah0 = (cpd * (HK(1.) - VV(qconv, nk)) + cl * VV(qconv, nk)) * VV(tconv, nk) + VV(qconv, nk) * (lv0 - cpvmcl * (VV(tconv, nk) - HK(273.15))) + VV(gz, nk);
  cpp = cpd * (HK(1.) - VV(qconv, nk)) + VV(qconv, nk) * cpv;
  cpinv = HK(1.) / cpp;
  if (kk == 1) {
    for (i = 1; i <= icb - 1; i++) VV(clw, i) = HK(0.0);
    for (i = nk; i <= icb - 1; i++) {
      VV(tp, i) = VV(tconv, nk) - (VV(gz, i) - VV(gz, nk)) * cpinv;
      VV(tvp, i) = VV(tp, i) * (HK(1.) + VV(qconv, nk) * epsi);
    }
  }
  nst = icb;
  nsb = icb;
  if (kk == 2) { nst = nl; nsb = icb + 1; }
  // (values that the reference stores and reads back at once stay in registers, the inputs of eight levels are requested together)
  const H q_nk = VV(qconv, nk);
  for (int i0 = nsb; i0 <= nst; i0 += 8) {
    H tcv[8], qsv[8], gzv[8], ppv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int ii = I_MIN(i0 + u, nst);
      tcv[u] = VV(tconv, ii); qsv[u] = VV(qsconv, ii); gzv[u] = VV(gz, ii); ppv[u] = VV(pconv_hpa, ii);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      i = i0 + u;
      if (i > nst) break;
      const H t_i = tcv[u], gz_i = gzv[u];
      tg = t_i;
      qg = qsv[u];
      alv = lv0 - cpvmcl * (t_i - HK(273.15));
      for (j = 1; j <= 2; j++) {
        s = cpd + alv * alv * qg / (rv * t_i * t_i);
        s = HK(1.) / s;
        ahg = cpd * tg + (cl - cpd) * q_nk * t_i + alv * qg + gz_i;
        tg = tg + s * (ah0 - ahg);
        tg = R_MAX(tg, HK(35.0));
        tc = tg - HK(273.15);
        denom = HK(243.5) + tc;
        if (tc >= HK(0.0)) es = HK(6.112) * M<H>::exp(HK(17.67) * tc / denom);
        else es = M<H>::exp(HK(23.33086) - HK(6111.72784) / tg + HK(0.15215) * M<H>::log(tg));
        qg = eps0 * es / (ppv[u] - es * (HK(1.) - eps0));
      }
      alv = lv0 - cpvmcl * (t_i - HK(273.15));
      const H tp_i = (ah0 - (cl - cpd) * q_nk * t_i - gz_i - alv * qg) / cpd;
      VV(tp, i) = tp_i;
      H clw_i = q_nk - qg;
      clw_i = R_MAX(HK(0.0), clw_i);
      VV(clw, i) = clw_i;
      rg = qg / (HK(1.) - q_nk);
      VV(tvp, i) = tp_i * (HK(1.) + rg * epsi);
    }
  }
}


// The two row-wise parts of the mixing computation, convect43c.f90:590-655 and :660-720: row i of SIJ / MENT depends on
// the column's vectors only, never on another row -- the serial kernel calls them in loops over i, the parallel one
// gives every row its own lane.  The operands of kJ consecutive j are requested together; the arithmetic of each j is
// the reference's.
constexpr int kJ = 8;
template <typename H>
__device__ void sij_row(const Scr<H> &Sx, int nk, int icb, int inb, int i) {
#pragma clang fp contract(off)
  const H cpd = HK(1005.7), cpv = HK(1870.0), rv = HK(461.5);
  H qti, bf2, anum, denom, dei, altem, cwat, stemp;
  int j;
  // MENT is assigned in rows icb+1..inb, columns icb..inb only, SIJ read one column beyond on either side: the reference
  // zeroes (nl+1)^2 elements of five matrices per column, here the region that is read is zeroed (adding the exact zeros of
  // the rest changes no sum)
  for (j = icb - 1; j <= inb + 1; j++) {
    MM(sij, i, j) = HK(0.0);
    if (j >= icb && j <= inb) MM(ment, i, j) = HK(0.0);
  }
    qti = VV(qconv, nk) - VV(ep, i) * VV(clw, i);
    const H h_i = VV(h, i), hp_i = VV(hp, i), qc_i = VV(qconv, i), m_i = VV(m, i);
    int nent_i = 0;
    for (j = icb; j <= inb; j += kJ) {
      H lvj[kJ], qsj[kJ], tcj[kJ], hj[kJ], qcj[kJ], clwj[kJ], epj[kJ];
#pragma unroll
      for (int u = 0; u < kJ; u++) {
        const int jj = I_MIN(j + u, inb);
        lvj[u] = VV(lv, jj); qsj[u] = VV(qsconv, jj); tcj[u] = VV(tconv, jj); hj[u] = VV(h, jj); qcj[u] = VV(qconv, jj);
        clwj[u] = VV(clw, jj); epj[u] = VV(ep, jj);
      }
#pragma unroll
      for (int u = 0; u < kJ; u++) {
        const int jj = j + u;
        if (jj > inb) break;
        bf2 = HK(1.) + lvj[u] * lvj[u] * qsj[u] / (rv * tcj[u] * tcj[u] * cpd);
        anum = hj[u] - hp_i + (cpv - cpd) * tcj[u] * (qti - qcj[u]);
        denom = h_i - hp_i + (cpd - cpv) * (qc_i - qti) * tcj[u];
        dei = denom;
        if (R_ABS(dei) < HK(0.01)) dei = HK(0.01);
        H sv = anum / dei;
        if (jj == i) sv = HK(1.0);                      // SIJ(I,I)=1.0 is set inside the j loop of the reference (:604)
        altem = sv * qc_i + (HK(1.) - sv) * qti - qsj[u];
        altem = altem / bf2;
        cwat = clwj[u] * (HK(1.) - epj[u]);
        stemp = sv;
        if ((stemp < HK(0.0) || stemp > HK(1.0) || altem > cwat) && jj > i) {
          anum = anum - lvj[u] * (qti - qsj[u] - cwat * bf2);
          denom = denom + lvj[u] * (qc_i - qti);
          if (R_ABS(denom) < HK(0.01)) denom = HK(0.01);
          sv = anum / denom;
        }
        if (sv > HK(0.0) && sv < HK(0.9)) {
          MM(ment, i, jj) = m_i / (HK(1.) - sv);
          nent_i = nent_i + 1;
        }
        sv = R_MAX(HK(0.0), sv);
        sv = R_MIN(HK(1.0), sv);
        MM(sij, i, jj) = sv;
      }
    }
    VV(nent, i) = (H)nent_i;
    if (nent_i == 0) {
      MM(ment, i, i) = m_i;
      MM(sij, i, i) = HK(1.0);
    }
}

template <typename H>
__device__ void norm_row(const Scr<H> &Sx, int nk, int icb, int inb, int i) {
#pragma clang fp contract(off)
  H qp1, anum, denom, scrit, alt, asij, smin, smid, sjmax, sjmin, delp, delm, bsum;
  int j;
    if (VV(nent, i) != 0) {
      qp1 = VV(qconv, nk) - VV(ep, i) * VV(clw, i);
      anum = VV(h, i) - VV(hp, i) - VV(lv, i) * (qp1 - VV(qsconv, i));
      denom = VV(h, i) - VV(hp, i) + VV(lv, i) * (VV(qconv, i) - qp1);
      if (R_ABS(denom) < HK(0.01)) denom = HK(0.01);
      scrit = anum / denom;
      alt = qp1 - VV(qsconv, i) + scrit * (VV(qconv, i) - qp1);
      if (alt < HK(0.0)) scrit = HK(1.0);
      scrit = R_MAX(scrit, HK(0.0));
      asij = HK(0.0);
      smin = HK(1.0);
      for (j = icb; j <= inb; j += kJ) {
        H sw[kJ + 2], me[kJ], ph[kJ + 1];                // SIJ(i, j-1 .. j+kJ), MENT(i, j .. j+kJ-1), PHCONV_HPA(j .. j+kJ)
#pragma unroll
        for (int u = 0; u < kJ + 2; u++) sw[u] = MM(sij, i, I_MIN(j - 1 + u, inb + 1));
#pragma unroll
        for (int u = 0; u < kJ; u++) me[u] = MM(ment, i, I_MIN(j + u, inb));
#pragma unroll
        for (int u = 0; u < kJ + 1; u++) ph[u] = VV(phconv_hpa, I_MIN(j + u, inb + 1));
#pragma unroll
        for (int u = 0; u < kJ; u++) {
          const int jj = j + u;
          if (jj > inb) break;
          const H s0 = sw[u], s1 = sw[u + 1], s2 = sw[u + 2];      // SIJ(i,jj-1), SIJ(i,jj), SIJ(i,jj+1)
          if (s1 > HK(0.0) && s1 < HK(0.9)) {
            if (jj > i) {
              smid = R_MIN(s1, scrit);
              sjmax = smid;
              sjmin = smid;
              if (smid < smin && s2 < smid) {
                smin = smid;
                sjmax = R_MIN(R_MIN(s2, s1), scrit);
                sjmin = R_MAX(s0, s1);
                sjmin = R_MIN(sjmin, scrit);
              }
            } else {
              sjmax = R_MAX(s2, scrit);
              smid = R_MAX(s1, scrit);
              sjmin = HK(0.0);
              if (jj > 1) sjmin = s0;
              sjmin = R_MAX(sjmin, scrit);
            }
            delp = R_ABS(sjmax - smid);
            delm = R_ABS(sjmin - smid);
            asij = asij + (delp + delm) * (ph[u] - ph[u + 1]);
            MM(ment, i, jj) = me[u] * (delp + delm) * (ph[u] - ph[u + 1]);
          }
        }
      }
      asij = R_MAX(HK(1.0e-21), asij);
      asij = HK(1.0) / asij;
      bsum = HK(0.0);
      for (j = icb; j <= inb; j += 8) {
        H me[8];
#pragma unroll
        for (int u = 0; u < 8; u++) me[u] = j + u <= inb ? MM(ment, i, j + u) : HK(0.);
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (j + u <= inb) { const H v = me[u] * asij; MM(ment, i, j + u) = v; bsum = bsum + v; }
      }
      if (bsum < HK(1.0e-18)) {
        VV(nent, i) = 0;
        MM(ment, i, i) = VV(m, i);
        MM(sij, i, i) = HK(1.0);
      }
    }
}

// sij_row + norm_row in one sweep over j, for the level-parallel path: SIJ(i, .) is consumed by the normalisation of the
// same row only (a window of three neighbours), so it stays in registers -- the normalisation of entry j-1 follows the
// mixing fraction of entry j -- and MENT is written once before the final scaling instead of being written, re-read and
// rewritten: 24 instead of 56 bytes of HBM traffic per matrix entry (fp64).  A row without entrainment (nent = 0) has no
// SIJ in (0, 0.9), hence nothing for the normalisation to act on: running it unconditionally changes nothing.  Same
// operations on the same operands in the same order as the two functions above.
template <typename H>
struct MixRow {
  H qti, h_i, hp_i, qc_i, m_i, scrit, asij, smin;
  H s_a, s_b, me_b;                                      // SIJ(i,jj-2), SIJ(i,jj-1), MENT(i,jj-1) before the normalisation
  int nent_i, i, icb, inb;

  __device__ void init(const Scr<H> &Sx, int nk, int icb_, int inb_, int i_) {
#pragma clang fp contract(off)
    i = i_; icb = icb_; inb = inb_;
    qti = VV(qconv, nk) - VV(ep, i) * VV(clw, i);
    h_i = VV(h, i); hp_i = VV(hp, i); qc_i = VV(qconv, i); m_i = VV(m, i);
    const H lv_i = VV(lv, i), qs_i = VV(qsconv, i);
    H anum = h_i - hp_i - lv_i * (qti - qs_i);
    H denom = h_i - hp_i + lv_i * (qc_i - qti);
    if (R_ABS(denom) < HK(0.01)) denom = HK(0.01);
    scrit = anum / denom;
    const H alt = qti - qs_i + scrit * (qc_i - qti);
    if (alt < HK(0.0)) scrit = HK(1.0);
    scrit = R_MAX(scrit, HK(0.0));
    asij = HK(0.0);
    smin = HK(1.0);
    nent_i = 0;
    s_a = HK(0.); s_b = HK(0.); me_b = HK(0.);
  }

  // entry jj, icb <= jj <= inb + 1: the mixing fraction of (i, jj), then the normalisation of (i, jj - 1).
  // lvj .. qcj: LV, QSCONV, TCONV, H, QCONV at level min(jj, inb); bf2, cwat: the two level-only factors the prelude left in TP, HM;
  // ph0, ph1: PHCONV_HPA(jj - 1), PHCONV_HPA(jj)
  __device__ void step(const Scr<H> &Sx, int jj, H lvj, H qsj, H tcj, H hj, H qcj, H bf2, H cwat, H ph0, H ph1) {
#pragma clang fp contract(off)
    const H cpd = HK(1005.7), cpv = HK(1870.0);
    H anum, denom, dei, altem, stemp, smid, sjmax, sjmin, delp, delm;
    H sv = HK(0.), me = HK(0.);
    if (jj <= inb) {
      anum = hj - hp_i + (cpv - cpd) * tcj * (qti - qcj);
      denom = h_i - hp_i + (cpd - cpv) * (qc_i - qti) * tcj;
      dei = denom;
      if (R_ABS(dei) < HK(0.01)) dei = HK(0.01);
      sv = anum / dei;
      if (jj == i) sv = HK(1.0);
      altem = sv * qc_i + (HK(1.) - sv) * qti - qsj;
      altem = altem / bf2;
      stemp = sv;
      if ((stemp < HK(0.0) || stemp > HK(1.0) || altem > cwat) && jj > i) {
        anum = anum - lvj * (qti - qsj - cwat * bf2);
        denom = denom + lvj * (qc_i - qti);
        if (R_ABS(denom) < HK(0.01)) denom = HK(0.01);
        sv = anum / denom;
      }
      if (sv > HK(0.0) && sv < HK(0.9)) {
        me = m_i / (HK(1.) - sv);
        nent_i = nent_i + 1;
      }
      sv = R_MAX(HK(0.0), sv);
      sv = R_MIN(HK(1.0), sv);
    }
    if (jj > icb) {                                        // normalisation of entry jn = jj - 1
      const int jn = jj - 1;
      const H s0 = s_a, s1 = s_b, s2 = sv;
      H out = me_b;
      if (s1 > HK(0.0) && s1 < HK(0.9)) {
        if (jn > i) {
          smid = R_MIN(s1, scrit);
          sjmax = smid;
          sjmin = smid;
          if (smid < smin && s2 < smid) {
            smin = smid;
            sjmax = R_MIN(R_MIN(s2, s1), scrit);
            sjmin = R_MAX(s0, s1);
            sjmin = R_MIN(sjmin, scrit);
          }
        } else {
          sjmax = R_MAX(s2, scrit);
          smid = R_MAX(s1, scrit);
          sjmin = HK(0.0);
          if (jn > 1) sjmin = s0;
          sjmin = R_MAX(sjmin, scrit);
        }
        delp = R_ABS(sjmax - smid);
        delm = R_ABS(sjmin - smid);
        asij = asij + (delp + delm) * (ph0 - ph1);
        out = me_b * (delp + delm) * (ph0 - ph1);
      }
      MM(ment, i, jn) = out;
    }
    s_a = s_b; s_b = sv; me_b = me;
  }

  __device__ void finish(const Scr<H> &Sx) {
#pragma clang fp contract(off)
    if (nent_i == 0) {
      VV(nent, i) = 0;
      MM(ment, i, i) = m_i;
      return;
    }
    asij = R_MAX(HK(1.0e-21), asij);
    asij = HK(1.0) / asij;
    H bsum = HK(0.0);
    constexpr int kF = 8;                                 // entries requested together (16, 32: slower -- registers of the whole block)
    for (int j = icb; j <= inb; j += kF) {
      H me[kF];
#pragma unroll
      for (int u = 0; u < kF; u++) me[u] = j + u <= inb ? MM(ment, i, j + u) : HK(0.);
#pragma unroll
      for (int u = 0; u < kF; u++)
        if (j + u <= inb) { const H v = me[u] * asij; MM(ment, i, j + u) = v; bsum = bsum + v; }
    }
    if (bsum < HK(1.0e-18)) { nent_i = 0; MM(ment, i, i) = m_i; }
    VV(nent, i) = (H)nent_i;
  }
};

template <typename H>
__device__ void mix_row(const Scr<H> &Sx, int nk, int icb, int inb, int i) {
#pragma clang fp contract(off)
  MixRow<H> row;
  row.init(Sx, nk, icb, inb, i);
  for (int j = icb; j <= inb + 1; j += kJ) {
    H lvj[kJ], qsj[kJ], tcj[kJ], hj[kJ], qcj[kJ], clwj[kJ], epj[kJ], ph[kJ + 1];
#pragma unroll
    for (int u = 0; u < kJ; u++) {
      const int jj = I_MIN(j + u, inb);
      lvj[u] = VV(lv, jj); qsj[u] = VV(qsconv, jj); tcj[u] = VV(tconv, jj); hj[u] = VV(h, jj); qcj[u] = VV(qconv, jj);
      clwj[u] = VV(tp, jj); epj[u] = VV(hm, jj);           // BF2(jj), CWAT(jj)
    }
#pragma unroll
    for (int u = 0; u < kJ + 1; u++) ph[u] = VV(phconv_hpa, I_MIN(j - 1 + u, inb + 1));      // PHCONV_HPA(j-1 .. j+kJ-1)
#pragma unroll
    for (int u = 0; u < kJ; u++) {
      if (j + u > inb + 1) break;
      row.step(Sx, j + u, lvj[u], qsj[u], tcj[u], hj[u], qcj[u], clwj[u], epj[u], ph[u], ph[u + 1]);
    }
  }
  row.finish(Sx);
}

// PHASE 1: up to the early exits (:79-~420: sounding, lifting condensation level, first TLIFT); returns whether the mixing
// computation is needed.  PHASE 2: the rest, with the matrices.  PHASE 3: the rest up to the normalised M (the part that is
// sequential along the column); returns whether the column goes on.  (TH, computed and never used by the scheme, is left out.)
template <typename H, int PHASE>
__device__ bool convect(const Scr<H> &Sx, int nl, H delt, CvState<H> &st, int &nconvtop_) {
#pragma clang fp contract(off)
  const H elcrit = HK(.0011), tlcrit = HK(-55.0), entp = HK(1.5), sigd = HK(0.05), sigs = HK(0.12), omtrain = HK(50.0), omtsnow = HK(5.5);
  const H coeffr = HK(1.0), coeffs = HK(0.8), beta = HK(10.0), dtmax = HK(0.9), alpha = HK(0.025), damp = HK(0.1);
  const H cpd = HK(1005.7), cpv = HK(1870.0), cl = HK(2500.0), rv = HK(461.5), rd = HK(287.04), lv0 = HK(2.501e6), g = HK(9.81), rowl = HK(1000.0);
  const H cpvmcl = cl - cpv, eps0 = rd / rv, epsi = HK(1.) / eps0, ginv = HK(1.0) / g, epsilon = HK(1.e-20);
  const int minorig = 1;
  int iflag = st.iflag, i, icb = 0, ihmin, inb, inb1, j, jtt, k, nk = 0;
  H cbmf = st.cbmf, precip, wd, tprime, qprime;
  H ad, afac, ahmax, ahmin, alt, altem, am, amp1, anum, asij, awat, b6, bf2, bsum, by, byp, c6, cape, capem, cbmfold, chi, coeff;
  H cpinv, cwat, damps, dbo, dbosum, defrac, dei, delm, delp, delt0, delti, denom, dhdp, dpinv, dtma, dtmin, dtpbl, elacrit, ents;
  H epmax, fac, fqold, frac, ftold, plcl, qp1, qsm, qstm, qti, rat, rdcp, revap, rh, scrit, sigt, sjmax, sjmin, smid, smin, stemp, tca;
  H tvaplcl, tvpplcl, tvx, tvy, wdtrain;

  delti = HK(1.0) / delt;
  precip = HK(0.0); wd = HK(0.0); tprime = HK(0.0); qprime = HK(0.0);
  plcl = HK(0.);
  if (PHASE == 1) {
  for (i = 1; i <= nl + 1; i++) {
    VV(fdown, i) = HK(0.0); VV(sub, i) = HK(0.0); VV(fup, i) = HK(0.0); VV(m, i) = HK(0.0);
  }
  iflag = 0;
#define RETURN_ do { st.iflag = iflag; st.cbmf = cbmf; (void)precip; (void)wd; (void)tprime; (void)qprime; return PHASE == 2; } while (0)
  VV(gz, 1) = HK(0.0);
  VV(cpn, 1) = cpd * (HK(1.) - VV(qconv, 1)) + VV(qconv, 1) * cpv;
  VV(h, 1) = VV(tconv, 1) * VV(cpn, 1);
  VV(lv, 1) = lv0 - cpvmcl * (VV(tconv, 1) - HK(273.15));
  VV(hm, 1) = VV(lv, 1) * VV(qconv, 1);
  VV(tv, 1) = VV(tconv, 1) * (HK(1.) + VV(qconv, 1) * epsi - VV(qconv, 1));
  ahmin = HK(1.0e12);
  ihmin = nl;
  {
    // (the running values of level i - 1 stay in registers, the inputs of eight levels are requested together: as written the
    // loop reads back what it stored one iteration earlier, a memory round trip per level)
    const H t_1 = VV(tconv, 1);
    H t_p = t_1, q_p = VV(qconv, 1), p_p = VV(pconv_hpa, 1), gz_p = HK(0.0), hm_p = VV(lv, 1) * VV(qconv, 1);
    for (int i0 = 2; i0 <= nl + 1; i0 += 8) {
      H tb[8], qb[8], pb[8], phb[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int ii = I_MIN(i0 + u, nl + 1);
        tb[u] = VV(tconv, ii); qb[u] = VV(qconv, ii); pb[u] = VV(pconv_hpa, ii); phb[u] = VV(phconv_hpa, ii);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        i = i0 + u;
        if (i > nl + 1) break;
        const H t_i = tb[u], q_i = qb[u];
        tvx = t_i * (HK(1.) + q_i * epsi - q_i);
        tvy = t_p * (HK(1.) + q_p * epsi - q_p);
        const H gz_i = gz_p + HK(0.5) * rd * (tvx + tvy) * (p_p - pb[u]) / phb[u];
        const H cpn_i = cpd * (HK(1.) - q_i) + cpv * q_i;
        const H lv_i = lv0 - cpvmcl * (t_i - HK(273.15));
        const H hm_i = (cpd * (HK(1.) - q_i) + cl * q_i) * (t_i - t_1) + lv_i * q_i + gz_i;
        VV(gz, i) = gz_i;
        VV(cpn, i) = cpn_i;
        VV(h, i) = t_i * cpn_i + gz_i;
        VV(lv, i) = lv_i;
        VV(hm, i) = hm_i;
        VV(tv, i) = t_i * (HK(1.) + q_i * epsi - q_i);
        if (i >= minorig && hm_i < ahmin && hm_i < hm_p) { ahmin = hm_i; ihmin = i; }
        t_p = t_i; q_p = q_i; p_p = pb[u]; gz_p = gz_i; hm_p = hm_i;
      }
    }
  }
  ihmin = I_MIN(ihmin, nl - 1);
  ahmax = HK(0.0);
  nk = minorig;
  for (i = minorig; i <= ihmin; i++)
    if (VV(hm, i) > ahmax) { nk = i; ahmax = VV(hm, i); }
  if (VV(tconv, nk) < HK(250.0) || VV(qconv, nk) <= HK(0.0) || ihmin == (nl - 1)) { iflag = 0; cbmf = HK(0.0); RETURN_; }
  rh = VV(qconv, nk) / VV(qsconv, nk);
  chi = VV(tconv, nk) / (HK(1669.0) - HK(122.0) * rh - VV(tconv, nk));
  plcl = VV(pconv_hpa, nk) * M<H>::pow(rh, chi);
  if (plcl < HK(200.0) || plcl >= HK(2000.0)) { iflag = 2; cbmf = HK(0.0); RETURN_; }
  icb = nl - 1;
  for (i = nk + 1; i <= nl; i++)
    if (VV(pconv_hpa, i) < plcl) icb = I_MIN(icb, i);
  if (icb >= (nl - 1)) { iflag = 3; cbmf = HK(0.0); RETURN_; }
  tlift<H>(Sx, icb, nk, nl, 1);
  for (i = nk; i <= icb; i++) VV(tvp, i) = VV(tvp, i) - VV(tp, i) * VV(qconv, nk);
  if (cbmf == HK(0.0) && VV(tvp, icb) <= (VV(tv, icb) - dtmax)) { iflag = 0; RETURN_; }
  if (iflag != 4) iflag = 1;
  st.nk = nk; st.icb = icb; st.plcl = plcl; st.iflag = iflag; st.cbmf = cbmf;
  return true;
  }   // PHASE 1
  nk = st.nk; icb = st.icb; plcl = st.plcl;
  tlift<H>(Sx, icb, nk, nl, 2);
  for (i = 1; i <= nk; i++) { VV(ep, i) = HK(0.0); VV(sigp, i) = sigs; }
  // (in the level loops from here to the normalised M the inputs of eight levels are requested together and values the
  // reference stores and reads back at once stay in registers: as written every level costs one to three memory round trips)
  epmax = HK(0.999);
  for (int i0 = nk + 1; i0 <= nl; i0 += 8) {
    H tpv[8], clv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int ii = I_MIN(i0 + u, nl); tpv[u] = VV(tp, ii); clv[u] = VV(clw, ii); }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      i = i0 + u;
      if (i > nl) break;
      tca = tpv[u] - HK(273.15);
      if (tca >= HK(0.0)) elacrit = elcrit; else elacrit = elcrit * (HK(1.0) - tca / tlcrit);
      elacrit = R_MAX(elacrit, HK(0.0));
      H ep_i = epmax * (HK(1.0) - elacrit / R_MAX(clv[u], HK(1.0e-8)));
      ep_i = R_MAX(ep_i, HK(0.0));
      ep_i = R_MIN(ep_i, epmax);
      VV(ep, i) = ep_i;
      VV(sigp, i) = sigs;
    }
  }
  {
    const H q_nk = VV(qconv, nk);
    H tvp_nl = HK(0.);
    for (int i0 = icb + 1; i0 <= nl; i0 += 8) {
      H a8[8], b8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int ii = I_MIN(i0 + u, nl); a8[u] = VV(tvp, ii); b8[u] = VV(tp, ii); }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        i = i0 + u;
        if (i > nl) break;
        tvp_nl = a8[u] - b8[u] * q_nk;
        VV(tvp, i) = tvp_nl;
      }
    }
    if (icb + 1 > nl) tvp_nl = VV(tvp, nl);
    VV(tvp, nl + 1) = tvp_nl - (VV(gz, nl + 1) - VV(gz, nl)) / cpd;
  }
  for (int i0 = 1; i0 <= nl + 1; i0 += 8) {
    H h8[8];
#pragma unroll
    for (int u = 0; u < 8; u++) h8[u] = VV(h, I_MIN(i0 + u, nl + 1));
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (i0 + u <= nl + 1) { VV(hp, i0 + u) = h8[u]; VV(nent, i0 + u) = 0; }
  }
  cape = HK(0.0); capem = HK(0.0);
  inb = icb + 1; inb1 = inb;
  byp = HK(0.0);
  for (int i0 = icb + 1; i0 <= nl - 1; i0 += 8) {
    H tvpv[9], tvv[9], phv[10], ppv[9];                    // levels i0 .. i0 + 8 (+ 9 for PHCONV_HPA); nl + 1 is the last that exists
#pragma unroll
    for (int u = 0; u < 9; u++) { const int ii = I_MIN(i0 + u, nl); tvpv[u] = VV(tvp, ii); tvv[u] = VV(tv, ii); ppv[u] = VV(pconv_hpa, ii); }
#pragma unroll
    for (int u = 0; u < 10; u++) phv[u] = VV(phconv_hpa, I_MIN(i0 + u, nl + 1));
#pragma unroll
    for (int u = 0; u < 8; u++) {
      i = i0 + u;
      if (i > nl - 1) break;
      by = (tvpv[u] - tvv[u]) * (phv[u] - phv[u + 1]) / ppv[u];
      cape = cape + by;
      if (by >= HK(0.0)) inb1 = i + 1;
      if (cape > HK(0.0)) {
        inb = i + 1;
        byp = (tvpv[u + 1] - tvv[u + 1]) * (phv[u + 1] - phv[u + 2]) / ppv[u + 1];
        capem = cape;
      }
    }
  }
  inb = I_MAX(inb, inb1);
  st.inb = inb;
  cape = capem + byp;
  defrac = capem - cape;
  defrac = R_MAX(defrac, HK(0.001));
  frac = -cape / defrac;
  frac = R_MIN(frac, HK(1.0));
  frac = R_MAX(frac, HK(0.0));
  {
    const H h_nk = VV(h, nk);
    for (int i0 = icb; i0 <= inb; i0 += 8) {
      H a8[8], b8[8], c8[8], d8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int ii = I_MIN(i0 + u, inb); a8[u] = VV(lv, ii); b8[u] = VV(tconv, ii); c8[u] = VV(ep, ii); d8[u] = VV(clw, ii); }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (i0 + u <= inb) VV(hp, i0 + u) = h_nk + (a8[u] + (cpd - cpv) * b8[u]) * c8[u] * d8[u];
    }
  }
  dbosum = HK(0.0);
  tvpplcl = VV(tvp, icb - 1) - rd * VV(tvp, icb - 1) * (VV(pconv_hpa, icb - 1) - plcl) / (VV(cpn, icb - 1) * VV(pconv_hpa, icb - 1));
  tvaplcl = VV(tv, icb) + (VV(tvp, icb) - VV(tvp, icb + 1)) * (plcl - VV(pconv_hpa, icb)) / (VV(pconv_hpa, icb) - VV(pconv_hpa, icb + 1));
  dtpbl = HK(0.0);
  for (i = nk; i <= icb - 1; i++) dtpbl = dtpbl + (VV(tvp, i) - VV(tv, i)) * (VV(phconv_hpa, i) - VV(phconv_hpa, i + 1));
  dtpbl = dtpbl / (VV(phconv_hpa, nk) - VV(phconv_hpa, icb));
  dtmin = tvpplcl - tvaplcl + dtmax + dtpbl;
  dtma = dtmin;
  cbmfold = cbmf;
  delt0 = delt / HK(3.);
  damps = damp * delt / delt0;
  cbmf = (HK(1.) - damps) * cbmf + HK(0.1) * alpha * dtma;
  cbmf = R_MAX(cbmf, HK(0.0));
  if (cbmf == HK(0.0) && cbmfold == HK(0.0)) RETURN_;
  VV(m, icb) = HK(0.0);
  for (int i0 = icb + 1; i0 <= inb; i0 += 8) {
    H a8[8], b8[8], c8[8], d8[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      k = I_MIN(I_MIN(i0 + u, inb), inb1);
      a8[u] = VV(tv, k); b8[u] = VV(tvp, k); c8[u] = VV(phconv_hpa, k); d8[u] = VV(phconv_hpa, k + 1);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (i0 + u > inb) break;
      dbo = R_ABS(a8[u] - b8[u]) + entp * HK(0.02) * (c8[u] - d8[u]);
      dbosum = dbosum + dbo;
      VV(m, i0 + u) = cbmf * dbo;
    }
  }
  for (int i0 = icb + 1; i0 <= inb; i0 += 8) {
    H a8[8];
#pragma unroll
    for (int u = 0; u < 8; u++) a8[u] = VV(m, I_MIN(i0 + u, inb));
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (i0 + u <= inb) VV(m, i0 + u) = a8[u] / dbosum;
  }
  // FUP(1), :800-810 (needs M only; the reference has it after the two loop nests below)
  dpinv = HK(0.01) / (VV(phconv_hpa, 1) - VV(phconv_hpa, 2));
  am = HK(0.0);
  if (nk == 1)
    for (k = 2; k <= inb; k++) am = am + VV(m, k);
  VV(fup, 1) = am;
  if ((HK(2.) * g * dpinv * am) >= delti) iflag = 4;
  if (PHASE == 3) {                                     // the level-parallel path continues in k_conv_rows ...
    // two factors of the mixing computation that depend on the level j only (convect43c.f90:594, :612), once per column instead
    // of once per matrix entry: BF2(j) into TP, CWAT(j) into HM (both free from here on in this path)
    for (int i0 = icb; i0 <= inb; i0 += 8) {
      H a8[8], b8[8], c8[8], d8[8], e8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int ii = I_MIN(i0 + u, inb);
        a8[u] = VV(lv, ii); b8[u] = VV(qsconv, ii); c8[u] = VV(tconv, ii); d8[u] = VV(clw, ii); e8[u] = VV(ep, ii);
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (i0 + u <= inb) {
          VV(tp, i0 + u) = HK(1.) + a8[u] * a8[u] * b8[u] / (rv * c8[u] * c8[u] * cpd);
          VV(hm, i0 + u) = d8[u] * (HK(1.) - e8[u]);
        }
    }
    st.iflag = iflag; st.cbmf = cbmf;
    return true;
  }
  for (i = icb + 1; i <= inb; i++) sij_row<H>(Sx, nk, icb, inb, i);
  MM(sij, inb, inb) = HK(1.0);
  for (i = icb + 1; i <= inb; i++) norm_row<H>(Sx, nk, icb, inb, i);
  // (the precipitation / unsaturated-downdraught part, :686-790, and the tendencies FT, FQ with their entropy correction,
  // :800-900, feed PRECIP, WD, TPRIME, QPRIME, FT, FQ only: nothing of that reaches the particles -- left out, together
  // with the matrices QENT and ELIJ they alone read)
  // FUP(i) = [i >= nk] sum_{k>i} M(k) + sum_{k<=i} sum_{j>i} MENT(k,j) and FDOWN(i) = sum_{k<i} sum_{j>=i} MENT(j,k): the reference
  // forms them with two O(levels^3) loop nests per column.  Here: one descending sweep with the column sums S_k = sum_{j>=i}
  // MENT(j,k) and one ascending sweep with C_j = sum_{k<=i} MENT(k,j) -- O(levels^2), the additions in another order (the only
  // place where this kernel departs from the reference's order of operations; it moves FUP, FDOWN by rounding).
  // S lives in tp (last read when EP was formed), C in hm (last read in the first part).
  {
    const int r0 = icb + 1, c0 = icb;
    for (k = c0; k <= inb; k++) { VV(tp, k) = HK(0.); VV(hm, k) = HK(0.); }
    for (i = inb; i >= 2; i--) {
      if (i >= r0)
        for (k = c0; k <= inb; k += 16) {
          H a8[16], b8[16];
#pragma unroll
          for (int u = 0; u < 16; u++) { a8[u] = k + u <= inb ? VV(tp, k + u) : HK(0.); b8[u] = k + u <= inb ? MM(ment, i, k + u) : HK(0.); }
#pragma unroll
          for (int u = 0; u < 16; u++) if (k + u <= inb) VV(tp, k + u) = a8[u] + b8[u];
        }
      ad = HK(0.0);
      const int kend = I_MIN(i - 1, inb);
      for (k = c0; k <= kend; k += 16) {
        H t8[16];
#pragma unroll
        for (int u = 0; u < 16; u++) t8[u] = k + u <= kend ? VV(tp, k + u) : HK(0.);
#pragma unroll
        for (int u = 0; u < 16; u++) ad = ad + t8[u];
      }
      VV(fdown, i) = ad;
    }
    for (i = 2; i <= inb; i++) {
      dpinv = HK(0.01) / (VV(phconv_hpa, i) - VV(phconv_hpa, i + 1));
      amp1 = HK(0.0);
      if (i >= nk)
        for (k = i + 1; k <= inb + 1; k += 16) {
          H t8[16];
#pragma unroll
          for (int u = 0; u < 16; u++) t8[u] = k + u <= inb + 1 ? VV(m, k + u) : HK(0.);
#pragma unroll
          for (int u = 0; u < 16; u++) amp1 = amp1 + t8[u];
        }
      if (i >= r0)
        for (j = c0; j <= inb; j += 16) {
          H a8[16], b8[16];
#pragma unroll
          for (int u = 0; u < 16; u++) { a8[u] = j + u <= inb ? VV(hm, j + u) : HK(0.); b8[u] = j + u <= inb ? MM(ment, i, j + u) : HK(0.); }
#pragma unroll
          for (int u = 0; u < 16; u++) if (j + u <= inb) VV(hm, j + u) = a8[u] + b8[u];
        }
      for (j = I_MAX(i + 1, c0); j <= inb; j += 16) {
        H t8[16];
#pragma unroll
        for (int u = 0; u < 16; u++) t8[u] = j + u <= inb ? VV(hm, j + u) : HK(0.);
#pragma unroll
        for (int u = 0; u < 16; u++) amp1 = amp1 + t8[u];
      }
      VV(fup, i) = amp1;
      if ((HK(2.) * g * dpinv * amp1) >= delti) iflag = 4;
    }
    // FMASS(j,i) = [j == nk] M(i) + MENT(j,i) is not stored: calcmatrix scales it into fmassfrac on the fly (fmass_at below);
    // here only nconvtop, the largest index with a mass flux above epsilon, and SUB
    VV(sub, 1) = HK(0.);
    nconvtop_ = 1;
    for (i = 1; i <= inb + 1; i++) {
      if (VV(m, i) > epsilon) nconvtop_ = I_MAX(nconvtop_, I_MAX(i, nk));
      if (i >= c0 && i <= inb)
        for (j = r0; j <= inb; j += 16) {
          H t8[16];
#pragma unroll
          for (int u = 0; u < 16; u++) t8[u] = j + u <= inb ? MM(ment, j + u, i) : HK(0.);
#pragma unroll
          for (int u = 0; u < 16; u++) if (t8[u] > epsilon) nconvtop_ = I_MAX(nconvtop_, I_MAX(i, j + u));
        }
      if (i > 1) VV(sub, i) = VV(fup, i - 1) - VV(fdown, i);
    }
  }
  nconvtop_ = nconvtop_ + 1;
  RETURN_;
#undef RETURN_
}


// what one call needs of the model-level fields: both time slots, compact [level][jy][ix], per wind-field domain
// (0 = mother grid, l = nest l); columns are numbered through all domains: off + jy * nx + ix
constexpr int kConvMaxDom = 5;
template <typename H>
struct Dom {
  const H *ps[2], *tt2[2], *td2[2], *tth[2], *qvh[2];
  H *cb;                              // cbaseflux / cbasefluxn(:,:,l)
  int nx, ny, off;
  H xl, yl, xr, yr, xres, yres;       // nests: xln, yln, xrn, yrn, xresoln, yresoln
};
template <typename H>
struct Fields {
  Dom<H> dom[kConvMaxDom];
  int ndom;
  const H *akz, *bkz, *akm, *bkm;     // [nuvz], 0-based
  int nuvz, nconvlev;
  int m1, m2;                         // which physical slot is memind(1), memind(2)
  H dt1, dt2, dtt, delt, eps;         // eps = nxmax/3.e5 (convmix.f90:79)
  __device__ __forceinline__ int domain_of(int col) const {
    int d = 0;
    while (d + 1 < ndom && col >= dom[d + 1].off) d++;
    return d;
  }
};

// convmix.f90:92-135: the grid (innermost nest that contains the particle, tested with eps as for ECMWF input) and the
// column of every particle that is due
template <typename R, typename H>
__global__ void k_conv_mark(Fields<H> F, const double *__restrict__ xt, const double *__restrict__ yt, const int *__restrict__ itra1, long long n,
                            int itime, int *__restrict__ pcol, unsigned int *__restrict__ colflag) {
#pragma clang fp contract(off)
  long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  int col = -1;
  if (itra1[s] == itime) {
    const H x = (H)xt[s], y = (H)yt[s];                  // convmix.f90:100-101: into default reals
    int d = 0;
    for (int j = F.ndom - 1; j >= 1; j--)
      if (x > F.dom[j].xl + F.eps && x < F.dom[j].xr - F.eps && y > F.dom[j].yl + F.eps && y < F.dom[j].yr - F.eps) { d = j; break; }
    H xg = x, yg = y;
    if (d > 0) { xg = (x - F.dom[d].xl) * F.dom[d].xres; yg = (y - F.dom[d].yl) * F.dom[d].yres; }
    const int ix = (int)(xg < 0 ? xg - HK(0.5) : xg + HK(0.5)), jy = (int)(yg < 0 ? yg - HK(0.5) : yg + HK(0.5));   // nint
    if (ix >= 0 && ix < F.dom[d].nx && jy >= 0 && jy < F.dom[d].ny) { col = F.dom[d].off + jy * F.dom[d].nx + ix; colflag[col] = 1u; }
  }
  pcol[s] = col;
}

__global__ void k_conv_list(const unsigned int *__restrict__ colflag, const unsigned int *__restrict__ rank, int ncol, int *__restrict__ act) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ncol && colflag[c]) act[rank[c]] = c;
}

// Columns per wave of the kernels that walk a sounding with one lane per column (k_conv_column_a, k_conv_prelude).  Fewer
// than 64 (the other lanes leave at once) multiplies the waves in flight; measured with 16: column_a 1.15 -> 1.40 ms, prelude
// 0.85 -> 0.90 ms at 61760 / 29248 columns -- the kernels are bound by the number of memory instructions, not by exposed latency.
constexpr int kSerialLanes = 64;
__host__ inline unsigned int serial_grid(int ncolumns) { return (unsigned int)((ncolumns + kSerialLanes - 1) / kSerialLanes); }

// per-column scalars kept between the two column kernels
enum Cst { C_psconv, C_tt2conv, C_td2conv, C_cbmf, C_cbmfold, C_plcl, C_nk, C_icb, C_iflag, C_inb, C_COUNT };

// convmix.f90:149-170 + calcmatrix.f90:56-90 + CONVECT up to its early exits, for every column that holds particles
template <typename H>
__global__ void __launch_bounds__(64) k_conv_column_a(Fields<H> F, H *__restrict__ vbuf, H *__restrict__ cst, int nv, const int *__restrict__ act,
                                                      int nact, unsigned int *__restrict__ alive) {
#pragma clang fp contract(off)
  if ((int)threadIdx.x >= kSerialLanes) return;
  const int c = blockIdx.x * kSerialLanes + threadIdx.x;
  if (c >= nact) return;
  Scr<H> Sx{vbuf, nullptr, nact, c, nv, 0, 0};
  const Dom<H> &D = F.dom[F.domain_of(act[c])];
  const int col = act[c] - D.off;
  const size_t n2 = (size_t)D.nx * D.ny;
  const int nuvz = F.nuvz, nl = F.nconvlev;
  const H dt1 = F.dt1, dt2 = F.dt2, dtt = F.dtt;
  const H psconv = (D.ps[F.m1][col] * dt2 + D.ps[F.m2][col] * dt1) * dtt;
  cst[(size_t)C_psconv * nact + c] = psconv;
  cst[(size_t)C_tt2conv * nact + c] = (D.tt2[F.m1][col] * dt2 + D.tt2[F.m2][col] * dt1) * dtt;
  cst[(size_t)C_td2conv * nact + c] = (D.td2[F.m1][col] * dt2 + D.td2[F.m2][col] * dt1) * dtt;
  // convmix.f90:149-160 and calcmatrix.f90:56-90 level by level.  The reference fills its arrays in three loops; here one loop
  // carries the values in registers (a value stored to the scratch and read back in the next loop, or the next iteration,
  // costs a memory round trip each time) and the field values of eight levels are requested together.
  VV(phconv, 1) = psconv;
  H ph_lo = psconv;                                        // phconv(k)
  for (int k0 = 1; k0 <= nuvz - 1; k0 += 8) {
    H t1[8], t2[8], q1[8], q2[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const size_t o = (size_t)I_MIN(k0 + u, nuvz - 1) * n2 + col;
      t1[u] = D.tth[F.m1][o]; t2[u] = D.tth[F.m2][o]; q1[u] = D.qvh[F.m1][o]; q2[u] = D.qvh[F.m2][o];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int k = k0 + u;
      if (k > nuvz - 1) break;
      const H tc = (t1[u] * dt2 + t2[u] * dt1) * dtt;
      VV(tconv, k) = tc;
      VV(qconv, k) = (q1[u] * dt2 + q2[u] * dt1) * dtt;
      const H pc = (F.akz[k] + F.bkz[k] * psconv);           // pconv(k), phconv(k+1): level kuvz = k + 1 of the 0-based tables
      const H ph_hi = (F.akm[k] + F.bkm[k] * psconv);
      VV(pconv, k) = pc;
      VV(phconv, k + 1) = ph_hi;
      VV(dpr, k) = ph_lo - ph_hi;
      VV(qsconv, k) = cp::f_qvsat<H>(pc, tc);
      if (k <= nl + 1) {
        VV(pconv_hpa, k) = pc / HK(100.);
        VV(phconv_hpa, k) = ph_lo / HK(100.);
      }
      ph_lo = ph_hi;
    }
  }
  CvState<H> st;
  st.nk = 0; st.icb = 0; st.inb = 0; st.iflag = 0; st.plcl = HK(0.);
  st.cbmf = D.cb[col];
  cst[(size_t)C_cbmfold * nact + c] = st.cbmf;
  int dummy = 0;
  const bool go = convect<H, 1>(Sx, nl, F.delt, st, dummy);
  cst[(size_t)C_cbmf * nact + c] = st.cbmf;
  cst[(size_t)C_plcl * nact + c] = st.plcl;
  cst[(size_t)C_nk * nact + c] = (H)st.nk;
  cst[(size_t)C_icb * nact + c] = (H)st.icb;
  cst[(size_t)C_iflag * nact + c] = (H)st.iflag;
  alive[c] = go ? 1u : 0u;
}

__global__ void k_conv_survivors(const unsigned int *__restrict__ alive, const unsigned int *__restrict__ srank, int nact, int *__restrict__ surv) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nact && alive[c]) surv[srank[c]] = c;
}

// redist.f90:83-121: heights of the half levels (the reference computes them with the first particle of the column)
template <typename H>
__device__ void half_level_heights(const Scr<H> &Sx, int nconvtop, H psconv, H tt2conv, H td2conv) {
#pragma clang fp contract(off)
  const H konst = HK(287.05) / HK(9.81);
  H tvold = tt2conv * (HK(1.) + HK(0.378) * vt::ew<H>(td2conv) / psconv);
  H pold = psconv;
  VV(uvzlev, 1) = HK(0.);
  H pint = VV(phconv, 2);
  H tv1 = VV(tconv, 1) * (HK(1.) + HK(0.608) * VV(qconv, 1));
  H tv2 = VV(tconv, 2) * (HK(1.) + HK(0.608) * VV(qconv, 2));
  H tv = tv1 + (tv2 - tv1) * (VV(pconv, 1) - VV(phconv, 2)) / (VV(pconv, 1) - VV(pconv, 2));
  if (R_ABS(tv - tvold) > HK(0.2)) VV(uvzlev, 2) = VV(uvzlev, 1) + konst * M<H>::log(pold / pint) * (tv - tvold) / M<H>::log(tv / tvold);
  else VV(uvzlev, 2) = VV(uvzlev, 1) + konst * M<H>::log(pold / pint) * tv;
  tvold = tv; tv1 = tv2; pold = pint;
  for (int kz = 3; kz <= nconvtop + 1; kz++) {
    pint = VV(phconv, kz);
    tv2 = VV(tconv, kz) * (HK(1.) + HK(0.608) * VV(qconv, kz));
    tv = tv1 + (tv2 - tv1) * (VV(pconv, kz - 1) - VV(phconv, kz)) / (VV(pconv, kz - 1) - VV(pconv, kz));
    if (R_ABS(tv - tvold) > HK(0.2)) VV(uvzlev, kz) = VV(uvzlev, kz - 1) + konst * M<H>::log(pold / pint) * (tv - tvold) / M<H>::log(tv / tvold);
    else VV(uvzlev, kz) = VV(uvzlev, kz - 1) + konst * M<H>::log(pold / pint) * tv;
    tvold = tv; tv1 = tv2; pold = pint;
  }
}

// the mixing part of CONVECT, calcmatrix.f90:92-137 and the half-level heights of redist.f90:83-121 for the surviving columns
// surv[m0 .. m0+Bm)
template <typename H>
__global__ void __launch_bounds__(64) k_conv_column_b(Fields<H> F, H *__restrict__ vbuf, H *__restrict__ mbuf, H *__restrict__ cst, int nv, int nact,
                                                      const int *__restrict__ act, const int *__restrict__ surv, int m0, int Bm, int nsurv,
                                                      int *__restrict__ lconv_out, int *__restrict__ ntop_out) {
#pragma clang fp contract(off)
  const int cm = blockIdx.x * blockDim.x + threadIdx.x;
  if (cm >= Bm || m0 + cm >= nsurv) return;
  const int c = surv[m0 + cm];
  Scr<H> Sx{vbuf, mbuf, nact, c, nv, Bm, cm};
  const int nl = F.nconvlev;
  CvState<H> st;
  st.cbmf = cst[(size_t)C_cbmf * nact + c];
  st.plcl = cst[(size_t)C_plcl * nact + c];
  st.nk = (int)cst[(size_t)C_nk * nact + c];
  st.icb = (int)cst[(size_t)C_icb * nact + c];
  st.iflag = (int)cst[(size_t)C_iflag * nact + c];
  st.inb = 0;
  const H cbmfold = cst[(size_t)C_cbmfold * nact + c];
  const H psconv = cst[(size_t)C_psconv * nact + c], tt2conv = cst[(size_t)C_tt2conv * nact + c], td2conv = cst[(size_t)C_td2conv * nact + c];
  int nconvtop = 0, lconv = 0;
  convect<H, 2>(Sx, nl, F.delt, st, nconvtop);
  H cbmf = st.cbmf;
  if (st.iflag != 1 && st.iflag != 4) cbmf = cbmfold;
  else if (cbmf <= HK(0.) && cbmfold <= HK(0.)) cbmf = cbmfold;
  else {
    const H ga = HK(9.81);
    lconv = 1;
    for (int k = 1; k <= nconvtop; k++) {
      const H rlevmass = VV(dpr, k) / ga;
      H summe = HK(0.);
      for (int kk = 1; kk <= nconvtop; kk += 16) {
        H f8[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int q = kk + u;
          H v = HK(0.);
          if (q <= nconvtop) {
            if (k > st.icb && k <= st.inb && q >= st.icb && q <= st.inb) v = MM(ment, k, q);
            if (k == st.nk) v = VV(m, q) + v;                  // FMASS(nk,i) = M(i) + MENT(nk,i), convect43c.f90:925-930
          }
          f8[u] = F.delt * v;
        }
#pragma unroll
        for (int u = 0; u < 16; u++)
          if (kk + u <= nconvtop) { MM(fmass, k, kk + u) = f8[u]; summe = summe + f8[u]; }     // fmassfrac(k,kk)
      }
      MM(fmass, k, k) = MM(fmass, k, k) + rlevmass - summe;
    }
    half_level_heights<H>(Sx, nconvtop, psconv, tt2conv, td2conv);
  }
  {
    const Dom<H> &D = F.dom[F.domain_of(act[c])];
    D.cb[act[c] - D.off] = cbmf;
  }
  lconv_out[c] = lconv;
  ntop_out[c] = lconv ? nconvtop : 0;
}

// ---- the mixing computation with a lane per (column, level) --------------------------------------------------------------
// k_conv_column_b gives every column one lane and is bound by the latency chain of the longest column (a few hundred waves
// on 1024 SIMDs).  Past the normalised M everything is a row- or column-wise pass over the matrices with no dependence
// between rows (columns), so these kernels take blockIdx.y + 1 as the level and keep the column on threadIdx.x (the
// interleaved scratch stays coalesced: the lanes of a wave share the level).  Every sum is formed in the order of the
// one-lane kernel: the two paths agree bit for bit (tests/test_convection.py).
//   k_conv_prelude  (column)        CONVECT from the second TLIFT to the normalised M, FUP(1)
//   k_conv_rows     (column, i)     SIJ / MENT row i and its normalisation in one sweep (mix_row)
//   k_conv_cols     (column, k)     running sums of column k of MENT: C_k(i) = sum_{r<=i} MENT(r,k) into FMASS(i,k), i < k;
//                                   S_k(i) = sum_{r>=i} MENT(r,k) into SIJ(i,k), i > k; the column's part of nconvtop
//   k_conv_flux     (column, i)     FDOWN(i) = sum_{k<i} S_k(i), FUP(i) = [i>=nk] sum_{k>i} M(k) + sum_{j>i} C_j(i), iflag 4
//   k_conv_matrix   (column, k)     row k of fmassfrac, SUB(k); the lane of level 1 also: cbaseflux, lconv, half-level heights
#define CONV_LANE_                                                           \
  const int cm = blockIdx.x * blockDim.x + threadIdx.x;                     \
  if (cm >= Bm || m0 + cm >= nsurv) return;                                 \
  const int c = surv[m0 + cm];                                              \
  Scr<H> Sx{vbuf, mbuf, nact, c, nv, Bm, cm};
// The (group of 64 columns, level) of a block of the level kernels.  All levels of a group re-read the group's vectors (360 KB
// at 100 levels), so they should run at the same time on the same XCD (its L2 holds 4 MB): consecutive workgroup ids go
// round the 8 XCDs, id % 8 picks the XCD, and within an XCD the levels of one group are consecutive.  (With the group as
// the fast index the resident blocks spanned all groups -- 160 MB of vectors -- and the kernel ran at the fabric's bandwidth.)
#define CONV_LEVEL_LANE_(nlev)                                               \
  const int q_ = blockIdx.x >> 3;                                           \
  const int cm = (int)(((blockIdx.x & 7) + 8 * (q_ / (nlev))) * kGroup + threadIdx.x); \
  const int lev = q_ % (nlev) + 1;                                          \
  if (cm >= Bm || m0 + cm >= nsurv) return;                                 \
  const int c = surv[m0 + cm];                                              \
  Scr<H> Sx{vbuf, mbuf, nact, c, nv, Bm, cm};
__host__ inline unsigned int level_grid(int ncolumns, int nlev) { return (unsigned int)(((ncolumns + kGroup - 1) / kGroup + 7) / 8 * 8 * nlev); }

template <typename H>
__global__ void __launch_bounds__(64) k_conv_prelude(Fields<H> F, H *__restrict__ vbuf, H *__restrict__ mbuf, H *__restrict__ cst, int nv, int nact,
                                                     const int *__restrict__ surv, int m0, int Bm, int nsurv, int *__restrict__ cflag,
                                                     int *__restrict__ ntop_raw) {
#pragma clang fp contract(off)
  if ((int)threadIdx.x >= kSerialLanes) return;
  const int cm = blockIdx.x * kSerialLanes + threadIdx.x;
  if (cm >= Bm || m0 + cm >= nsurv) return;
  const int c = surv[m0 + cm];
  Scr<H> Sx{vbuf, mbuf, nact, c, nv, Bm, cm};
  CvState<H> st;
  st.cbmf = cst[(size_t)C_cbmf * nact + c];
  st.plcl = cst[(size_t)C_plcl * nact + c];
  st.nk = (int)cst[(size_t)C_nk * nact + c];
  st.icb = (int)cst[(size_t)C_icb * nact + c];
  st.iflag = (int)cst[(size_t)C_iflag * nact + c];
  st.inb = 0;
  int dummy = 0;
  const bool go = convect<H, 3>(Sx, F.nconvlev, F.delt, st, dummy);
  cst[(size_t)C_cbmf * nact + c] = st.cbmf;
  cst[(size_t)C_inb * nact + c] = go ? (H)st.inb : HK(0.);     // inb 0: no level passes the range tests of the kernels below
  cflag[c] = st.iflag;
  ntop_raw[c] = 1;
}

template <typename H>
__global__ void __launch_bounds__(64) k_conv_rows(H *__restrict__ vbuf, H *__restrict__ mbuf, const H *__restrict__ cst, int nv, int nact,
                                                  const int *__restrict__ surv, int m0, int Bm, int nsurv, int nlev) {
#pragma clang fp contract(off)
  CONV_LEVEL_LANE_(nlev)
  const int i = lev;
  const int icb = (int)cst[(size_t)C_icb * nact + c], inb = (int)cst[(size_t)C_inb * nact + c];
  if (i < icb + 1 || i > inb) return;
  const int nk = (int)cst[(size_t)C_nk * nact + c];
  mix_row<H>(Sx, nk, icb, inb, i);
}

// k_conv_rows with the operands shared: the rows of a group of columns all walk the same seven vectors (and PHCONV_HPA) of
// that group, level by level.  A block holds kRowsPerBlock rows (waves) of one group; each batch of kJ levels is fetched once
// per block -- every wave a share of the 65 segments -- and parked in LDS [item][lane], from where every row reads its own
// column's values (conflict-free: consecutive lanes, consecutive words).  The next batch is in flight while this one is
// computed.  Same MixRow steps as mix_row: bit-identical.
constexpr int kRowsPerBlock = 8;      // (4 rows: 3.56 ms, 16 rows: 3.10 ms, 8 rows: 2.87 ms at 361x181x138)
constexpr int kStageItems = 7 * kJ + kJ + 1;
template <typename H>
__global__ void __launch_bounds__(64 * kRowsPerBlock) k_conv_rows_lds(H *__restrict__ vbuf, H *__restrict__ mbuf, const H *__restrict__ cst, int nv,
                                                                      int nact, const int *__restrict__ surv, int m0, int Bm, int nsurv, int nlev) {
#pragma clang fp contract(off)
  __shared__ H stage[kStageItems][kGroup];
  const int nrb = (nlev + kRowsPerBlock - 1) / kRowsPerBlock;
  const int q_ = blockIdx.x >> 3;
  const int lane = threadIdx.x, w = threadIdx.y;
  const int cm = (int)(((blockIdx.x & 7) + 8 * (q_ / nrb)) * kGroup + lane);
  const int row0 = (q_ % nrb) * kRowsPerBlock + 1;           // the block's rows: row0 .. row0 + kRowsPerBlock - 1
  const int i = row0 + w;
  const bool valid = cm < Bm && m0 + cm < nsurv;
  const int c = valid ? surv[m0 + cm] : 0;
  Scr<H> Sx{vbuf, mbuf, nact, c, nv, Bm, valid ? cm : 0};
  int icb = 0, inb = 0, nk = 1;
  if (valid) { icb = (int)cst[(size_t)C_icb * nact + c]; inb = (int)cst[(size_t)C_inb * nact + c]; nk = (int)cst[(size_t)C_nk * nact + c]; }
  // a column takes part when one of the block's rows lies in icb+1 .. inb; the same in every wave of the block
  const bool col_on = valid && row0 <= inb && row0 + kRowsPerBlock - 1 >= icb + 1;
  int nb = col_on ? (inb + 1 - icb) / kJ + 1 : 0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(nb, o); nb = I_MAX(nb, other); }   // (the macro would shuffle twice, once under a condition)
  if (nb == 0) return;                                       // (uniform over the block)
  const bool row_on = col_on && i >= icb + 1 && i <= inb;
  MixRow<H> row;
  if (row_on) row.init(Sx, nk, icb, inb, i);
  constexpr int kShare = (kStageItems + kRowsPerBlock - 1) / kRowsPerBlock;
  const int vec_of[7] = {V_lv, V_qsconv, V_tconv, V_h, V_qconv, V_tp, V_hm};         // TP, HM: BF2(j), CWAT(j) from the prelude
  H pre[kShare];
  auto fetch = [&](int b) {
    const int j = icb + b * kJ;
#pragma unroll
    for (int q = 0; q < kShare; q++) {
      const int t = w + q * kRowsPerBlock;
      H v = HK(0.);
      if (col_on && t < kStageItems) {
        if (t < 7 * kJ) v = Sx.vb[((size_t)vec_of[t / kJ] * nv + (size_t)I_MIN(j + t % kJ, inb)) * kGroup];
        else v = VV(phconv_hpa, I_MIN(j - 1 + (t - 7 * kJ), inb + 1));
      }
      pre[q] = v;
    }
  };
  fetch(0);
  for (int b = 0; b < nb; b++) {
    __syncthreads();                                         // the previous batch has been consumed
#pragma unroll
    for (int q = 0; q < kShare; q++) {
      const int t = w + q * kRowsPerBlock;
      if (t < kStageItems) stage[t][lane] = pre[q];
    }
    __syncthreads();
    if (b + 1 < nb) fetch(b + 1);
    const int j = icb + b * kJ;
    if (row_on && j <= inb + 1) {
#pragma unroll
      for (int u = 0; u < kJ; u++) {
        if (j + u > inb + 1) break;
        row.step(Sx, j + u, stage[0 * kJ + u][lane], stage[1 * kJ + u][lane], stage[2 * kJ + u][lane], stage[3 * kJ + u][lane],
                 stage[4 * kJ + u][lane], stage[5 * kJ + u][lane], stage[6 * kJ + u][lane], stage[7 * kJ + u][lane], stage[7 * kJ + u + 1][lane]);
      }
    }
  }
  if (row_on) row.finish(Sx);
}

template <typename H>
__global__ void __launch_bounds__(64) k_conv_cols(H *__restrict__ vbuf, H *__restrict__ mbuf, const H *__restrict__ cst, int nv, int nact,
                                                  const int *__restrict__ surv, int m0, int Bm, int nsurv, int nlev, int *__restrict__ ntop_raw) {
#pragma clang fp contract(off)
  CONV_LEVEL_LANE_(nlev)
  const int k = lev;
  const int icb = (int)cst[(size_t)C_icb * nact + c], inb = (int)cst[(size_t)C_inb * nact + c];
  if (k < icb || k > inb) return;
  const int nk = (int)cst[(size_t)C_nk * nact + c];
  const int r0 = icb + 1;
  const H epsilon = HK(1.e-20);
  int top = 1;
  if (VV(m, k) > epsilon) top = I_MAX(k, nk);
  // (rows r0..k upwards, inb..k+1 downwards: every entry is read once; kC rows requested together)
  constexpr int kC = 32;
  H run = HK(0.);
  const int kup = I_MIN(k, inb);
  for (int i = r0; i <= kup; i += kC) {                  // upwards: C_k(i), needed for i < k
    H b[kC];
#pragma unroll
    for (int u = 0; u < kC; u++) b[u] = i + u <= kup ? MM(ment, i + u, k) : HK(0.);
#pragma unroll
    for (int u = 0; u < kC; u++)
      if (i + u <= kup) {
        if (b[u] > epsilon) top = I_MAX(top, k);        // max(k, i + u) = k
        run = run + b[u];
        if (i + u < k) MM(fmass, i + u, k) = run;
      }
  }
  run = HK(0.);
  const int rdn = I_MAX(r0, k + 1);
  for (int i = inb; i >= rdn; i -= kC) {                 // downwards: S_k(i), needed for i > k
    H b[kC];
#pragma unroll
    for (int u = 0; u < kC; u++) b[u] = i - u >= rdn ? MM(ment, i - u, k) : HK(0.);
#pragma unroll
    for (int u = 0; u < kC; u++)
      if (i - u >= rdn) {
        if (b[u] > epsilon) top = I_MAX(top, i - u);    // max(k, i - u) = i - u
        run = run + b[u];
        MM(sij, i - u, k) = run;
      }
  }
  if (top > 1) atomicMax(&ntop_raw[c], top);
}

template <typename H>
__global__ void __launch_bounds__(64) k_conv_flux(Fields<H> F, H *__restrict__ vbuf, H *__restrict__ mbuf, const H *__restrict__ cst, int nv, int nact,
                                                  const int *__restrict__ surv, int m0, int Bm, int nsurv, int nlev, int *__restrict__ cflag) {
#pragma clang fp contract(off)
  CONV_LEVEL_LANE_(nlev)
  const int i = lev;
  const int icb = (int)cst[(size_t)C_icb * nact + c], inb = (int)cst[(size_t)C_inb * nact + c];
  if (i < 2 || i > inb) return;
  const int nk = (int)cst[(size_t)C_nk * nact + c];
  const int r0 = icb + 1, c0 = icb;
  const H g = HK(9.81), delti = HK(1.0) / F.delt;
  H ad = HK(0.0);
  if (i >= r0) {
    const int kend = I_MIN(i - 1, inb);
    for (int k = c0; k <= kend; k += 16) {
      H t8[16];
#pragma unroll
      for (int u = 0; u < 16; u++) t8[u] = k + u <= kend ? MM(sij, i, k + u) : HK(0.);
#pragma unroll
      for (int u = 0; u < 16; u++) ad = ad + t8[u];
    }
  }
  VV(fdown, i) = ad;
  const H dpinv = HK(0.01) / (VV(phconv_hpa, i) - VV(phconv_hpa, i + 1));
  H amp1 = HK(0.0);
  if (i >= nk)
    for (int k = i + 1; k <= inb + 1; k += 16) {
      H t8[16];
#pragma unroll
      for (int u = 0; u < 16; u++) t8[u] = k + u <= inb + 1 ? VV(m, k + u) : HK(0.);
#pragma unroll
      for (int u = 0; u < 16; u++) amp1 = amp1 + t8[u];
    }
  if (i >= r0)
    for (int j = i + 1; j <= inb; j += 16) {
      H t8[16];
#pragma unroll
      for (int u = 0; u < 16; u++) t8[u] = j + u <= inb ? MM(fmass, i, j + u) : HK(0.);
#pragma unroll
      for (int u = 0; u < 16; u++) amp1 = amp1 + t8[u];
    }
  VV(fup, i) = amp1;
  if ((HK(2.) * g * dpinv * amp1) >= delti) atomicMax(&cflag[c], 4);
}

template <typename H>
__global__ void __launch_bounds__(64) k_conv_matrix(Fields<H> F, H *__restrict__ vbuf, H *__restrict__ mbuf, const H *__restrict__ cst, int nv, int nact,
                                                    const int *__restrict__ act, const int *__restrict__ surv, int m0, int Bm, int nsurv, int nlev,
                                                    const int *__restrict__ cflag, const int *__restrict__ ntop_raw,
                                                    int *__restrict__ lconv_out, int *__restrict__ ntop_out) {
#pragma clang fp contract(off)
  CONV_LEVEL_LANE_(nlev)
  const int k = lev;
  const int iflag = cflag[c];
  const H cbmfold = cst[(size_t)C_cbmfold * nact + c];
  H cbmf = cst[(size_t)C_cbmf * nact + c];
  int lconv = 0;
  if (iflag != 1 && iflag != 4) cbmf = cbmfold;
  else if (cbmf <= HK(0.) && cbmfold <= HK(0.)) cbmf = cbmfold;
  else lconv = 1;
  const int nconvtop = ntop_raw[c] + 1;
  if (lconv) {
    const int icb = (int)cst[(size_t)C_icb * nact + c], inb = (int)cst[(size_t)C_inb * nact + c], nk = (int)cst[(size_t)C_nk * nact + c];
    const H ga = HK(9.81);
    if (k <= inb + 1) VV(sub, k) = k > 1 ? VV(fup, k - 1) - VV(fdown, k) : HK(0.);
    if (k <= nconvtop) {
      const H rlevmass = VV(dpr, k) / ga;
      H summe = HK(0.);
      for (int kk = 1; kk <= nconvtop; kk += 16) {
        H f8[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int q = kk + u;
          H v = HK(0.);
          if (q <= nconvtop) {
            if (k > icb && k <= inb && q >= icb && q <= inb) v = MM(ment, k, q);
            if (k == nk) v = VV(m, q) + v;                  // FMASS(nk,i) = M(i) + MENT(nk,i), convect43c.f90:925-930
          }
          f8[u] = F.delt * v;
        }
#pragma unroll
        for (int u = 0; u < 16; u++)
          if (kk + u <= nconvtop) { MM(fmass, k, kk + u) = f8[u]; summe = summe + f8[u]; }     // fmassfrac(k,kk)
      }
      MM(fmass, k, k) = MM(fmass, k, k) + rlevmass - summe;
    }
  }
  if (k != 1) return;
  if (lconv) half_level_heights<H>(Sx, nconvtop, cst[(size_t)C_psconv * nact + c], cst[(size_t)C_tt2conv * nact + c], cst[(size_t)C_td2conv * nact + c]);
  {
    const Dom<H> &D = F.dom[F.domain_of(act[c])];
    D.cb[act[c] - D.off] = cbmf;
  }
  lconv_out[c] = lconv;
  ntop_out[c] = lconv ? nconvtop : 0;
}
// k_conv_matrix for forward runs, with fmassfrac stored walk-major (Scr::fm_walk): k_conv_redist walks along row levold, and in
// the interleaved layout every entry of a row is another cache line (the kernel moved 1.5 KB per particle, 85 % of the HBM
// peak).  A lane still forms row k of its column; sixteen entries of each of the 64 columns are parked in LDS and written
// out as 128-byte runs.  The diagonal entry, which takes the row sum, is written by its owner at the end.
template <typename H>
__global__ void __launch_bounds__(64) k_conv_matrix_walk(Fields<H> F, H *__restrict__ vbuf, H *__restrict__ mbuf, const H *__restrict__ cst, int nv, int nact,
                                                         const int *__restrict__ act, const int *__restrict__ surv, int m0, int Bm, int nsurv, int nlev,
                                                         const int *__restrict__ cflag, const int *__restrict__ ntop_raw,
                                                         int *__restrict__ lconv_out, int *__restrict__ ntop_out) {
#pragma clang fp contract(off)
  constexpr int kWQ = 32, kWQs = 5;                      // entries of a row per tile (2^kWQs)
  __shared__ H tile[kGroup][kWQ + 1];
  __shared__ int s_top[kGroup];
  const int q_ = blockIdx.x >> 3;
  const int lane = threadIdx.x;
  const int cm = (int)(((blockIdx.x & 7) + 8 * (q_ / nlev)) * kGroup + lane);
  const int k = q_ % nlev + 1;
  const bool valid = cm < Bm && m0 + cm < nsurv;
  if (!__syncthreads_or(valid)) return;
  const int c = valid ? surv[m0 + cm] : 0;
  Scr<H> Sx{vbuf, mbuf, nact, c, nv, Bm, valid ? cm : (int)(((blockIdx.x & 7) + 8 * (q_ / nlev)) * kGroup)};
  int lconv = 0, nconvtop = 0, icb = 0, inb = 0, nk = 0;
  H cbmf = HK(0.);
  if (valid) {
    const int iflag = cflag[c];
    const H cbmfold = cst[(size_t)C_cbmfold * nact + c];
    cbmf = cst[(size_t)C_cbmf * nact + c];
    if (iflag != 1 && iflag != 4) cbmf = cbmfold;
    else if (cbmf <= HK(0.) && cbmfold <= HK(0.)) cbmf = cbmfold;
    else lconv = 1;
    nconvtop = ntop_raw[c] + 1;
    icb = (int)cst[(size_t)C_icb * nact + c]; inb = (int)cst[(size_t)C_inb * nact + c]; nk = (int)cst[(size_t)C_nk * nact + c];
  }
  const bool row_on = lconv && k <= nconvtop;
  s_top[lane] = row_on ? nconvtop : 0;
  int top_max = row_on ? nconvtop : 0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(top_max, o); top_max = I_MAX(top_max, other); }
  const H ga = HK(9.81);
  if (lconv && k <= inb + 1) VV(sub, k) = k > 1 ? VV(fup, k - 1) - VV(fdown, k) : HK(0.);
  H summe = HK(0.), fkk = HK(0.);
  H *const blk = Sx.fm_walk_group();
  const size_t nvnv = (size_t)nv * nv;
  for (int kk = 1; kk <= top_max; kk += kWQ) {
    H f8[kWQ];
#pragma unroll
    for (int u = 0; u < kWQ; u++) {
      const int q = kk + u;
      H v = HK(0.);
      if (row_on && q <= nconvtop) {
        if (k > icb && k <= inb && q >= icb && q <= inb) v = MM(ment, k, q);
        if (k == nk) v = VV(m, q) + v;                  // FMASS(nk,i) = M(i) + MENT(nk,i), convect43c.f90:925-930
      }
      f8[u] = F.delt * v;
    }
    __syncthreads();                                     // the previous tile has been written out
#pragma unroll
    for (int u = 0; u < kWQ; u++) {
      tile[lane][u] = f8[u];
      if (row_on && kk + u <= nconvtop) { summe = summe + f8[u]; if (kk + u == k) fkk = f8[u]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kWQ; r++) {
      const int e = r * kGroup + lane, col = e >> kWQs, u = e & (kWQ - 1), q = kk + u;
      if (q <= s_top[col] && q != k) blk[(size_t)col * nvnv + (size_t)k * nv + q] = tile[col][u];     // fmassfrac(k,q) of column col
    }
  }
  if (row_on) blk[(size_t)lane * nvnv + (size_t)k * nv + k] = fkk + VV(dpr, k) / ga - summe;
  if (k != 1 || !valid) return;
  if (lconv) half_level_heights<H>(Sx, nconvtop, cst[(size_t)C_psconv * nact + c], cst[(size_t)C_tt2conv * nact + c], cst[(size_t)C_td2conv * nact + c]);
  {
    const Dom<H> &D = F.dom[F.domain_of(act[c])];
    D.cb[act[c] - D.off] = cbmf;
  }
  lconv_out[c] = lconv;
  ntop_out[c] = lconv ? nconvtop : 0;
}
// The same for backward runs, where k_conv_redist walks down column levold: the block holds sixteen consecutive rows k of its
// 64 columns, so that the entries (k .. k+15, q) of a column -- consecutive in the transposed walk-major form
// [column][q][k] -- leave as one 128-byte run.  Four q at a time go through the LDS tile.
constexpr int kWalkRows = 16, kWalkQ = 4;
template <typename H>
__global__ void __launch_bounds__(64 * kWalkRows) k_conv_matrix_walk_t(Fields<H> F, H *__restrict__ vbuf, H *__restrict__ mbuf, const H *__restrict__ cst,
                                                                       int nv, int nact, const int *__restrict__ act, const int *__restrict__ surv, int m0,
                                                                       int Bm, int nsurv, int nlev, const int *__restrict__ cflag,
                                                                       const int *__restrict__ ntop_raw, int *__restrict__ lconv_out,
                                                                       int *__restrict__ ntop_out) {
#pragma clang fp contract(off)
  __shared__ H tile[kGroup][kWalkQ][kWalkRows + 1];
  __shared__ int s_top[kGroup];
  __shared__ int s_max;
  const int nrb = (nlev + kWalkRows - 1) / kWalkRows;
  const int q_ = blockIdx.x >> 3;
  const int lane = threadIdx.x, w = threadIdx.y;
  const int grp0 = (int)(((blockIdx.x & 7) + 8 * (q_ / nrb)) * kGroup);
  const int cm = grp0 + lane;
  const int k0 = (q_ % nrb) * kWalkRows + 1;
  const int k = k0 + w;
  const bool valid = cm < Bm && m0 + cm < nsurv;
  if (!__syncthreads_or(valid)) return;
  const int c = valid ? surv[m0 + cm] : 0;
  Scr<H> Sx{vbuf, mbuf, nact, c, nv, Bm, valid ? cm : grp0};
  int lconv = 0, nconvtop = 0, icb = 0, inb = 0, nk = 0;
  H cbmf = HK(0.);
  if (valid) {
    const int iflag = cflag[c];
    const H cbmfold = cst[(size_t)C_cbmfold * nact + c];
    cbmf = cst[(size_t)C_cbmf * nact + c];
    if (iflag != 1 && iflag != 4) cbmf = cbmfold;
    else if (cbmf <= HK(0.) && cbmfold <= HK(0.)) cbmf = cbmfold;
    else lconv = 1;
    nconvtop = ntop_raw[c] + 1;
    icb = (int)cst[(size_t)C_icb * nact + c]; inb = (int)cst[(size_t)C_inb * nact + c]; nk = (int)cst[(size_t)C_nk * nact + c];
  }
  const bool row_on = lconv && k <= nconvtop && k <= nlev;
  if (w == 0) {
    s_top[lane] = lconv && k0 <= nconvtop ? nconvtop : 0;       // columns with at least one of the block's rows
    int top_max = s_top[lane];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(top_max, o); top_max = I_MAX(top_max, other); }
    if (lane == 0) s_max = top_max;
  }
  __syncthreads();
  const int top_max = s_max;
  const H ga = HK(9.81);
  if (lconv && k <= nlev && k <= inb + 1) VV(sub, k) = k > 1 ? VV(fup, k - 1) - VV(fdown, k) : HK(0.);
  H summe = HK(0.), fkk = HK(0.);
  H *const blk = Sx.fm_walk_group();
  const size_t nvnv = (size_t)nv * nv;
  const int t = w * kGroup + lane;                                  // 0 .. 1023
  for (int kk = 1; kk <= top_max; kk += kWalkQ) {
    H f4[kWalkQ];
#pragma unroll
    for (int u = 0; u < kWalkQ; u++) {
      const int q = kk + u;
      H v = HK(0.);
      if (row_on && q <= nconvtop) {
        if (k > icb && k <= inb && q >= icb && q <= inb) v = MM(ment, k, q);
        if (k == nk) v = VV(m, q) + v;                  // FMASS(nk,i) = M(i) + MENT(nk,i), convect43c.f90:925-930
      }
      f4[u] = F.delt * v;
    }
    __syncthreads();                                     // the previous tile has been written out
#pragma unroll
    for (int u = 0; u < kWalkQ; u++) {
      tile[lane][u][w] = f4[u];
      if (row_on && kk + u <= nconvtop) { summe = summe + f4[u]; if (kk + u == k) fkk = f4[u]; }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < kWalkQ; p++) {
      const int pair = p * kGroup + (t >> 4), col = pair >> 2, u = pair & 3, kr = t & 15;
      const int q = kk + u, kx = k0 + kr;
      if (q <= s_top[col] && kx <= s_top[col] && q != kx) blk[(size_t)col * nvnv + (size_t)q * nv + kx] = tile[col][u][kr];   // fmassfrac(kx,q) of column col
    }
  }
  if (row_on) blk[(size_t)lane * nvnv + (size_t)k * nv + k] = fkk + VV(dpr, k) / ga - summe;
  if (k != 1 || !valid) return;
  if (lconv) half_level_heights<H>(Sx, nconvtop, cst[(size_t)C_psconv * nact + c], cst[(size_t)C_tt2conv * nact + c], cst[(size_t)C_td2conv * nact + c]);
  {
    const Dom<H> &D = F.dom[F.domain_of(act[c])];
    D.cb[act[c] - D.off] = cbmf;
  }
  lconv_out[c] = lconv;
  ntop_out[c] = lconv ? nconvtop : 0;
}
#undef CONV_LANE_
#undef CONV_LEVEL_LANE_

// redist.f90:124-236 for the particles of the surviving columns surv[m0 .. m0+Bm).  rn_in: the uniform number of each particle (serial
// stream replayed by the host) or NULL: drawn from the counter generator.  probe != 0: only report which particles would draw.
// colslot[column] = {active-column index, matrix slot in this batch, nconvtop, 1} for the convective columns of the batch, zero
// otherwise: a particle finds out with one gather whether it has anything to do
__global__ void k_conv_slots(const int *__restrict__ act, const unsigned int *__restrict__ alive, const unsigned int *__restrict__ srank,
                             const int *__restrict__ lconv, const int *__restrict__ ntop, int nact, int m0, int Bm, int4 *__restrict__ colslot) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nact || !alive[r]) return;
  const int ms = (int)srank[r] - m0;
  if (ms < 0 || ms >= Bm || !lconv[r]) return;
  colslot[act[r]] = make_int4(r, ms, ntop[r], 1);
}

template <typename R, typename H, typename RNGF>
__global__ void k_conv_redist(const int *__restrict__ pcol, const int4 *__restrict__ colslot, R *__restrict__ zt, long long n,
                              H *__restrict__ vbuf, H *__restrict__ mbuf, int nv, int nact, int Bm, int ldirect, int lsynctime, H height_nz,
                              RNGF rngf, unsigned char *__restrict__ draws, int probe, unsigned long long *__restrict__ nmoved, int walk) {
#pragma clang fp contract(off)
  long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int col = pcol[s];
  if (col < 0) return;
  const int4 slot = colslot[col];
  if (!slot.w) return;
  const int r = slot.x, ms = slot.y;
  Scr<H> Sx{vbuf, mbuf, nact, r, nv, Bm, ms};
  const int nconvtop = slot.z;
  const H r_air = HK(287.05), ga = HK(9.81);
  H ztold = (H)zt[s], znew = ztold;
  bool touched = false;                                // the height is written back only where the routine assigns it
  // redist.f90:124-131 takes the first kz in 2..nconvtop with uvzlev(kz) >= ztold by a linear scan.  The half-level heights
  // increase strictly with kz (every step adds konst * log(pold / pint) * tv > 0: pressure falls, temperatures are positive),
  // so a bisection finds the same kz with seven gathers instead of up to nconvtop (every level of the interleaved scratch is
  // another cache line; measured: 3 % of this kernel's HBM traffic, 4 % of its time).
  int levold = 0;
  if (nconvtop >= 2 && VV(uvzlev, nconvtop) >= ztold) {
    int lo = 2, hi = nconvtop;                           // invariant: uvzlev(hi) >= ztold, uvzlev(kz) < ztold for 2 <= kz < lo
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (VV(uvzlev, mid) >= ztold) hi = mid; else lo = mid + 1;
    }
    levold = lo - 1;
  }
  if (levold > 0) {
    if (probe) { draws[s] = 1; return; }
    const H rn = rngf(s);
    int levnew = levold;
    H ffraction = HK(0.), dlevfrac = HK(0.);
    const H totlevmass = VV(dpr, levold) / ga;
    bool found = false;
    const H *fmw = walk ? Sx.fm_walk() + (size_t)levold * Sx.nv : nullptr;      // walk != 0: row (forward run) / column (backward) levold, consecutive
    for (int k = 1; k <= nconvtop && !found; k += 8) {
      H f8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int q = I_MIN(k + u, nconvtop);
        f8[u] = walk ? fmw[q] : ldirect == 1 ? MM(fmass, levold, q) : MM(fmass, q, levold);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (found || k + u > nconvtop) break;
        const H f = f8[u];
        ffraction = ffraction + f / totlevmass;
        if (rn <= ffraction) {
          levnew = k + u;
          if (ffraction > HK(1.e-20)) dlevfrac = (ffraction - rn) / f * totlevmass;
          else dlevfrac = HK(0.5);
          found = true;
        }
      }
    }
    if (levnew <= nconvtop) {
      if (levnew != levold) {
        const H dlogp = (HK(1.) - dlevfrac) * (M<H>::log(VV(phconv, levnew + 1)) - M<H>::log(VV(phconv, levnew)));
        const H pint = M<H>::log(VV(phconv, levnew)) + dlogp;
        const H dz1 = pint - M<H>::log(VV(phconv, levnew));
        const H dz2 = M<H>::log(VV(phconv, levnew + 1)) - pint;
        const H dz = dz1 + dz2;
        znew = (VV(uvzlev, levnew) * dz2 + VV(uvzlev, levnew + 1) * dz1) / dz;
        if (znew < HK(0.)) znew = HK(-1.) * znew;
        touched = true;
      }
    }
    if (levnew <= nconvtop && levnew == levold) {
      ztold = znew;
      H wsub_lo, wsub_hi;
      if (levold > 1) {
        const H temp_levold = VV(tconv, levold - 1) + (VV(tconv, levold) - VV(tconv, levold - 1)) * (VV(pconv, levold - 1) - VV(phconv, levold)) /
                                                          (VV(pconv, levold - 1) - VV(pconv, levold));
        const H sub_levold = VV(sub, levold) / (HK(1.) - VV(sub, levold) / VV(dpr, levold) * ga);
        wsub_lo = HK(-1.) * sub_levold * r_air * temp_levold / (VV(phconv, levold));
      } else wsub_lo = HK(0.);
      const H temp_levold1 = VV(tconv, levold) + (VV(tconv, levold + 1) - VV(tconv, levold)) * (VV(pconv, levold) - VV(phconv, levold + 1)) /
                                                     (VV(pconv, levold) - VV(pconv, levold + 1));
      const H sub_levold1 = VV(sub, levold + 1) / (HK(1.) - VV(sub, levold + 1) / VV(dpr, levold + 1) * ga);
      wsub_hi = HK(-1.) * sub_levold1 * r_air * temp_levold1 / (VV(phconv, levold + 1));
      const H dz1 = ztold - VV(uvzlev, levold);
      const H dz2 = VV(uvzlev, levold + 1) - ztold;
      const H dz = dz1 + dz2;
      const H wsubpart = (dz2 * wsub_lo + dz1 * wsub_hi) / dz;
      znew = ztold + wsubpart * (H)lsynctime;
      if (znew < HK(0.)) znew = HK(-1.) * znew;
      touched = true;
    }
  } else if (probe) return;
  if (znew > height_nz - HK(0.5)) { znew = height_nz - HK(0.5); touched = true; }
  if (touched) {
    zt[s] = (R)znew;
    if (nmoved) atomicAdd(nmoved, 1ull);
  }
}

#undef VV
#undef MM
#undef HK
#undef R_ABS
#undef R_MAX
#undef R_MIN
#undef I_MAX
#undef I_MIN

}  // namespace conv
FPX_TU_CLOSE
}  // namespace fpx
