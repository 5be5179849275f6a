// fpx_device.hpp -- per-particle physics of the trajectory step, written for
// one GPU thread per particle (gfx950).  All state the reference keeps in
// Fortran module globals (interpol_mod.f90:7-16, hanna_mod.f90:5-6) lives in
// registers of the owning thread here, so the routine is re-entrant and
// order-independent.  Reference citations are relative to /root/reference/src.
//
// Templated on the arithmetic type R (double: the all-fp64 build of
// BASELINE.json configs 2-3; float: the reference's own typing, where only the
// horizontal position is 8-byte).  Constants are written K(x) so that they
// round like the Fortran literals of a default-real-R build.
#pragma once
#include "fpx_tu.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpx {
FPX_TU_OPEN

#define K(x) ((R)(x))
#define FPX_DEV __device__ __forceinline__
#define FPX_HD __host__ __device__ __forceinline__

constexpr int kMaxSpec = 5;
constexpr int kMaxNests = 4;
constexpr int kDead = -999999999;

// ---------------------------------------------------------------------------
// math in R
// ---------------------------------------------------------------------------
FPX_HD float m_exp(float x) { return expf(x); }
FPX_HD double m_exp(double x) { return exp(x); }
FPX_HD float m_log(float x) { return logf(x); }
FPX_HD double m_log(double x) { return log(x); }
FPX_HD float m_sqrt(float x) { return sqrtf(x); }
FPX_HD double m_sqrt(double x) { return sqrt(x); }
// a*b + c with the contraction spelled out: for a*b + c*d the compiler may fuse either product, and two instantiations of
// the same template (k_prep with and without the branch for new particles) did choose differently
FPX_HD float m_fma(float a, float b, float c) { return fmaf(a, b, c); }
FPX_HD double m_fma(double a, double b, double c) { return fma(a, b, c); }
FPX_HD float m_sin(float x) { return sinf(x); }
FPX_HD double m_sin(double x) { return sin(x); }
FPX_HD float m_cos(float x) { return cosf(x); }
FPX_HD double m_cos(double x) { return cos(x); }
FPX_HD float m_erf(float x) { return erff(x); }
FPX_HD double m_erf(double x) { return erf(x); }
FPX_HD float m_pow(float x, float y) { return powf(x, y); }
FPX_HD double m_pow(double x, double y) { return pow(x, y); }
template <typename R> FPX_HD R m_abs(R x) { return x < 0 ? -x : x; }
template <typename R> FPX_HD R m_max(R a, R b) { return a > b ? a : b; }
template <typename R> FPX_HD R m_min(R a, R b) { return a < b ? a : b; }
template <typename R> FPX_HD R m_sign(R a, R b) { R m = m_abs(a); return (b < 0 || (b == 0 && signbit(b))) ? -m : m; }
// The floating-point forms as the hardware has them: |x| is a source modifier (free), max / min one instruction, sign() one bit-field
// insert -- written as comparisons and selects each costs a compare and two conditional moves per fp64 value (9.5 issue cycles
// where the instruction takes 4.4 or nothing), and the fine loop of the Langevin kernel has ten of them per sub-step.  They differ
// from the select forms only for NaN operands (max / min return the other operand) and for -0.0 (abs(-0.0) = +0.0, as in Fortran).
FPX_HD float m_abs(float x) { return __builtin_fabsf(x); }
FPX_HD double m_abs(double x) { return __builtin_fabs(x); }
FPX_HD float m_max(float a, float b) { return __builtin_fmaxf(a, b); }
FPX_HD double m_max(double a, double b) { return __builtin_fmax(a, b); }
FPX_HD float m_min(float a, float b) { return __builtin_fminf(a, b); }
FPX_HD double m_min(double a, double b) { return __builtin_fmin(a, b); }
FPX_HD float m_sign(float a, float b) { return __builtin_copysignf(a, b); }     // sign(a,b) of Fortran: |a| with the sign bit of b
FPX_HD double m_sign(double a, double b) { return __builtin_copysign(a, b); }
// ---------------------------------------------------------------------------
// Cheap fp64 building blocks for the Langevin inner loop.  Issue costs measured on MI355X
// with tools/valu_rates.hip (cycles per wave64 instruction): add/mul/fma f64 4.4, v_rcp/v_rsq_f64 16,
// 32-bit ALU 2.4, f32 transcendental 8; library calls: a/b 64, sqrt 89, exp 98, erf 246,
// log 358, pow 780.  The loop is VALU-bound, so the helpers below trade the last half ulp
// (results are within 1-2 ulp instead of correctly rounded) for 2-4x fewer cycles.
// The float overloads keep the plain operations.
// ---------------------------------------------------------------------------
FPX_DEV float m_rcp(float b) { return __builtin_amdgcn_rcpf(b); }   // v_rcp_f32, 1 ulp
FPX_DEV double m_rcp(double b) {   // 1/b, b finite and non-zero: hardware seed + two Newton steps (34 cycles)
  double r = __builtin_amdgcn_rcp(b);
  double e = fma(-b, r, 1.0);
  r = fma(r, e, r);
  e = fma(-b, r, 1.0);
  return fma(r, e, r);
}
// a/b where the divisor is finite and non-zero by construction: m_rcp + one residual correction
FPX_DEV float m_divf(float a, float b) { return a / b; }
FPX_DEV double m_divf(double a, double b) {
  double r = m_rcp(b);
  double q = a * r;
  return fma(fma(-b, q, a), r, q);
}
// sqrt(x) and 1/sqrt(x) together for x > 0 (normal range): v_rsq_f64 + one coupled
// Goldschmidt step + one residual correction each (56 cycles for both)
FPX_DEV void m_sqrt_rsqrt(float x, float &s, float &rs) { s = __builtin_amdgcn_sqrtf(x); rs = __builtin_amdgcn_rsqf(x); }   // v_sqrt_f32 / v_rsq_f32, 1 ulp
FPX_DEV void m_sqrt_rsqrt(double x, double &s, double &rs) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  r = fma(-h, g, 0.5);
  h = fma(h, r, h);
  s = g;
  rs = h + h;
}
FPX_DEV float m_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
FPX_DEV double m_rsqrt(double x) {   // x > 0
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  h = fma(h, r, h);
  return h + h;
}
// sqrt(x) for x >= 0 (0 allowed), 47 cycles
FPX_DEV float m_sqrtp(float x) { return __builtin_amdgcn_sqrtf(x); }
FPX_DEV double m_sqrtp(double x) {
  // rsq(0) = inf would make 0 * inf = NaN: bounded by 1e300 (1/sqrt of the smallest subnormal is 4.5e161) every step below
  // gives exactly 0 for x = 0 -- one v_min_f64 instead of a compare and two conditional moves on the result
  double y = __builtin_fmin(__builtin_amdgcn_rsq(x), 1.0e300);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  return g;
}
// log(x) for the x**y = exp(y*log x) of the turbulence profiles: argument reduction to
// [sqrt(1/2), sqrt(2)) and the degree-7 series in s = f/(2+f) (the classical fdlibm scheme);
// absolute error < 3e-16 * max(1, |log x|).  Zero, negative, subnormal, inf and NaN take the
// library path.
FPX_DEV float m_logp(float x) { return __logf(x); }   // v_log_f32 based, ~2 ulp
FPX_DEV double m_logp(double x) {
  if (__builtin_expect(!(x >= 2.3e-308 && x <= 1.7e308), 0)) return log(x);
  int e;
  double m = frexp(x, &e);
  const bool lo = m < 0.70710678118654752;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f * m_rcp(2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double Rr = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + Rr) + dk * 1.90821492927058770002e-10)) - f);
}
// ln(x) to an ABSOLUTE error of 1e-13 (not a relative one) for positive normal x, at a third of m_logp's instructions: the mantissa
// m in [0.5, 1) is divided by the centre c of its 1/64-wide interval (32-entry table of 1/c and ln c, 512 B; the Langevin kernel
// reads a copy in LDS), which leaves log1p(f) with |f| <= 2^-6 for a degree-6 series (truncation f^7/7 < 4e-14).  Used where a
// logarithm only feeds a factor x**delta with |delta| ~ 1e-5 (hanna_short's exponents .33333 and .66666 against 1/3 and 2/3):
// the result's relative error is delta times this absolute one.
static __device__ const double kLogTab[32][2] = {
  {1.9692307692307693, -0.6776429940239801},
  {1.9104477611940298, -0.6473376445286511},
  {1.855072463768116, -0.6179237593223578},
  {1.8028169014084507, -0.5893503868783018},
  {1.7534246575342465, -0.561570822771226},
  {1.7066666666666668, -0.5345421503833068},
  {1.6623376623376624, -0.5082248420659333},
  {1.620253164556962, -0.48258241145259567},
  {1.5802469135802468, -0.4575811092471784},
  {1.5421686746987953, -0.43318965612301924},
  {1.5058823529411764, -0.4093790074293007},
  {1.471264367816092, -0.38612214526503347},
  {1.4382022471910112, -0.3633938941874773},
  {1.4065934065934067, -0.34117075740276714},
  {1.3763440860215055, -0.3194307707663612},
  {1.3473684210526315, -0.29815337231907635},
  {1.3195876288659794, -0.27731928541623435},
  {1.292929292929293, -0.2569104137850272},
  {1.2673267326732673, -0.2369097470783577},
  {1.2427184466019416, -0.2173012756899814},
  {1.2190476190476192, -0.1980699137620938},
  {1.1962616822429906, -0.179201429457711},
  {1.1743119266055047, -0.16068238169047347},
  {1.1531531531531531, -0.14250006260728304},
  {1.1327433628318584, -0.1246424452072766},
  {1.1130434782608696, -0.1070981355563671},
  {1.0940170940170941, -0.08985632912186105},
  {1.0756302521008403, -0.07290677080808779},
  {1.0578512396694215, -0.05623971832287608},
  {1.0406504065040652, -0.039845908547199674},
  {1.024, -0.023716526617316044},
  {1.0078740157480315, -0.007843177461025893}
};
// a*b + c with the literal c held in a scalar register pair.  Left to itself the compiler picks the two-address v_fmac_f64,
// whose addend must sit in vector registers: two v_mov_b32 per Horner step (4.8 issue cycles next to the fma's 4.4), since
// the literals are re-materialised inside the loop (MachineLICM is off, see __graft_entry__.py).  The three-address v_fma_f64
// reads the constant from SGPRs, which the otherwise idle scalar unit loads.
FPX_DEV double m_fma_k(double a, double b, double c /* compile-time constant */) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}
// 2**(j/32), j = 0 .. 31 (m_exp_tab)
static __device__ const double kExpTab[32] = {
  1.0, 1.0218971486541166, 1.0442737824274138, 1.0671404006768237,
  1.0905077326652577, 1.1143867425958924, 1.1387886347566916, 1.1637248587775775,
  1.189207115002721, 1.215247359980469, 1.241857812073484, 1.2690509571917332,
  1.2968395546510096, 1.3252366431597413, 1.3542555469368927, 1.383909881963832,
  1.4142135623730951, 1.4451808069770467, 1.4768261459394993, 1.5091644275934228,
  1.5422108254079407, 1.5759808451078865, 1.6104903319492543, 1.645755478153965,
  1.681792830507429, 1.718619298122478, 1.7562521603732995, 1.7947090750031072,
  1.8340080864093424, 1.8741676341103, 1.9152065613971474, 1.9571441241754002
};
// The two tables as the Langevin kernel reads them: a copy per block in LDS, [0..63] = kLogTab, [64..95] = kExpTab.  768 bytes:
// the kernel's 25-slot stash + height column + these must stay under 53248 B per block for three blocks per CU (the hardware
// allocates LDS in coarser units than hipOccupancyMaxActiveBlocksPerMultiprocessor assumes: 54352 B was measured to hold two).
constexpr int kLdsTabDoubles = 96, kLdsExpTabAt = 64;
typedef __attribute__((address_space(3))) const double *lds_tab_ptr;
template <typename TP>
FPX_DEV double m_log_abs(double x, TP tab /* kLogTab layout: [32][2] */) {
  const double m = __builtin_amdgcn_frexp_mant(x);                  // [0.5, 1)
  const int e = __builtin_amdgcn_frexp_exp(x);
  const unsigned int j = (__double2hiint(m) >> 15) & 31u;           // top five bits of the fraction
  const double ic = tab[2 * j], lc = tab[2 * j + 1];
  const double f = fma(m, ic, -1.0);
  double p = m_fma_k(f, -1.0 / 6.0, 0.2);
  p = m_fma_k(f, p, -0.25);
  p = m_fma_k(f, p, 1.0 / 3.0);
  p = fma(f, p, -0.5);
  p = fma(f, p, 1.0);
  return fma((double)e, 6.93147180559945309417e-01, lc) + f * p;
}
FPX_DEV double m_log_abs(double x) { return m_log_abs(x, &kLogTab[0][0]); }
// exp(x) with a 32-entry table of 2**(j/32) (Tang's scheme): k = rint(x*32/ln2), r = x - k*ln2/32 (two-part constant, |r| <= 0.011),
// exp(x) = 2**(k>>5) * T[k&31] * (1 + r + r^2/2 + .. + r^6/720) (truncation r^7/5040 < 4e-18).  16 instructions against the 21 of
// m_expp; about 1 ulp.  Saturates like m_expp (v_ldexp_f64), NaN propagates.
template <typename TP>
FPX_DEV double m_exp_tab(double x, TP tab /* kExpTab layout */) {
  const double k = rint(x * 46.16624130844683);
  double r = fma(k, -0.02166084938653512, x);
  r = fma(k, -5.9631716539705866e-12, r);
  const int ki = (int)k;
  const double t = tab[ki & 31];
  double p = m_fma_k(r, 1.3888888888888888889e-03, 8.3333333333333333333e-03);
  p = m_fma_k(r, p, 4.1666666666666666667e-02);
  p = m_fma_k(r, p, 1.6666666666666666667e-01);
  p = fma(r, p, 0.5);
  p = fma(r, p, 1.0);
  return ldexp(fma(t, r * p, t), ki >> 5);
}
// x**(-1/3) for x in the f32 exponent range: f32 seed, two Newton steps r <- r + r*(1 - x*r^3)/3
FPX_DEV double m_rcbrt(double x) {
  double r = (double)__builtin_amdgcn_exp2f(__log2f((float)x) * (-1.0f / 3.0f));
  const double third = 1.0 / 3.0;
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const double e = fma(-(x * (r * r)), r, 1.0);
    r = fma(r * third, e, r);
  }
  return r;
}
// exp(x): k = rint(x/ln2), r = x - k*ln2 (two-part ln2), exp(r) by the degree-13 Taylor polynomial
// split into even and odd halves (|r| <= 0.347: truncation 4e-18), scaled by ldexp.  Leaves out the
// library's range screening: v_ldexp_f64 saturates to 0 / inf by itself, NaN propagates.
FPX_DEV float m_expp(float x) { return __expf(x); }   // v_exp_f32 based, ~2 ulp
FPX_DEV double m_expp(double x) {
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  const double r2 = r * r;
  double pe = fma(r2, 1.1470745597729724714e-11, 2.0876756987868098979e-09);   // 1/14!, 1/12!
  pe = fma(r2, pe, 2.7557319223985890653e-07);                                  // 1/10!
  pe = fma(r2, pe, 2.4801587301587301587e-05);                                  // 1/8!
  pe = fma(r2, pe, 1.3888888888888888889e-03);                                  // 1/6!
  pe = fma(r2, pe, 4.1666666666666666667e-02);                                  // 1/4!
  pe = fma(r2, pe, 0.5);                                                        // 1/2!
  double po = fma(r2, 1.6059043836821614599e-10, 2.5052108385441718775e-08);   // 1/13!, 1/11!
  po = fma(r2, po, 2.7557319223985890653e-06);                                  // 1/9!
  po = fma(r2, po, 1.9841269841269841270e-04);                                  // 1/7!
  po = fma(r2, po, 8.3333333333333333333e-03);                                  // 1/5!
  po = fma(r2, po, 1.6666666666666666667e-01);                                  // 1/3!
  // exp(r) = 1 + r + r2*(pe + r*po)
  const double p = fma(r2, fma(r, po, pe), r) + 1.0;
  return ldexp(p, (int)k);
}
// erf(x) where E = exp(-x*x) is already known (cbl.f90 evaluates the Gaussian next to its error function):
// erf(x) = sign(x) * (1 - E*g(|x|)), g = exp(x^2)*erfc(x) by a degree-17 polynomial in t = (x-3)/(x+3) on
// 0 <= x <= 6.5 (Chebyshev fit in 50-digit arithmetic; beyond 6.5 E < 5e-19 and the clamp is invisible).
// Branch-free: 40 instructions where the library's two-range erf costs 110 per wave as soon as the lanes
// straddle |x| = 1.  Absolute error <= 5e-16 (the relative accuracy of erf near 0 is given up: the value is
// only used as a term of O(1) sums).
// The coefficients sit in constant memory: the wave fetches them with three scalar loads (s_load_dwordx16/x4) instead of
// materialising 18 fp64 literals with 36 s_mov_b32 per call -- a wave issues one instruction per turn of its SIMD, so every
// scalar instruction it spends is a turn in which it cannot feed the VALU (three waves per SIMD do not hide that).
__constant__ double kErfC[18] = {      // not static, not const: a constant the compiler could fold back into literals
  -4.6179706490756721764e-8, -2.4893389607188758124e-7, -2.1766619811183263845e-7, 1.3314749826389837962e-6,
  2.1314123772486134504e-6, -8.0027446235445720124e-6, -1.2872811953274865798e-5, 6.4058505135542794022e-5,
  4.52577749220917884e-5, -5.970618059279705004e-4, 7.0774621626041840357e-4, 4.2691363041920969877e-3,
  -2.439249930876277646e-2, 7.1665837199359721633e-2, -1.5011593650098095142e-1, 2.4560380171230996017e-1,
  -3.2623356004303588051e-1, 1.7900115118138999674e-1};
template <typename T>
FPX_DEV T m_erf_e(T x, T E) {
  const T ax = m_abs(x);
  const T xc = m_min(ax, (T)6.5);
  const T t = (xc - (T)3.0) * m_rcp(xc + (T)3.0);
  T p = (T)kErfC[0];
#pragma unroll
  for (int i = 1; i < 18; i++) p = p * t + (T)kErfC[i];
  const T r = (T)1.0 - E * p;
  return m_sign(r, x);
}
// Two error functions at once (cbl.f90:195-204 needs erf(aperfa) and erf(aperfb)): one fetch of the coefficients serves both
// Horner chains.  The pointer passes through an empty asm so that the scalar loads stay where they are used -- hoisted to
// the kernel's prologue (the table is loop-invariant) the 36 SGPRs would be spilled to vector lanes for the whole kernel.
template <typename T>
FPX_DEV void m_erf_e2(T xa, T Ea, T xb, T Eb, T &ra, T &rb) {
  typedef const double __attribute__((address_space(4))) *const_ptr;      // constant address space: uniform loads are scalar loads
  const_ptr c = (const_ptr)kErfC;
  asm volatile("" : "+s"(c));
  const T aa = m_abs(xa), ab = m_abs(xb);
  const T ca = m_min(aa, (T)6.5), cb = m_min(ab, (T)6.5);
  const T ta = (ca - (T)3.0) * m_rcp(ca + (T)3.0), tb = (cb - (T)3.0) * m_rcp(cb + (T)3.0);
  T pa = (T)c[0], pb = (T)c[0];
#pragma unroll
  for (int i = 1; i < 18; i++) { const T k = (T)c[i]; pa = pa * ta + k; pb = pb * tb + k; }
  const T qa = (T)1.0 - Ea * pa, qb = (T)1.0 - Eb * pb;
  // the sign of x by a bit-field insert instead of a compare and two selects; qa, qb >= 0 up to the rounding of 1 - E*p at x = 0,
  // where the two forms can differ by an ulp of 1 in the sign of a value of that size
  ra = m_sign(qa, xa);
  rb = m_sign(qb, xb);
}
// x**y for the reference's non-integer exponents (0.33333, 0.66666, 0.8 ...).  fp64: exp(y*log(x)),
// relative error <= ~|y ln x| ulp (a few 1e-16 here) at a fraction of the cost of the
// correctly-rounded pow (fp32: a few ulp of powf at a third of its cost).  x == 0 and x < 0 behave like pow (0/inf, NaN).
FPX_DEV float m_powr(float x, float y) { return x > 0.0f ? expf(y * logf(x)) : powf(x, y); }
FPX_DEV double m_powr(double x, double y) { return m_expp(y * m_logp(x)); }
// cos(x) for |x| <= pi/2 + a little (the latitude of a particle in radians, advance.f90:755): even Taylor
// polynomial to x^26 (remainder 1.7^28/28! = 9e-24), no argument reduction.  14 instructions instead of the
// library's ~90; absolute error < 2e-16.
FPX_DEV float m_coslat(float x) { return cosf(x); }
FPX_DEV double m_coslat(double x) {
  if (!(fabs(x) < 1.7)) return cos(x);
  const double z = x * x;
  double p = -1.0 / 403291461126605635584000000.0;          // -1/26!
  p = fma(p, z, 1.0 / 620448401733239439360000.0);           //  1/24!
  p = fma(p, z, -1.0 / 1124000727777607680000.0);            // -1/22!
  p = fma(p, z, 1.0 / 2432902008176640000.0);                //  1/20!
  p = fma(p, z, -1.0 / 6402373705728000.0);                  // -1/18!
  p = fma(p, z, 1.0 / 20922789888000.0);                     //  1/16!
  p = fma(p, z, -1.0 / 87178291200.0);                       // -1/14!
  p = fma(p, z, 1.0 / 479001600.0);                          //  1/12!
  p = fma(p, z, -1.0 / 3628800.0);                           // -1/10!
  p = fma(p, z, 1.0 / 40320.0);                              //  1/8!
  p = fma(p, z, -1.0 / 720.0);                               // -1/6!
  p = fma(p, z, 1.0 / 24.0);                                 //  1/4!
  p = fma(p, z, -0.5);
  return fma(p, z, 1.0);
}
// x * icbt for icbt = +-1 kept as a sign mask (0 | 0x80000000): one v_xor_b32 on the word with the sign bit instead of an
// int -> fp64 conversion and a multiplication (8.7 issue cycles) in every fine sub-step; the same bits
FPX_DEV double m_flip(double x, unsigned int flip) { return __hiloint2double(__double2hiint(x) ^ (int)flip, __double2loint(x)); }
FPX_DEV float m_flip(float x, unsigned int flip) { return __uint_as_float(__float_as_uint(x) ^ flip); }
// is x a positive number the f32 seeds below can hold (2^-122 <= x < 2^122)?  One integer subtraction and one unsigned compare on
// the high word instead of two fp64 compares: zero, negative numbers, subnormals, infinities and NaN all fall outside.
FPX_DEV bool m_in_seed_range(double x) { return (unsigned int)(__double2hiint(x) - 0x38500000) < (unsigned int)(0x47900000 - 0x38500000); }
// x**0.8 for 0 <= x (hanna.f90:97, hanna_short.f90:80: tlw = 0.1*h/sigw*zeta**0.8): 0.8 = 4/5, so x**0.8 = x*r with
// r = x**(-1/5) from an f32 seed and two Newton steps r <- r + r*(1 - x*r^5)/5 (the literal 0.8 differs from 4/5
// by 4e-17: invisible).  19 instructions instead of log + exp (58).
FPX_DEV float m_pow08(float x) { return x > 0.0f ? __expf(0.8f * __logf(x)) : 0.0f; }
FPX_DEV double m_pow08(double x) {
  if (__builtin_expect(!m_in_seed_range(x), 0)) return x > 0.0 ? m_expp(0.8 * m_logp(x)) : (x == 0.0 ? 0.0 : x * __builtin_nan(""));   // x < 0: NaN like pow
  double r = (double)__builtin_amdgcn_exp2f(__log2f((float)x) * -0.2f);
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const double r2 = r * r, r5 = (r2 * r2) * r;
    const double e = fma(-x, r5, 1.0);
    r = fma(r * 0.2, e, r);
  }
  return x * r;
}
// c = x**0.333333333 and ic2 = x**(-2*0.333333333) for x > 0 (the two "cuberoot" calls of cbl.f90:115-121,
// exponent as written at cbl.f90:227).  fp64: r = x**(-1/3) from an f32 seed and two Newton steps
// r <- r + r*(1 - x*r^3)/3, then the difference between 1/3 and 0.333333333 as the first-order
// factor exp(-+delta*ln x) with ln x from the f32 logarithm (delta = 3.3e-10, so its 1e-7 relative
// error is invisible).  Arguments outside the f32 exponent range take the exp/log path.
FPX_DEV void m_cuberoot_parts(float x, float &c, float &ic2) {
  const float l = __logf(x);
  c = __expf(0.333333333f * l);
  ic2 = __expf(-2.0f * 0.333333333f * l);
}
FPX_DEV void m_cuberoot_parts(double x, double &c, double &ic2) {
  if (__builtin_expect(!m_in_seed_range(x), 0)) {
    const double l = log(x);
    c = exp(0.333333333 * l);
    ic2 = exp(-2.0 * 0.333333333 * l);
    return;
  }
  const float l2 = __log2f((float)x);
  double r = (double)__builtin_amdgcn_exp2f(l2 * (-1.0f / 3.0f));
  const double third = 1.0 / 3.0;
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const double e = fma(-(x * (r * r)), r, 1.0);
    r = fma(r * third, e, r);
  }
  const double dl = (1.0 / 3.0 - 0.333333333) * (double)(l2 * 0.693147181f);   // delta * ln x
  const double rc = fma(r, dl, r);              // x**(-0.333333333)
  const double c0 = x * (r * r);                // x**(1/3)
  c = fma(-c0, dl, c0);                         // x**(0.333333333)
  ic2 = rc * rc;
}
FPX_HD float m_fmod(float x, float y) { return fmodf(x, y); }
FPX_HD double m_fmod(double x, double y) { return fmod(x, y); }
FPX_HD double d_modulo(double a, double p) { double r = fmod(a, p); if (r != 0.0 && ((r < 0) != (p < 0))) r += p; return r; }

// ---------------------------------------------------------------------------
// device view of everything the path reads (com_mod / par_mod variables)
// ---------------------------------------------------------------------------
template <typename R>
struct NestDesc {
  int nx, ny;                       // nxn, nyn
  R xl, yl, xr, yr, xres, yres;     // xln, yln, xrn, yrn, xresoln, yresoln
  const R *w3, *r2, *sfc, *hcell, *tropo, *vdep;
};

template <typename R>
struct View {
  // grid, com_mod.f90:298-299,551-560
  int nx, ny, nz, nxmin1, nymin1, nmixz;
  R dx, dy, xlon0, ylat0, dxconst, dyconst;
  int xglobal, nglobal, sglobal;
  R switchnorthg, switchsouthg;
  R northpolemap[9], southpolemap[9];
  const R *polemaps;   // the same 18 values in device memory (the out-of-line polar move reads them there)
  // wind-field window, com_mod.f90:276,286
  int memtime0, memtime1, m1, m2, lwindinterv;   // m1/m2: physical slot (0|1) of memind(1)/(2)
  R meso_r, meso_rs;   // r = exp(-2*|lsynctime|/lwindinterv), sqrt(1-r*r): advance.f90:728-729, set with the wind window
  // switches
  int ldirect, lsynctime, method, mintime, ifine, turbswitch, cblflag, mdomainfill, lsettling;
  int turboff, interpolhmix;     // com_mod.f90:777-778 as run-time switches (fpx_config)
  int pbl_cost_buckets;          // work-list order: cost buckets below the stability class (k_prep; scheduling only)
  int nspec, drydep, drydepspec[kMaxSpec];
  R ctl, fine, d_trop, d_strat, turbmesoscale;
  R density[kMaxSpec], dquer[kMaxSpec], vsetaver[kMaxSpec], cunningham[kMaxSpec], decay[kMaxSpec];   // for compile-time subscripts only
  const R *spec;                 // [kMaxSpec][4] = (density, dquer, vsetaver, cunningham) of each species in device memory: what a per-lane subscript reads (spec_row)
  // release-point tables of point_mod (fpx_set_release_points), indexed per lane by npoint(j): device memory
  int numpoint, mquasilag;
  const R *rel_xmass;            // [maxspec][numpoint]: xmass(numpoint, maxspec), column-major like the host's
  const int *rel_npart;          // [numpoint]
  const signed char *rel_nsp;    // [numpoint] 0-based species of the settling pick (advance.f90:518-524), precomputed on the host
  int lage_last;
  // fields, device layout (see DESIGN.md "data layout in HBM"):
  const R *height;   // [nz]
  const R *w3;       // [ny][nx][nz][2 slots][3]  (uu, vv, ww)
  const R *w3pol;    // same with (uupol, vvpol, ww); only when a pole is in the grid
  // Time-blended copies of w3 for the step in flight, [jy][ix][iz][3], or NULL (see Engine::blend_winds): the wind at itime
  // (w3t0) and at itime + lsynctime*ldirect (w3t1, the Petterssen re-interpolation).  The time weights are the same for every
  // particle of a step, so a gather that needs no standard deviations reads half the bytes from these.
  const R *w3t0, *w3t1;
  const R *r2t0;     // the same for (rho, drhodz) at itime, [jy][ix][iz][2]: the Langevin kernel's profile fetch
  const R *r2;       // [ny][nx][nz][2 slots][2]  (rho, drhodz)
  const R *sfc;      // [ny][nx][2 slots][4]      (ustar, wstar, oli, hmix)
  const R *hcell;    // [ny][nx]  max of hmix over the cell's 4 corners x 2 slots
  const R *tropo;    // [ny][nx]  tropopause, literal time slot 1 (advance.f90:253)
  const R *vdep;     // [ny][nx][2 slots][nspec]
  const R *rhott;    // [ny][nx][nz][2] (rho, tt) of literal slot 1 (get_settling.f90:83-84)
  // nested grids (com_mod.f90:464-541): same packed layouts, nest extents nxn x nyn.  The table
  // is subscripted per lane, so it lives in device memory: a by-value array with a divergent
  // subscript would drag the whole kernel argument (and every uniform scalar) out of SGPRs.
  int numbnests;
  const NestDesc<R> *nest;   // [numbnests]
  R eps;             // nxmax/3.e5 with the host's par_mod nxmax (advance.f90:107)
  // RNG
  const R *rannumb;  // [maxrand], 0-based copy of rannumb(1:maxrand)
  int maxrand, rng_mode;
  unsigned long long seed;
  unsigned int pid_base;   // global number of this rank's particle 0 (fpx_config.particle_base): key of the counter RNG
};

// A per-lane (divergent) subscript into a small table held in REGISTERS: a select chain over constant subscripts.
// Never use it on a table inside View: a dynamic subscript forces the whole kernel argument into private (scratch) memory and
// with it every uniform scalar into vector registers -- and a select chain does not prevent that: the optimiser sees the
// function before it is inlined, View then being a plain pointer, and merges the conditional loads into ONE load at a selected
// address (measured: 576 B of scratch per lane and 13 ms in the f32 k_prep instances with initialize()); reading all elements
// first avoids it but holds 40 scalar registers, which the polar k_prep instances paid with 22 spilled VGPRs (0.72 -> 0.85 ms).
// Tables with a per-lane subscript live in device memory (View::nest, View::spec, View::polemaps).
template <typename T, int N>
FPX_DEV T pick(const T (&a)[N], int l) {
  T r = a[0];
#pragma unroll
  for (int k = 1; k < N; k++) r = (l == k) ? a[k] : r;
  return r;
}

// particle SoA in HBM (com_mod.f90:678-695), R-typed except the position
template <typename R>
struct Parts {
  double *xt, *yt;
  R *zt, *up, *vp, *wp, *us, *vs, *ws;
  int *idt, *itra1, *itramem, *npoint, *nclass;
  int *itrasplit;        // com_mod.f90:683 (release + splitting on the device)
  short *cbt;
  R *xmass1;             // [nspec][cap]
  R *xscav;              // [nspec][cap] xscav_frac1 (com_mod.f90:683,712), only in backward runs with DRYBKDEP / WETBKDEP; else null
  unsigned int *pid;     // reference particle number - 1 (stable across locality sorts)
  long long cap;
};

// per-step RNG inputs of the TABLE_SEQ (parity) mode, indexed by pid
struct SeqRng {
  const int *nrand_adv;    // start index for advance (advance.f90:153)
  const int *nrand_init;   // start index for initialize (initialize.f90:68); 0 = not initialised
  const float *cbl_dcas;   // initialize_cbl_vel.f90:75 uniform
  const float *cbl_dcas1;  // initialize_cbl_vel.f90:78/81 gaussian
  const double *cbl_dcas_d, *cbl_dcas1_d;  // same in fp64 hosts
};

struct Stats {
  unsigned long long n_due, n_init, n_left, n_minmass, n_maxage, nan_count, nan_count2, n_badpos;
  // diagnostics of an instrumented build (-DFPX_LANE_STATS): per code region of the Langevin kernel, how often a wave ran it
  // and with how many lanes (fpx_lane_stats)
  unsigned long long lanes[16][2];
};
// regions: 0 a pass, 1 a fine sub-step, 2 its CBL branch, 3 the Gaussian branch under cblflag, 4 the exponential-form branch,
// 5/6/7 hanna_short neutral / unstable / stable, 8 lane refill, 9 hand-over of a finished particle, 10 the kernel's loop iterations
#ifdef FPX_LANE_STATS
#define FPX_LANES(st, region)                                                                     \
  do {                                                                                            \
    const unsigned long long m_ = __ballot(1);                                                    \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) {                                  \
      atomicAdd(&(st)->lanes[region][0], 1ull);                                                   \
      atomicAdd(&(st)->lanes[region][1], (unsigned long long)__popcll(m_));                       \
    }                                                                                             \
  } while (0)
#else
#define FPX_LANES(st, region) do {} while (0)
#endif

// ---------------------------------------------------------------------------
// counter-based generator: Philox4x32 (Salmon et al. 2011), keyed by seed,
// counter = (particle id, step, draw index).  Seven rounds: the fewest for which the authors report that the
// generator passes BigCrush ("Crush-resistant", their Table 2; ten is their default with a safety margin) -- the
// rounds are 32-bit multiplies at a quarter of the VALU rate, and the Langevin kernel is VALU-bound.
// ---------------------------------------------------------------------------
constexpr int kPhiloxRounds = 7;
FPX_DEV void philox4x32(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                        unsigned int k0, unsigned int k1, unsigned int out[4]) {
#pragma unroll
  for (int r = 0; r < kPhiloxRounds; r++) {
    // the 64-bit product in ONE instruction (v_mad_u64_u32, 4.3 issue cycles) instead of v_mul_hi_u32 + v_mul_lo_u32 (8.3)
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
    const unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
    unsigned int n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Gaussian table value source: either the reference's table or its counter twin.
// MODE >= 0 fixes the rng mode at compile time (hot kernels), -1 reads it at run time.
template <typename R, int MODE = -1>
struct Rng {
  const R *tab;
  int maxrand, mode;
  unsigned int pid, step, k0, k1;
  // counter mode: one Philox call yields four normals (two Box-Muller pairs); the last block is
  // kept so that consecutive draws share it.  Copies taken before the first draw start empty.
  mutable unsigned int cblk;
  mutable float c0, c1, c2, c3;
  static constexpr bool kCounter = MODE == 2;
  // rannumb(idx), 1-based like the reference
  FPX_DEV R at(int idx) const {
    if ((MODE < 0 ? mode : MODE) != 2) return tab[min(idx, maxrand) - 1];   // the reference reads past the table on rare CBL re-draws; clamp instead
    const unsigned int blk = (unsigned int)idx >> 2;
    if (blk != cblk) {
      unsigned int o[4];
      philox4x32(pid, step, blk, 0x47415553u, k0, k1, o);
      // clipped Box-Muller pairs (the distribution of gasdev1, random_mod.f90:70-90), on the hardware's f32 transcendentals:
      // v_log_f32 is a base-2 logarithm, v_sin_f32 / v_cos_f32 take their argument in revolutions -- sin(2*pi*u) is one
      // instruction, no range reduction (30 instructions for four normals where the library forms took 74)
      const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u3 = ((float)(o[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u4 = ((float)(o[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u) = sqrt(-2 ln2 log2 u)
      const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u3));
      c0 = __builtin_amdgcn_fmed3f(ra * __builtin_amdgcn_cosf(u2), -3.0f, 3.0f);
      c1 = __builtin_amdgcn_fmed3f(ra * __builtin_amdgcn_sinf(u2), -3.0f, 3.0f);
      c2 = __builtin_amdgcn_fmed3f(rb * __builtin_amdgcn_cosf(u4), -3.0f, 3.0f);
      c3 = __builtin_amdgcn_fmed3f(rb * __builtin_amdgcn_sinf(u4), -3.0f, 3.0f);
      cblk = blk;
    }
    // the draw's normal out of the block's four, branch-free: two sign-extended bit fields of the index as masks and three
    // bit-field inserts (left as a chain of ?: the compiler makes nested divergent branches of it: three compares, four
    // conditional moves and a dozen exec-mask instructions per draw)
    // (as inline assembly: from the C form the compiler builds ten and / or / sub instructions and a select)
    int m0, m1;
    float lo, hi, g;
    asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m0) : "v"(idx));                       // bit 0 of the index, sign-extended: 0 | ~0
    asm("v_bfe_i32 %0, %1, 1, 1" : "=v"(m1) : "v"(idx));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(m0), "v"(c1), "v"(c0));     // (m0 & c1) | (~m0 & c0)
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(m0), "v"(c3), "v"(c2));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(g) : "v"(m1), "v"(hi), "v"(lo));
    return (R)g;
  }
  // uniform [0,1) and start index for the counter modes
  FPX_DEV unsigned int bits(unsigned int stream) const {
    unsigned int o[4];
    philox4x32(pid, step, stream, 0x554e4946u, k0, k1, o);
    return o[0];
  }
  FPX_DEV int start_index(unsigned int stream) const {
    // int(ran3*real(maxrand-1))+1 with a counter-based uniform (advance.f90:153)
    return (int)(((unsigned long long)bits(stream) * (unsigned long long)(maxrand - 1)) >> 32) + 1;
  }
};

// ---------------------------------------------------------------------------
// one model level of the horizontally + time interpolated profiles
// (interpol_all.f90:135-240 loop body == interpol_misslev.f90:56-159)
// ---------------------------------------------------------------------------
template <typename R>
struct Level {
  R u, v, w, rho, rhograd, usig, vsig, wsig;
};

// horizontal weights and the corner indices of the cell (interpol_mod.f90:13-14)
template <typename R>
struct Cell {
  R p1, p2, p3, p4;
  int ix, jy, ixp, jyp;
};
// time weights dt1, dt2, dtt (interpol_mod.f90:13): the same for every particle of a step,
// so they live in scalar registers
template <typename R>
struct TimeW {
  R dt1, dt2, dtt;
};

template <typename R>
FPX_DEV TimeW<R> time_weights(const View<R> &V, int itime) {   // interpol_all.f90:69-71
  TimeW<R> W;
  W.dt1 = (R)(itime - V.memtime0);
  W.dt2 = (R)(V.memtime1 - itime);
  W.dtt = K(1.) / (W.dt1 + W.dt2);
  return W;
}

template <typename R>
FPX_DEV void cell_setup(Cell<R> &C, int ix, int jy, int ixp, int jyp, R xt, R yt) {
  // interpol_all.f90:57-64 (identical blocks open interpol_wind / interpol_wind_short)
  R ddx = xt - (R)ix, ddy = yt - (R)jy;
  R rddx = K(1.) - ddx, rddy = K(1.) - ddy;
  C.p1 = rddx * rddy; C.p2 = ddx * rddy; C.p3 = rddx * ddy; C.p4 = ddx * ddy;
  C.ix = ix; C.jy = jy; C.ixp = ixp; C.jyp = jyp;
}

// the field arrays of the grid a particle is on: mother grid (lat-lon or polar winds) or a nest
template <typename R>
struct Fld {
  int nx;
  const R *w3, *r2, *sfc, *hcell, *tropo, *vdep;
  const R *w3t0, *w3t1, *r2t0;   // time-blended packs of the mother grid (View), NULL elsewhere
};
template <typename R>
FPX_DEV Fld<R> fld_of(const View<R> &V, int ngrid) {
  Fld<R> F;
  if (ngrid > 0) {
    const int l = ngrid - 1;
    const NestDesc<R> &N = V.nest[l];
    F.nx = N.nx; F.w3 = N.w3; F.r2 = N.r2; F.sfc = N.sfc; F.hcell = N.hcell; F.tropo = N.tropo; F.vdep = N.vdep;
    F.w3t0 = nullptr; F.w3t1 = nullptr; F.r2t0 = nullptr;
  } else {
    F.nx = V.nx; F.w3 = ngrid < 0 ? V.w3pol : V.w3; F.r2 = V.r2; F.sfc = V.sfc; F.hcell = V.hcell; F.tropo = V.tropo; F.vdep = V.vdep;
    F.w3t0 = ngrid < 0 ? nullptr : V.w3t0; F.w3t1 = ngrid < 0 ? nullptr : V.w3t1; F.r2t0 = ngrid < 0 ? nullptr : V.r2t0;
  }
  return F;
}

// element offsets of the four corner columns (ix,jy) (ixp,jy) (ix,jyp) (ixp,jyp)
template <typename R>
struct Cols {
  long long c00, c10, c01, c11;
};
template <typename R>
FPX_DEV Cols<R> cols_of(int nx, const Cell<R> &C) {
  Cols<R> Q;
  Q.c00 = (long long)C.jy * nx + C.ix;
  Q.c10 = (long long)C.jy * nx + C.ixp;
  Q.c01 = (long long)C.jyp * nx + C.ix;
  Q.c11 = (long long)C.jyp * nx + C.ixp;
  return Q;
}

// load (a,b,c) of one corner/level/slot from a [..][nz][2][3] array
template <typename R>
FPX_DEV void ld3(const R *base, long long col, int nz, int n, int slot, R &a, R &b, R &c) {
  const R *p = base + ((col * nz + (n - 1)) * 2 + slot) * 3;
  a = p[0]; b = p[1]; c = p[2];
}

template <typename R, bool WITH_WIND, bool WITH_RHO, bool WITH_SIG>
FPX_DEV void level_profile(const View<R> &V, const Fld<R> &F, const Cell<R> &C, const TimeW<R> &W, int n, Level<R> &L) {
  const R eps = K(1.0e-30);
#ifdef FPX_EXP_FAKE_GATHER   // timing experiment only: every lane reads the same column (no memory latency), results are wrong
  Cols<R> Q = cols_of(F.nx, C);
  Q.c00 = 0; Q.c10 = 1; Q.c01 = 2; Q.c11 = 3; n = 1 + (n & 1);
#else
  const Cols<R> Q = cols_of(F.nx, C);
#endif
  const R *w3 = F.w3;
  R y1[2], y2[2], y3[2], rho1[2], rhograd1[2];
  R usl = 0, vsl = 0, wsl = 0, usq = 0, vsq = 0, wsq = 0;
#pragma unroll
  for (int m = 0; m < 2; m++) {
    int slot = m == 0 ? V.m1 : V.m2;
    if (WITH_WIND || WITH_SIG) {
      R u00, v00, w00, u10, v10, w10, u01, v01, w01, u11, v11, w11;
      ld3(w3, Q.c00, V.nz, n, slot, u00, v00, w00);
      ld3(w3, Q.c10, V.nz, n, slot, u10, v10, w10);
      ld3(w3, Q.c01, V.nz, n, slot, u01, v01, w01);
      ld3(w3, Q.c11, V.nz, n, slot, u11, v11, w11);
      if (WITH_WIND) {
        y1[m] = C.p1 * u00 + C.p2 * u10 + C.p3 * u01 + C.p4 * u11;
        y2[m] = C.p1 * v00 + C.p2 * v10 + C.p3 * v01 + C.p4 * v11;
        y3[m] = C.p1 * w00 + C.p2 * w10 + C.p3 * w01 + C.p4 * w11;
      }
      if (WITH_SIG) {
        usl = usl + u00 + u10 + u01 + u11;
        vsl = vsl + v00 + v10 + v01 + v11;
        wsl = wsl + w00 + w10 + w01 + w11;
        // fused multiply-adds spelled out (the variance usq - usl^2/8 cancels to rounding noise in smooth wind:
        // which product the compiler fuses must not depend on the kernel variant)
        usq = m_fma(u11, u11, m_fma(u01, u01, m_fma(u10, u10, m_fma(u00, u00, usq))));
        vsq = m_fma(v11, v11, m_fma(v01, v01, m_fma(v10, v10, m_fma(v00, v00, vsq))));
        wsq = m_fma(w11, w11, m_fma(w01, w01, m_fma(w10, w10, m_fma(w00, w00, wsq))));
      }
    }
    if (WITH_RHO) {
      const R *q00 = F.r2 + ((Q.c00 * V.nz + (n - 1)) * 2 + slot) * 2;
      const R *q10 = F.r2 + ((Q.c10 * V.nz + (n - 1)) * 2 + slot) * 2;
      const R *q01 = F.r2 + ((Q.c01 * V.nz + (n - 1)) * 2 + slot) * 2;
      const R *q11 = F.r2 + ((Q.c11 * V.nz + (n - 1)) * 2 + slot) * 2;
      rho1[m] = C.p1 * q00[0] + C.p2 * q10[0] + C.p3 * q01[0] + C.p4 * q11[0];
      rhograd1[m] = C.p1 * q00[1] + C.p2 * q10[1] + C.p3 * q01[1] + C.p4 * q11[1];
    }
  }
  if (WITH_WIND) {
    L.u = (y1[0] * W.dt2 + y1[1] * W.dt1) * W.dtt;
    L.v = (y2[0] * W.dt2 + y2[1] * W.dt1) * W.dtt;
    L.w = (y3[0] * W.dt2 + y3[1] * W.dt1) * W.dtt;
  }
  if (WITH_RHO) {
    L.rho = (rho1[0] * W.dt2 + rho1[1] * W.dt1) * W.dtt;
    L.rhograd = (rhograd1[0] * W.dt2 + rhograd1[1] * W.dt1) * W.dtt;
  }
  if (WITH_SIG) {   // 8-point standard deviation, interpol_all.f90:218-238
    R xaux = usq - usl * usl / K(8.);
    L.usig = xaux < eps ? K(0.) : m_sqrt(xaux / K(7.));
    xaux = vsq - vsl * vsl / K(8.);
    L.vsig = xaux < eps ? K(0.) : m_sqrt(xaux / K(7.));
    xaux = wsq - wsl * wsl / K(8.);
    L.wsig = xaux < eps ? K(0.) : m_sqrt(xaux / K(7.));
  }
}

// level below zt: first i in 2..nz with height(i) > zt gives indz = i-1
// (the linear scans of interpol_all.f90:118-125 etc.; bisection finds the same
// index because height is strictly increasing).  hgt points to LDS.
template <typename R>
FPX_DEV int find_level(const R *hgt, int nz, R zt) {
  int lo = 2, hi = nz;             // answer i in [2, nz]; if none qualifies clamp to nz
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (hgt[mid - 1] > zt) hi = mid; else lo = mid + 1;
  }
  return lo - 1;
}

// ---------------------------------------------------------------------------
// Hanna turbulence state (hanna_mod.f90:5-6)
// ---------------------------------------------------------------------------
template <typename R>
struct Turb {
  R ust, wst, ol, h, zeta, sigu, sigv, tlu, tlv, tlw, sigw, dsigwdz, dsigw2dz;
  R isigw;   // 1/sigw, kept by hanna_short for the Langevin step that follows (PBL loop only)
};

// Where exp and the absolute-accuracy logarithm come from: the polynomial / global-table forms (k_prep, k_pbl_finish: not
// VALU-bound), or the Langevin kernel's stash with its LDS copies of the tables (Stash::expt, Stash::logabs).
struct PlainMath {
  FPX_DEV double expt(double x) const { return m_expp(x); }
  FPX_DEV double logabs(double x) const { return m_log_abs(x); }
  FPX_DEV float expt(float x) const { return m_expp(x); }
  FPX_DEV float logabs(float x) const { return m_logp(x); }
};

// zeta**0.66666 and max(zeta,1.e-3)**(-.33333) of hanna.f90:67-70 / hanna_short.f90:60-63.
// fp64: zeta**(-.33333) = zeta**(-1/3) * exp(+(1/3 - .33333)*log zeta) and zeta**0.66666 = zeta * zeta**(-.33333) * exp(-1.e-5*log zeta)
// (0.66666 = 1 - 0.33333 - 1.e-5): the cube root by Newton steps from an f32 seed, the two tiny exponents through
// their cubic Taylor polynomials (|d| < 3e-4: remainder < 1e-15) with a logarithm that is only good to 1e-13 ABSOLUTE
// (m_log_abs) -- a relative 1e-18 in the factors.  39 instructions where one m_logp and one m_expp took 68.
template <typename R, typename MS>
FPX_DEV void zeta_powers(R zeta, const MS &M, R &z23, R &zm13) {
  if (sizeof(R) == 8) {
    // 0 and what the f32 seed cannot hold: the seed and the logarithm see 1e-37 instead (one v_max_f64, no selects); z23 below
    // is then zeta * 2e12 < 2e-25 for such a zeta (exactly 0 for zeta = 0) -- under the rounding of the ust**2 term (ust >= 1e-4)
    // of sigw either way
    const R zc = m_max(zeta, K(1.e-37));
    const R lz = M.logabs(zc);
    const R r13 = m_rcbrt(zc);
    const R d1 = ((K(1.) / K(3.)) - K(.33333)) * lz;
    const R e13 = r13 * (K(1.) + d1 * (K(1.) + d1 * (K(0.5) + d1 * K(0.16666666666666666))));
    const R d = ((K(1.) - K(.33333)) - K(0.66666)) * lz;       // 1.e-5 * log(zeta)
    const R corr = K(1.) - d * (K(1.) - d * (K(0.5) - d * K(0.16666666666666666)));
    z23 = zeta * e13 * corr;
    zm13 = zeta > K(1.e-3) ? e13 : K(9.9997697441416293);   // (1.e-3)**(-.33333)
  } else {   // reference typing: x**y as exp(y*log x) in f32 (a few ulp from powf, at a third of its cost)
    const R lz = m_logp(zeta);
    z23 = zeta > K(0.) ? m_expp(K(0.66666) * lz) : K(0.);
    zm13 = m_expp(K(-.33333) * (zeta > K(1.e-3) ? lz : m_logp(K(1.e-3))));
  }
}
// x**0.33333 for x >= 12 (hanna.f90:60: sigu = ust*(12 - 0.5*h/ol)**0.33333): x * (x**(-1/3))**2 * exp(-(1/3 - .33333)*log x)
template <typename MS>
FPX_DEV double m_pow13(double x, const MS &M) {
  if (__builtin_expect(!(x < 1.0e37), 0)) return m_powr(x, 0.33333);
  const double r = m_rcbrt(x);
  const double d = ((1.0 / 3.0) - .33333) * M.logabs(x);
  return (x * (r * r)) * (1.0 - d * (1.0 - d * (0.5 - d * 0.16666666666666666)));
}
template <typename MS>
FPX_DEV float m_pow13(float x, const MS &) { return m_powr(x, 0.33333f); }

template <typename R, typename MS>
FPX_DEV R tlw_unstable(const Turb<R> &T, R z, const MS &M) {   // hanna.f90:78-84
  if (z < m_abs(T.ol)) return K(0.1) * z * m_rcp(T.sigw * (K(0.55) - K(0.38) * m_abs(z * m_rcp(T.ol))));
  if (T.zeta < K(0.1)) return K(0.59) * z * m_rcp(T.sigw);
  return K(0.15) * T.h * m_rcp(T.sigw) * (K(1.) - M.expt(K(-5) * T.zeta));
}

template <typename R, typename MS>
FPX_DEV void sigw_unstable(Turb<R> &T, const MS &M) {   // hanna.f90:67-70 == hanna_short.f90:60-63
  R z23, zm13;
  zeta_powers(T.zeta, M, z23, zm13);
  T.sigw = m_sqrtp(K(1.2) * (T.wst * T.wst) * (K(1.) - K(.9) * T.zeta) * z23 + (K(1.8) - K(1.4) * T.zeta) * (T.ust * T.ust)) + K(1.e-2);
  T.dsigwdz = K(0.5) * m_rcp(T.sigw * T.h) * (K(-1.4) * (T.ust * T.ust) + (T.wst * T.wst) * (K(0.8) * zm13 - K(1.8) * z23));
}

template <typename R, typename MS = PlainMath>
FPX_DEV void hanna(Turb<R> &T, R z, const MS &M = MS()) {   // hanna.f90:41-106
  if (T.h / m_abs(T.ol) < K(1.)) {
    T.ust = m_max(K(1.e-4), T.ust);
    R corr = z * m_rcp(T.ust);
    T.sigu = K(1.e-2) + K(2.0) * T.ust * M.expt(K(-3.e-4) * corr);
    T.sigw = K(1.3) * T.ust * M.expt(K(-2.e-4) * corr);
    T.dsigwdz = K(-2.e-4) * T.sigw;
    T.sigw = T.sigw + K(1.e-2);
    T.sigv = T.sigw;
    T.tlu = K(0.5) * z * m_rcp(T.sigw * (K(1.) + K(1.5e-3) * corr));
    T.tlv = T.tlu;
    T.tlw = T.tlu;
  } else if (T.ol < K(0.)) {
    T.sigu = K(1.e-2) + T.ust * m_pow13(K(12) - K(0.5) * T.h * m_rcp(T.ol), M);
    T.sigv = T.sigu;
    sigw_unstable(T, M);
    T.tlu = K(0.15) * T.h * m_rcp(T.sigu);
    T.tlv = T.tlu;
    T.tlw = tlw_unstable(T, z, M);
  } else {
    T.sigu = K(1.e-2) + K(2.) * T.ust * (K(1.) - T.zeta);
    T.sigv = K(1.e-2) + K(1.3) * T.ust * (K(1.) - T.zeta);
    T.sigw = T.sigv;
    T.dsigwdz = K(-1.3) * T.ust * m_rcp(T.h);
    T.tlu = K(0.15) * T.h * m_rcp(T.sigu) * m_sqrtp(T.zeta);
    T.tlv = K(0.467) * T.tlu;
    T.tlw = K(0.1) * T.h * m_rcp(T.sigw) * m_pow08(T.zeta);
  }
  T.tlu = m_max(K(10.), T.tlu);
  T.tlv = m_max(K(10.), T.tlv);
  T.tlw = m_max(K(30.), T.tlw);
  if (T.dsigwdz == K(0.)) T.dsigwdz = K(1.e-10);
}

template <typename R>
FPX_DEV void hanna1(Turb<R> &T, R z) {   // hanna1.f90:41-129
  if (T.h / m_abs(T.ol) < K(1.)) {
    T.ust = m_max(K(1.e-4), T.ust);
    T.sigu = K(2.0) * T.ust * m_exp(K(-3.e-4) * z / T.ust);
    T.sigu = m_max(T.sigu, K(1.e-5));
    T.sigv = K(1.3) * T.ust * m_exp(K(-2.e-4) * z / T.ust);
    T.sigv = m_max(T.sigv, K(1.e-5));
    T.sigw = T.sigv;
    T.dsigw2dz = K(-6.76e-4) * T.ust * m_exp(K(-4.e-4) * z / T.ust);
    T.tlu = K(0.5) * z / T.sigw / (K(1.) + K(1.5e-3) * z / T.ust);
    T.tlv = T.tlu;
    T.tlw = T.tlu;
  } else if (T.ol < K(0.)) {
    T.sigu = T.ust * m_powr(K(12) - K(0.5) * T.h / T.ol, K(0.33333));
    T.sigu = m_max(T.sigu, K(1.e-6));
    T.sigv = T.sigu;
    if (T.zeta < K(0.03)) {
      T.sigw = K(0.96) * T.wst * m_powr(K(3) * T.zeta - T.ol / T.h, K(0.33333));
      T.dsigw2dz = K(1.8432) * T.wst * T.wst / T.h * m_powr(K(3) * T.zeta - T.ol / T.h, K(-0.33333));
    } else if (T.zeta < K(0.4)) {
      R s1 = K(0.96) * m_powr(K(3) * T.zeta - T.ol / T.h, K(0.33333));
      R s2 = K(0.763) * m_powr(T.zeta, K(0.175));
      if (s1 < s2) {
        T.sigw = T.wst * s1;
        T.dsigw2dz = K(1.8432) * T.wst * T.wst / T.h * m_powr(K(3) * T.zeta - T.ol / T.h, K(-0.33333));
      } else {
        T.sigw = T.wst * s2;
        T.dsigw2dz = K(0.203759) * T.wst * T.wst / T.h * m_powr(T.zeta, K(-0.65));
      }
    } else if (T.zeta < K(0.96)) {
      T.sigw = K(0.722) * T.wst * m_powr(K(1) - T.zeta, K(0.207));
      T.dsigw2dz = K(-.215812) * T.wst * T.wst / T.h * m_powr(K(1) - T.zeta, K(-0.586));
    } else if (T.zeta < K(1.00)) {
      T.sigw = K(0.37) * T.wst;
      T.dsigw2dz = K(0.);
    }  // zeta >= 1: the reference leaves sigw/dsigw2dz at whatever the module held; here: the thread's previous value
    T.sigw = m_max(T.sigw, K(1.e-6));
    T.tlu = K(0.15) * T.h / T.sigu;
    T.tlv = T.tlu;
    T.tlw = tlw_unstable(T, z, PlainMath());
  } else {
    T.sigu = K(2.) * T.ust * (K(1.) - T.zeta);
    T.sigv = K(1.3) * T.ust * (K(1.) - T.zeta);
    T.sigu = m_max(T.sigu, K(1.e-6));
    T.sigv = m_max(T.sigv, K(1.e-6));
    T.sigw = T.sigv;
    T.dsigw2dz = K(3.38) * T.ust * T.ust * (T.zeta - K(1.)) / T.h;
    T.tlu = K(0.15) * T.h / T.sigu * m_sqrt(T.zeta);
    T.tlv = K(0.467) * T.tlu;
    T.tlw = K(0.1) * T.h / T.sigw * m_powr(T.zeta, K(0.8));
  }
  T.tlu = m_max(K(10.), T.tlu);
  T.tlv = m_max(K(10.), T.tlv);
  T.tlw = m_max(K(30.), T.tlw);
}

// ---------------------------------------------------------------------------
// The PBL loop kernel keeps the part of a lane's state that is touched once per PASS, not
// once per fine sub-step, in LDS instead of registers: the two cached profile levels, the
// horizontal weights of the cell, the displacement sums, the grid-scale wind, the horizontal
// turbulent velocities, and the invariants the fine loop reads once per sub-step (surface-layer
// scales, density and its gradient): 25 values of R per lane, which leaves room for three blocks
// of 256 threads per CU (3 x (25 x 2 KB + height column) <= 160 KB).  The fine loop (cbl/hanna_short, ~120 live
// registers of its own) then fits the 256-VGPR budget of two waves per SIMD without spilling
// to scratch memory -- scratch spills of a persistent kernel are HBM traffic, LDS is not.
// Layout: slot-major [S_COUNT][block], one column per lane; the accesses are volatile so that
// the compiler does not forward the values through registers across the loop.
// ---------------------------------------------------------------------------
enum StashSlot {
  S_U, S_V, S_W,                                                                // interpol_mod u, v, w of the current pass (advance.f90:342-346)
  S_DDX, S_DDY,                                                                 // position inside the cell (interpol_all.f90:57-58); p1..p4 follow from it
  S_DX, S_DY, S_DAW, S_DCW,                                                     // dxsave, dysave, dawsave, dcwsave
  S_UP, S_VP,                                                                   // turbulent velocities along/across wind
  S_UST, S_WST, S_OL, S_TRANS,                                                  // hanna_mod ust, wst, ol; wst^3 * the transition of cbl.f90:79-81
  S_RHOAUX,                                                                     // per-pass invariant of the fine loop: rhograd/rhoa
  S_NPASS,                                                                      // passes the lane has run for its particle in this launch (time slices, k_pbl_loop)
  S_COUNT_LEAN,                                                                 // the gas kernels (LEAN) stop here
  S_TDEP = S_COUNT_LEAN,                                                        // aerosol kernels: sum of |dt| over the passes that ended below 2*href (advance.f90:582-599)
  S_SETCELL,                                                                    // aerosol kernels: column (njy*nx + nix) of get_settling -- so that xt, yt need not stay in registers
  S_SET_NUM, S_SET_DQ6, S_SET_V0,                                               // ... and its species constants 4*ga*dquer/1.e6*density*cunningham (0: no settling), dquer/1.e6, vsetaver
  S_COUNT_AERO64,                                                               // the fp64 aerosol kernels stop here (22 slots = 45 KB per block: three blocks per CU; with the five below it would be two)
  S_RT_TAG = S_COUNT_AERO64, S_RT_RHO1, S_RT_TT1, S_RT_RHO2, S_RT_TT2,          // f32 aerosol kernels: rho, tt of get_settling.f90:83-84 for the level pair S_RT_TAG (0: none) of the lane's column
  S_COUNT
};
// stash slots of a Langevin kernel instance (the host sizes the block's LDS with the same rule)
// (a one-launch gas kernel -- SUSP = false -- never touches S_NPASS, the last of its slots)
template <typename R> constexpr int stash_slots(bool lean, bool susp = true) { return lean ? (susp ? (int)S_COUNT_LEAN : (int)S_NPASS) : (sizeof(R) == 8 ? (int)S_COUNT_AERO64 : (int)S_COUNT); }
constexpr int kStashStride = 256;   // threads per block of the loop kernel
template <typename R>
struct Stash {
  // an LDS (address space 3) pointer, so that the accesses are ds_read/ds_write and not flat memory operations
  typedef __attribute__((address_space(3))) volatile R *lds_ptr;
  lds_ptr p;       // &lds[0][threadIdx.x]
  lds_tab_ptr tab; // the block's copy of kLogTab | kExpTab (fp64 build only)
  FPX_DEV double logabs(double x) const { return m_log_abs(x, tab); }
  FPX_DEV double expt(double x) const { return m_exp_tab(x, tab + kLdsExpTabAt); }
  FPX_DEV float logabs(float x) const { return m_logp(x); }
  FPX_DEV float expt(float x) const { return m_expp(x); }
  FPX_DEV R get(int k) const { return p[k * kStashStride]; }
  FPX_DEV void put(int k, R v) const { p[k * kStashStride] = v; }
  FPX_DEV void add(int k, R v) const { p[k * kStashStride] = p[k * kStashStride] + v; }
};

// hanna_short runs once per fine sub-step with the same h, ol, ust: the stability regime and
// the reciprocals of the step-invariant divisors are taken once per pass
template <typename R>
struct HsInv {
  R ih, iaux;    // 1/h; 1/ust (neutral) or 1/ol (unstable)
  int regime;    // 0 neutral (hanna_short.f90:46-52), 1 unstable (:57-72), 2 stable (:77-81)
};
template <typename R>
FPX_DEV HsInv<R> hanna_short_prepare(Turb<R> &T) {
  HsInv<R> I;
  I.ih = m_rcp(T.h);
  if (T.h / m_abs(T.ol) < K(1.)) {
    I.regime = 0;
    T.ust = m_max(K(1.e-4), T.ust);
    I.iaux = m_rcp(T.ust);
  } else if (T.ol < K(0.)) {
    I.regime = 1;
    I.iaux = m_rcp(T.ol);
  } else {
    I.regime = 2;
    I.iaux = K(0.);
  }
  return I;
}

// ST: where ust, wst, ol are read from (the LDS stash of the loop kernel)
template <typename R, typename ST>
FPX_DEV void hanna_short(Turb<R> &T, R z, const HsInv<R> &I, const ST &S) {   // hanna_short.f90:41-92
  if (I.regime == 0) {
    const R corr = z * I.iaux;
    T.sigw = K(1.3) * S.expt(K(-2.e-4) * corr);
    T.dsigwdz = K(-2.e-4) * T.sigw;
    T.sigw = T.sigw * S.get(S_UST) + K(1.e-2);
    const R qn = K(1.) + K(1.5e-3) * corr;
    const R in2 = m_rcp(T.sigw * qn);
    T.isigw = in2 * qn;
    T.tlw = K(0.5) * z * in2;
  } else if (I.regime == 1) {
    R z23, zm13;
    zeta_powers(T.zeta, S, z23, zm13);
    const R ust = S.get(S_UST), wst = S.get(S_WST);
    const R ust2 = ust * ust, wst2 = wst * wst;
    T.sigw = m_sqrtp(K(1.2) * wst2 * (K(1.) - K(.9) * T.zeta) * z23 + (K(1.8) - K(1.4) * T.zeta) * ust2) + K(1.e-2);
    // tlw (hanna.f90:78-84) and dsigwdz share the reciprocal of sigw
    const bool low = z < m_abs(S.get(S_OL));
    const R q = low ? K(0.55) - K(0.38) * m_abs(z * I.iaux) : K(1.);
    const R i2 = m_rcp(T.sigw * q);
    const R isig = i2 * q;
    T.isigw = isig;
    T.dsigwdz = K(0.5) * isig * I.ih * (K(-1.4) * ust2 + wst2 * (K(0.8) * zm13 - K(1.8) * z23));
    // (kept as branches: computing the third arm's exponential for every lane and selecting measured 0.2 % slower)
    if (low) T.tlw = K(0.1) * z * i2;
    else if (T.zeta < K(0.1)) T.tlw = K(0.59) * z * isig;
    else T.tlw = K(0.15) * T.h * isig * (K(1.) - S.expt(K(-5) * T.zeta));
  } else {
    const R ust = S.get(S_UST);
    T.sigw = K(1.e-2) + K(1.3) * ust * (K(1.) - T.zeta);
    T.dsigwdz = K(-1.3) * ust * I.ih;
    T.isigw = m_rcp(T.sigw);
    T.tlw = K(0.1) * T.h * T.isigw * m_pow08(T.zeta);
  }
  T.tlu = m_max(K(10.), T.tlu);
  T.tlv = m_max(K(10.), T.tlv);
  T.tlw = m_max(K(30.), T.tlw);
  // Never true in practice (every regime's dsigwdz carries ust >= 1e-4 or sigw > 0).  fp64: a branch that is not taken costs less
  // than the compare, two moves and two conditional moves of the select form in every sub-step (-0.45 % of the kernel); f32:
  // the select is one conditional move and measured 0.8 % faster than the branch
  if (sizeof(R) == 8) { if (__builtin_expect(T.dsigwdz == K(0.), 0)) { asm volatile("" ::: "memory"); T.dsigwdz = K(1.e-10); } }
  else if (T.dsigwdz == K(0.)) T.dsigwdz = K(1.e-10);
}

// ---------------------------------------------------------------------------
// skewed convective-boundary-layer scheme: cbl.f90
// ---------------------------------------------------------------------------
#define FPX_PI_PAR K(3.14159265)   // par_mod.f90:59

template <typename R>
FPX_DEV R cuberoot(R x) { return m_sign(m_pow(m_abs(x), K(0.333333333)), x); }   // cbl.f90:220-234

template <typename R>
FPX_DEV R cbl_transition(R h, R ol) {   // cbl.f90:79-81
  R transition = K(1.);
  if (-h / ol < K(15)) transition = m_sin(((-h / ol + K(10.)) / K(10.)) * FPX_PI_PAR) / K(2.) + K(0.5);
  return transition;
}

// cbl.f90:70-210 -> drift ath, diffusion bth, blow-up flag.
// Same mathematics as the reference, re-derived for a VALU-bound fp64 kernel (results differ by rounding only).
// With f = fluarw, s = skew, a1 = 1+f^2, a3 = 3+f^2, x = xluarw, r = rluarw, al = aluarw, bl = bluarw, sw = sigmaw:
//   w2**0.5 = sw (sw > 0), so s = w3/sw^3, ds = dw3/sw^3 - 3 s dsw/sw, dradw2 = dsigmawdz;
//   r = x^2 identically (cbl.f90:122-123: (1+f^2)^3 s^2/((3+f^2)^2 f^2) against (1+f^2)^1.5 s/((3+f^2) f)), hence
//     dr = 2 x dx, and with 4 + r - x^2 = 4 the derivative :144-146 collapses to dal = -2 dx (4+r)^-1.5;
//   al*bl = (1 - x^2/(4+r))/4 = 1/(4+r), so (bl/(al a1))**0.5 = bl*rs with rs = (4+r)^0.5 * a1^-0.5:
//     sigmawa = sw*rs*bl, sigmawb = sw*rs*al, 1/sigmawa = (a1^0.5 (4+r)^0.5 / sw) * al, 1/sigmawb = (..) * bl
//     -- two coupled sqrt/rsqrt pairs (a1 and 4+r) give every root and reciprocal root of :148-167;
//   the derivative of the first quotient, dbl*t1 - bl*(dal*a1 + al*ffd) over t1^2 with dbl = -dal and al+bl = 1,
//     is -(dal*a1 + al*bl*ffd)*bl^2*rs^4, so 0.5/qa05 times it = -0.5*bl*rs^3*(dal*a1 + al*bl*ffd);
//     for the second quotient +0.5*al*rs^3*(dal*a1 - al*bl*ffd);
//   sigmawa*dwa - wa*dsigmawa = dfluarw*sigmawa^2 (dwa = df*sigmawa + f*dsigmawa, wa = f*sigmawa), so the last bracket
//     of :193-194 is wold*dfluarw (and -wold*dfluarw in :198-199);
//   alfa = 2 w2/(C0 tlw) and bth = sqrt(C0 alfa) = sw*sqrt(2/tlw) from one rsqrt(tlw);
//   the two cube roots of cbl.f90:115-121 from one x**(-1/3) (m_cuberoot_parts).
// What is left is 2 reciprocals, 1 sqrt, 2 coupled sqrt+rsqrt, 1 rsqrt, 2 exp and 2 erf per call
// (the straightforward form has 20 divisions, 7 square roots, 1 log and 4 exp).
// `wt` = wst^3 * transition (cbl.f90:79-81,103-104): depends on h, ol, wst only and is passed in.
// ST: the source of the lookup tables of exp (the loop kernel's stash)
template <typename R, typename ST>
FPX_DEV void cbl(const ST &S, int ldirect, R wp, R zp, R wt /* wst^3 * transition */, R ih /* 1/h */, R rhoaux /* rhograd/rhoa */, R sigmaw, R irw /* 1/sigmaw */, R dsigmawdz, R tlw,
                 R &ath, R &bth, int &flagrein) {
  const R usurad2 = K(0.7071067812), usurad2p = K(0.3989422804), C0 = K(3), costluar4 = K(0.66667), eps = K(0.000001);
  const R timedir = (R)ldirect;
  const R z = zp * ih;
  const R w2 = sigmaw * sigmaw;
  const R rtl = m_rsqrt(tlw);                        // tlw >= 30 (hanna_short.f90:91)
  const R alfa = (K(2.) / C0) * w2 * (rtl * rtl);
  const R wold = timedir * wp;
  const R omz = K(1.) - z;
  const R omz05 = m_sqrtp(omz), omz15 = omz * omz05;
  const R w3 = (K(1.2) * z * omz15 + eps) * wt;
  const R dw3 = (K(1.2) * (omz15 - z * K(1.5) * omz05)) * (wt * ih);
  const R irw2 = irw * irw, irw3 = irw2 * irw;
  const R skew = w3 * irw3;
  const R dskew = irw3 * dw3 - K(3.) * skew * dsigmawdz * irw;   // (dw3*w2**1.5 - w3*1.5*w2**0.5*dw2)/w2**3 with w2 = sigmaw^2
  R fluarw = K(0.), dfluarw = K(0.), xluarw = K(0.), dxluarw = K(0.);
  if (skew != K(0)) {
    R croot, icroot2;
    m_cuberoot_parts(m_abs(skew), croot, icroot2);
    fluarw = costluar4 * m_sign(croot, skew);                         // costluar4*cuberoot(skew)
    dfluarw = costluar4 * (K(1.) / K(3.)) * icroot2 * dskew;          // cuberoot(skew**(-2.))
  }
  const R fluarw2 = fluarw * fluarw;
  const R a1 = K(1.) + fluarw2, a3 = K(3.) + fluarw2;
  const R ffd = K(2.) * fluarw * dfluarw;
  R a105, ia105;
  m_sqrt_rsqrt(a1, a105, ia105);
  if (skew != K(0)) {
    const R a115 = a1 * a105;
    const R ix = m_rcp(a3 * fluarw);
    xluarw = a115 * skew * ix;
    dxluarw = ((K(1.5) * a105 * ffd * skew + a115 * dskew) - xluarw * (K(3.) * dfluarw * a1)) * ix;
  }
  const R r4 = m_fma(xluarw, xluarw, K(4.));         // 4 + rluarw
  R r405, ir405;
  m_sqrt_rsqrt(r4, r405, ir405);
  const R aluarw = K(0.5) * (K(1.) - xluarw * ir405);
  const R bluarw = K(1.) - aluarw;
  const R ir4 = ir405 * ir405;                       // = aluarw*bluarw
  const R daluarw = K(-2.) * dxluarw * (ir405 * ir4);
  const R dbluarw = -daluarw;
  const R rs = r405 * ia105, rs3 = rs * (rs * rs);
  const R qa05 = bluarw * rs, qb05 = aluarw * rs;
  const R sigmawa = sigmaw * qa05, sigmawb = sigmaw * qb05;
  const R iv = irw * (a105 * r405);                  // 1/sigmawa = 1/(sigmaw*rs*bl) = (a1^0.5 (4+r)^-0.5/sigmaw) * al*(4+r) = iv*al
  const R isa = iv * aluarw, isb = iv * bluarw;
  const R da1 = daluarw * a1, abf = ir4 * ffd, hs3 = K(0.5) * sigmaw * rs3;
  const R dsigmawa = dsigmawdz * qa05 - hs3 * bluarw * (da1 + abf);
  const R dsigmawb = dsigmawdz * qb05 + hs3 * aluarw * (da1 - abf);
  const R wa = fluarw * sigmawa, wb = fluarw * sigmawb;
  const R dwa = dfluarw * sigmawa + fluarw * dsigmawa;
  const R dwb = dfluarw * sigmawb + fluarw * dsigmawb;
  const R deltawa = wold - wa, deltawb = wold + wb;
  const R wold2 = wold * wold;
  const R isa2 = isa * isa, isb2 = isb * isb;
  const R da = deltawa * isa, db = deltawb * isb;
  if (m_abs(da) > K(6.) && m_abs(db) > K(6.)) flagrein = 1;   // abs(deltawa) > 6*sigmawa .and. abs(deltawb) > 6*sigmawb
  const R da2 = da * da, db2 = db * db;
  const R ea = S.expt(-(K(0.5) * da2)), eb = S.expt(-(K(0.5) * db2));
  const R pa = (usurad2p * isa) * ea;
  const R pb = (usurad2p * isb) * eb;
  const R aperfa = da * usurad2;
  const R aperfb = db * usurad2;
  // The air density multiplies every term of ptot, Q and Phi (cbl.f90:175-205) and cancels in
  // ath = (-(C0/2)*alfa*Q + Phi)/ptot; what is left of it is rx = rhograd/rhoa.
  const R rx = rhoaux;
  const R apa = aluarw * pa, bpb = bluarw * pb;
  const R ptot = apa + bpb;
  const R Ta = aluarw * (dwa + wa * rx) + wa * daluarw;
  const R Tb = bluarw * (dwb + wb * rx) + wb * dbluarw;
  const R wdf = wold * dfluarw;
  const R Ua = sigmawa * (aluarw * (dsigmawa * (wold2 * isa2 + K(1.)) + wdf) + sigmawa * (daluarw + rx * aluarw));
  const R Ub = sigmawb * (bluarw * (dsigmawb * (wold2 * isb2 + K(1.)) - wdf) + sigmawb * (dbluarw + rx * bluarw));
  // exp(-aperf^2) from exp(-d^2/2): aperf = d*usurad2 and usurad2^2 - 0.5 = 5.9e-12 (the reference's 10-digit 1/sqrt(2))
  const R cu = usurad2 * usurad2 - K(0.5);
  R erfa, erfb;
  m_erf_e2(aperfa, ea - ea * (da2 * cu), aperfb, eb - eb * (db2 * cu), erfa, erfb);
  const R Phi = K(0.5) * (Tb * erfb - Ta * erfa) + Ua * pa + Ub * pb;
  const R Q = timedir * ((da * isa) * apa + (db * isb) * bpb);
  ath = m_rcp(ptot) * (-(C0 / K(2.)) * alfa * Q + Phi);
  bth = (sigmaw * K(1.4142135623730951)) * rtl;     // sqrt(C0*alfa) = sigmaw*sqrt(2/tlw)
}

// bi-Gaussian pdf parameters shared by re_initialize_particle.f90:47-70 and initialize_cbl_vel.f90:46-73
template <typename R>
FPX_DEV void cbl_pdf(R zp, R wst, R h, R sigmaw, R ol, R &aluarw, R &sigmawa, R &sigmawb, R &wa, R &wb) {
  // x**0.5 -> sqrt, x**1.5 -> x*sqrt(x), x**2., x**3. -> products (within an ulp of pow, and a tenth of its code:
  // this cold path sits inside the Langevin kernel)
  const R costluar4 = K(0.66667), eps = K(0.000001);
  R z = zp / h;
  R transition = cbl_transition(h, ol);
  R w2 = sigmaw * sigmaw;
  const R omz = K(1.) - z;
  R w3 = ((K(1.2) * z * (omz * m_sqrt(omz)) + eps) * (wst * wst * wst)) * transition;
  R radw2 = m_sqrt(w2);
  R skew = w3 / (w2 * radw2);
  R skew2 = skew * skew;
  R fluarw = costluar4 * m_powr(skew, K(0.333333333333333));
  R fluarw2 = fluarw * fluarw;
  const R a1 = K(1.) + fluarw2, a3 = K(3.) + fluarw2;
  R rluarw = (a1 * a1 * a1) * skew2 / ((a3 * a3) * fluarw2);
  R xluarw = m_sqrt(rluarw);
  aluarw = K(0.5) * (K(1.) - xluarw / m_sqrt(K(4.) + rluarw));
  R bluarw = K(1.) - aluarw;
  sigmawa = radw2 * m_sqrt(bluarw / (aluarw * a1));
  sigmawb = radw2 * m_sqrt(aluarw / (bluarw * a1));
  wa = fluarw * sigmawa;
  wb = fluarw * sigmawb;
}

// re_initialize_particle.f90:44-90; bounded re-draw loops (the reference's are unbounded)
template <typename R, typename RNG>
FPX_DEV void re_initialize_particle(int ldirect, const RNG &G, R zp, R wst, R h, R sigmaw, R &wp, int &nrand, R ol) {
  R aluarw, sigmawa, sigmawb, wa, wb;
  nrand = nrand + 1;
  R dcas1 = G.at(nrand);
  R timedir = (R)ldirect;
  cbl_pdf(zp, wst, h, sigmaw, ol, aluarw, sigmawa, sigmawb, wa, wb);
  R sg = m_sign(K(1.), wp) * timedir;
  if (sg > 0) {
    for (int it = 0; it < 10000; it++) {
      wp = dcas1 * sigmawa + wa;
      if (wp < 0) { nrand = nrand + 1; dcas1 = G.at(nrand); continue; }
      break;
    }
    wp = wp * timedir;
  } else if (sg < 0) {
    for (int it = 0; it < 10000; it++) {
      wp = dcas1 * sigmawb - wb;
      if (wp > 0) { nrand = nrand + 1; dcas1 = G.at(nrand); continue; }
      break;
    }
    wp = wp * timedir;
  }
}

// windalign.f90:36-54
template <typename R>
FPX_DEV void windalign(R u, R v, R ffap, R ffcp, R &ux, R &vy) {
  const R eps = K(1.e-30);
  if (ffap == K(0.) && ffcp == K(0.)) { ux = K(0.); vy = K(0.); return; }   // every particle that never was in the PBL this step
  R ffinv = K(1.) / m_max(m_sqrt(u * u + v * v), eps);
  R sinphi = v * ffinv;
  R vy1 = sinphi * ffap;
  R cosphi = u * ffinv;
  R ux1 = cosphi * ffap;
  R ux2 = -sinphi * ffcp;
  R vy2 = cosphi * ffcp;
  ux = ux1 + ux2;
  vy = vy1 + vy2;
}

// ---------------------------------------------------------------------------
// polar stereographic map subset: cmapf_mod.f90
// ---------------------------------------------------------------------------
#define CM_REARTH K(6371.2)
#define CM_ALMST1 K(.9999999)
#define CM_PI K(3.14159265358979)
#define CM_RADPDG (CM_PI / K(180.))
#define CM_DGPRAD (K(180.) / CM_PI)

template <typename R>
FPX_HD R cspanf(R value, R begin, R end) {   // cmapf_mod.f90:494-524
  R first = m_min(begin, end), last = m_max(begin, end);
  R val = m_fmod(value - first, last - first);
  return val <= K(0.) ? val + last : val + first;
}

template <typename R>
FPX_HD R cgszll(const R *s, R xlat) {   // cmapf_mod.f90:190-238
  double slat, ymerc, efact;
  if (xlat > K(89.985)) {
    if (s[0] > K(0.9999)) return K(2.) * s[6];
    efact = (double)m_cos(CM_RADPDG * xlat);
    if (efact <= 0.) return K(0.);
    ymerc = -log(efact / (double)(K(1.) + m_sin(CM_RADPDG * xlat)));
  } else if (xlat < K(-89.985)) {
    if (s[0] < K(-0.9999)) return K(2.) * s[6];
    efact = (double)m_cos(CM_RADPDG * xlat);
    if (efact <= 0.) return K(0.);
    ymerc = log(efact / (double)(K(1.) - m_sin(CM_RADPDG * xlat)));
  } else {
    slat = (double)m_sin(CM_RADPDG * xlat);
    ymerc = log((1. + slat) / (1. - slat)) / 2.;
  }
  return (R)((double)(s[6] * m_cos(CM_RADPDG * xlat)) * exp((double)s[0] * ymerc));
}

template <typename R>
FPX_HD void cnllxy(const R *s, R xlat, R xlong, R &xi, R &eta) {   // cmapf_mod.f90:310-365
  double gamma = (double)s[0];
  double dlat = (double)xlat;
  double dlong = (double)cspanf<R>(xlong - s[1], K(-180.), K(180.));
  dlong = dlong * (double)CM_RADPDG;
  R gdlong = (R)(gamma * dlong);
  R sndgam, csdgam, rhog1;
  if (m_abs(gdlong) < K(.01)) {
    gdlong = gdlong * gdlong;
    sndgam = (R)(dlong * (double)(K(1.) - K(1.) / K(6.) * gdlong * (K(1.) - K(1.) / K(20.) * gdlong * (K(1.) - K(1.) / K(42.) * gdlong))));
    csdgam = (R)(dlong * dlong * (double)K(.5) * (double)(K(1.) - K(1.) / K(12.) * gdlong * (K(1.) - K(1.) / K(30.) * gdlong * (K(1.) - K(1.) / K(56.) * gdlong))));
  } else {
    sndgam = (R)((double)m_sin(gdlong) / gamma);
    csdgam = (R)((double)(K(1.) - m_cos(gdlong)) / gamma / gamma);
  }
  double slat = sin((double)CM_RADPDG * dlat);
  if (slat >= (double)CM_ALMST1 || slat <= -(double)CM_ALMST1) {
    eta = K(1.) / s[0];
    xi = K(0.);
    return;
  }
  double mercy = .5 * log((1. + slat) / (1. - slat));
  double gmercy = gamma * mercy;
  if (fabs(gmercy) < (double)K(.001)) {
    rhog1 = (R)(mercy * (1. - .5 * gmercy * (1. - (double)(K(1.) / K(3.)) * gmercy * (1. - (double)(K(1.) / K(4.)) * gmercy))));
  } else {
    rhog1 = (R)((1. - exp(-gmercy)) / gamma);
  }
  eta = (R)((double)rhog1 + (1. - gamma * (double)rhog1) * gamma * (double)csdgam);
  xi = (R)((1. - gamma * (double)rhog1) * (double)sndgam);
}

template <typename R>
FPX_HD void cll2xy(const R *s, R xlat, R xlong, R &x, R &y) {   // cmapf_mod.f90:295-308
  R xi, eta;
  cnllxy(s, xlat, xlong, xi, eta);
  x = s[2] + CM_REARTH / s[6] * (xi * s[4] + eta * s[5]);
  y = s[3] + CM_REARTH / s[6] * (eta * s[4] - xi * s[5]);
}

template <typename R>
FPX_DEV void cnxyll(const R *s, double xi, double eta, R &xlat, R &xlong) {   // cmapf_mod.f90:367-425
  double gamma = (double)s[0], temp, ymerc, along;
  double arg2 = 2. * eta - gamma * (xi * xi + eta * eta);
  double arg1 = gamma * arg2;
  if (fabs(arg1) < (double)K(.01)) {
    temp = (arg1 / (2. - arg1)) * (arg1 / (2. - arg1));
    ymerc = arg2 / (2. - arg1) * (1. + temp * ((double)(K(1.) / K(3.)) + temp * ((double)(K(1.) / K(5.)) + temp * ((double)(K(1.) / K(7.))))));
  } else {
    ymerc = -log(1. - arg1) / 2. / gamma;
  }
  temp = exp(-fabs(ymerc));
  double a = atan2((1. - temp) * (1. + temp), 2. * temp);
  xlat = (R)(ymerc < 0 ? -fabs(a) : fabs(a));
  double gxi = gamma * xi, cgeta = 1. - gamma * eta;
  if (fabs(gxi) < (double)K(.01) * cgeta) {
    temp = (gxi / cgeta) * (gxi / cgeta);
    along = xi / cgeta * (1. - temp * ((double)(K(1.) / K(3.)) - temp * ((double)(K(1.) / K(5.)) - temp * ((double)(K(1.) / K(7.))))));
  } else {
    along = atan2(gxi, cgeta) / gamma;
  }
  xlong = (R)((double)s[1] + (double)CM_DGPRAD * along);
  xlat = xlat * CM_DGPRAD;
}

template <typename R>
FPX_DEV void cxy2ll(const R *s, R x, R y, R &xlat, R &xlong) {   // cmapf_mod.f90:526-543
  double xi0 = (double)((x - s[2]) * s[6] / CM_REARTH);
  double eta0 = (double)((y - s[3]) * s[6] / CM_REARTH);
  double xi = xi0 * (double)s[4] - eta0 * (double)s[5];
  double eta = eta0 * (double)s[4] + xi0 * (double)s[5];
  cnxyll(s, xi, eta, xlat, xlong);
  xlong = cspanf<R>(xlong, K(-180.), K(180.));
}

// Host-side map set-up (what gridcheck_ecmwf.f90:341-366 does through cmapf_mod): only used
// by hosts that do not have the reference's own northpolemap/southpolemap at hand.
template <typename R>
inline void stlmbr(R *s, R tnglat, R xlong) {   // cmapf_mod.f90:780-814
  R xi, eta;
  s[0] = m_sin(CM_RADPDG * tnglat);
  s[1] = cspanf<R>(xlong, K(-180.), K(180.));
  s[2] = K(0.); s[3] = K(0.); s[4] = K(1.); s[5] = K(0.);
  s[6] = CM_REARTH;
  cnllxy(s, K(89.), xlong, xi, eta);
  s[7] = K(2.) * eta - s[0] * eta * eta;
  cnllxy(s, K(-89.), xlong, xi, eta);
  s[8] = K(2.) * eta - s[0] * eta * eta;
}
template <typename R>
inline void stcm2p(R *s, R x1, R y1, R xlat1, R xlong1, R x2, R y2, R xlat2, R xlong2) {   // cmapf_mod.f90:603-633
  R x1a, y1a, x2a, y2a;
  for (int k = 2; k < 6; k++) s[k] = K(0.);
  s[4] = K(1.);
  s[6] = K(1.);
  cll2xy(s, xlat1, xlong1, x1a, y1a);
  cll2xy(s, xlat2, xlong2, x2a, y2a);
  R den = m_sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
  R dena = m_sqrt((x1a - x2a) * (x1a - x2a) + (y1a - y2a) * (y1a - y2a));
  s[4] = ((x1a - x2a) * (x1 - x2) + (y1a - y2a) * (y1 - y2)) / den / dena;
  s[5] = ((y1a - y2a) * (x1 - x2) - (x1a - x2a) * (y1 - y2)) / den / dena;
  s[6] = s[6] * dena / den;
  cll2xy(s, xlat1, xlong1, x1a, y1a);
  s[2] = s[2] + x1 - x1a;
  s[3] = s[3] + y1 - y1a;
}
// gridcheck_ecmwf.f90:341-366: maps for the two polar caps of a global lat/lon grid
template <typename R>
inline void polar_maps(R dy, R *north, R *south) {
  const R switchnorth = K(75.), switchsouth = K(-75.);   // par_mod.f90:123
  R sizesouth = K(6.) * (switchsouth + K(90.)) / dy;
  stlmbr(south, K(-90.), K(0.));
  stcm2p(south, K(0.), K(0.), switchsouth, K(0.), sizesouth, sizesouth, switchsouth, K(180.));
  R sizenorth = K(6.) * (K(90.) - switchnorth) / dy;
  stlmbr(north, K(90.), K(0.));
  stcm2p(north, K(0.), K(0.), switchnorth, K(0.), sizenorth, sizenorth, switchnorth, K(180.));
}

// ---------------------------------------------------------------------------
// gravitational settling: get_settling.f90:52-127, dynamic_viscosity.f90:7-17
// ---------------------------------------------------------------------------
// FAST (the Langevin kernel, once per pass): the level search starts from the pass's level (the particle rarely leaves it),
// the divisions and the root are the 1-2 ulp helpers of the fine loop, t**1.5 is t*sqrt(t), and everything that does not
// change inside the iteration is taken out of it (the same expressions in the same order: 4*ga*dquer/1.e6*density*cunningham
// and dquer/1.e6 are evaluated once instead of up to twenty times).
template <typename R>
FPX_DEV int settling_column(const View<R> &V, R xt, R yt) {   // nix = int(xt), njy = int(yt) of get_settling.f90:52-53 as one column index
  int nix = (int)xt, njy = (int)yt;
  nix = min(max(nix, 0), V.nx - 1);
  njy = min(max(njy, 0), V.ny - 1);
  return njy * V.nx + nix;
}
template <typename R> struct Stash;
// the per-species constants of the settling routine for a per-lane species index: one row of View::spec (device memory).
// An index outside the table reads species 0, as the select chain this replaces did.
template <typename R>
struct alignas(4 * sizeof(R)) SpecRow { R density, dquer, vsetaver, cunningham; };
template <typename R>
FPX_DEV SpecRow<R> spec_row(const View<R> &V, int nsp) {
  typedef const SpecRow<R> __attribute__((address_space(1))) *row_ptr;   // a global load, said so (as a flat one it tripped the compiler in the Langevin kernel, whose other tables are in LDS)
  const SpecRow<R> __attribute__((address_space(1))) &g = ((row_ptr)V.spec)[(unsigned int)nsp < (unsigned int)kMaxSpec ? nsp : 0];
  SpecRow<R> r;
  r.density = g.density; r.dquer = g.dquer; r.vsetaver = g.vsetaver; r.cunningham = g.cunningham;
  return r;
}
template <typename R>
struct SettleSpec { R num, dq6, vset; };   // the species constants of get_settling.f90:96-117 as the routine evaluates them
template <typename R>
FPX_DEV SettleSpec<R> settle_spec(const View<R> &V, int nsp) {
  const R ga = K(9.81);
  SettleSpec<R> c;
  const SpecRow<R> sp = spec_row(V, nsp);
  c.dq6 = sp.dquer / K(1.e6);
  c.vset = sp.vsetaver;
  c.num = K(4) * ga * sp.dquer / K(1.e6) * sp.density * sp.cunningham;
  return c;
}
template <typename R, bool FAST = false>
FPX_DEV R get_settling(const View<R> &V, const R *hgt, int column, R zt, int nsp, int level_hint = 0, const Stash<R> *ST = nullptr) {
#if defined(FPX_EXP_SETTLE) && FPX_EXP_SETTLE == 1   // timing experiment only (wrong results): no settling computation at all
  if (FAST) return spec_row(V, nsp).vsetaver;
#endif
  int indz = level_hint;
  // the level with height(indz) <= zt < height(indz+1), as the search of get_settling.f90:58-64 finds it
  if (!(FAST && indz >= 1 && indz <= V.nz - 1 && hgt[indz] > zt && (indz == 1 || !(hgt[indz - 1] > zt)))) indz = find_level(hgt, V.nz, zt);
  R rho1, tt1, rho2, tt2;
  const bool cached = sizeof(R) == 4 && ST != nullptr;      // (the fp64 kernels have no room for the five slots at three blocks per CU)
  if (cached && (int)ST->get(S_RT_TAG) == indz) {
    // the lane's column does not change inside the step and the particle leaves its level pair in a fifth of its passes:
    // the four values stay in the stash from pass to pass (before: one dependent 16-byte gather -- a cache line of its own
    // -- at the end of every pass: 56 GB of fetches per launch at 1e8 particles, 5 times the rest of the kernel's)
    rho1 = ST->get(S_RT_RHO1); tt1 = ST->get(S_RT_TT1); rho2 = ST->get(S_RT_RHO2); tt2 = ST->get(S_RT_TT2);
  } else {
    const R *q = V.rhott + ((long long)column * V.nz + (indz - 1)) * 2;
    rho1 = q[0]; tt1 = q[1]; rho2 = q[2]; tt2 = q[3];
    if (cached) {
      ST->put(S_RT_TAG, (R)indz); ST->put(S_RT_RHO1, rho1); ST->put(S_RT_TT1, tt1); ST->put(S_RT_RHO2, rho2); ST->put(S_RT_TT2, tt2);
    }
  }
  R dz = FAST ? m_rcp(hgt[indz] - hgt[indz - 1]) : K(1.) / (hgt[indz] - hgt[indz - 1]);
  R dz1 = (zt - hgt[indz - 1]) * dz;
  R dz2 = (hgt[indz] - zt) * dz;
  R temperature = dz2 * tt1 + dz1 * tt2;
  R airdens = dz2 * rho1 + dz1 * rho2;
  const R cc = K(120.), t_0 = K(291.15), eta_0 = K(1.827e-5);
  R dq6, vset, num;
  if (FAST && ST != nullptr) {   // taken once per particle at the lane's refill (a per-lane subscript into the species tables is a memory round trip)
    dq6 = ST->get(S_SET_DQ6); vset = ST->get(S_SET_V0); num = ST->get(S_SET_NUM);
  } else {
    const SettleSpec<R> c = settle_spec(V, nsp);
    dq6 = c.dq6; vset = c.vset; num = c.num;
  }
  if (FAST) {
    const R tr = temperature * m_rcp(t_0);
    const R vis_dyn = eta_0 * (t_0 + cc) * m_rcp(temperature + cc) * (tr * m_sqrtp(tr));   // dynamic_viscosity.f90:14
    const R ivis_kin = airdens * m_rcp(vis_dyn);
    R reynolds = dq6 * m_abs(vset) * ivis_kin;
    R settling_old = vset, settling = K(0.);
#if defined(FPX_EXP_SETTLE) && FPX_EXP_SETTLE == 2   // timing experiment only (wrong results): one iteration
    for (int i = 1; i <= 1; i++) {
#else
    for (int i = 1; i <= 20; i++) {
#endif
      R icd;      // 1 / c_d
      if (__builtin_expect(reynolds < K(1.917), 1)) icd = reynolds * (K(1.) / K(24.));
      else if (reynolds < K(500.)) icd = m_pow(reynolds, K(0.6)) * (K(1.) / K(18.5));
      else icd = K(1.) / K(0.44);
      settling = K(-1.) * m_sqrtp(num * icd * m_rcp(K(3.) * airdens));
      if (m_abs((settling - settling_old) * m_rcp(settling)) < K(0.01)) break;
      reynolds = dq6 * m_abs(settling) * ivis_kin;
      settling_old = settling;
    }
    return settling;
  }
  R vis_dyn = eta_0 * (t_0 + cc) / (temperature + cc) * m_pow(temperature / t_0, K(1.5));
  R vis_kin = vis_dyn / airdens;
  R reynolds = dq6 * m_abs(vset) / vis_kin;
  R settling_old = vset, settling = K(0.), c_d;
  for (int i = 1; i <= 20; i++) {
    if (reynolds < K(1.917)) c_d = K(24.) / reynolds;
    else if (reynolds < K(500.)) c_d = K(18.5) / m_pow(reynolds, K(0.6));
    else c_d = K(0.44);
    settling = K(-1.) * m_sqrt(num / (K(3.) * c_d * airdens));
    if (m_abs((settling - settling_old) / settling) < K(0.01)) break;
    reynolds = dq6 * m_abs(settling) / vis_kin;
    settling_old = settling;
  }
  return settling;
}

// 0-based release point of a particle, clamped into the tables (the reference would read outside them)
template <typename R>
FPX_DEV int release_index(const View<R> &V, int npoint) { return min(max(npoint, 1), max(V.numpoint, 1)) - 1; }

// species pick of advance.f90:518-524 for a particle of release point `npoint`: the first species with
// xmass(nrelpoint,nsp) > eps3, else nspec (the table is evaluated once on the host, in the host's real kind)
template <typename R>
FPX_DEV int settling_species(const View<R> &V, int npoint) {
  if (!V.lsettling || !V.rel_nsp) return 0;
  return (int)V.rel_nsp[release_index(V, npoint)];
}

// settling velocity of species nsp, the block repeated at advance.f90:518-531,686-699,893-906
template <typename R, bool FAST = false>
FPX_DEV R settling_velocity(const View<R> &V, const R *hgt, double xt, double yt, R zt, int nsp, int level_hint = 0) {
  if (V.mdomainfill != 0 || !V.lsettling) return K(0.);
  if (!(spec_row(V, nsp).density > K(0.))) return K(0.);
  return get_settling<R, FAST>(V, hgt, settling_column(V, (R)xt, (R)yt), zt, nsp, level_hint);
}

// ---------------------------------------------------------------------------
// the per-thread trajectory step
// ---------------------------------------------------------------------------
template <typename R, bool MOTHER = false> FPX_DEV int pick_grid(const View<R> &V, double xt, double yt);
template <typename R> FPX_DEV int pick_polar(const View<R> &V, double yt);

template <typename R>
struct PState {   // one particle in registers
  double xt, yt;
  R zt, up, vp, wp, usigold, vsigold, wsigold;
  int ldt;
  short icbt;
};

template <typename R>
FPX_DEV int pick_polar(const View<R> &V, double yt) {   // the polar part of advance.f90:161-164
  if (V.nglobal && yt > (double)V.switchnorthg) return -1;
  if (V.sglobal && yt < (double)V.switchsouthg) return -2;
  return 0;
}
// advance.f90:161-175: -1/-2 polar caps, j = highest-numbered nest containing the particle, 0 mother.
// MOTHER: the caller knows that the run has neither polar caps nor nests (compile-time ngrid = 0)
template <typename R, bool MOTHER>
FPX_DEV int pick_grid(const View<R> &V, double xt, double yt) {
  if (MOTHER) return 0;
  const int p = pick_polar(V, yt);
  if (p != 0) return p;
  for (int j = V.numbnests; j >= 1; j--) {
    const NestDesc<R> &N = V.nest[j - 1];
    if (xt > (double)(N.xl + V.eps) && xt < (double)(N.xr - V.eps) &&
        yt > (double)(N.yl + V.eps) && yt < (double)(N.yr - V.eps)) return j;
  }
  return 0;
}

// boundary conditions advance.f90:784-813 == :956-985
template <typename R>
FPX_DEV int boundary(const View<R> &V, const R *hgt, double &xt, double &yt, R &zt, R eps) {
  if (V.xglobal) {
    double xm = (double)(R)V.nxmin1;
    if (xt >= xm) xt = xt - xm;
    if (xt < 0.) xt = xt + xm;
    if (xt <= (double)eps) xt = (double)eps;
    if (fabs(xt - xm) <= (double)eps) xt = (double)((R)V.nxmin1 - eps);
    if (yt < 0.) {
      xt = d_modulo(xt * (double)V.dx + 180., 360.) / (double)V.dx;
      yt = -yt;
    } else if (yt > (double)(R)V.nymin1) {
      xt = d_modulo(xt * (double)V.dx + 180., 360.) / (double)V.dx;
      yt = (double)(K(2) * (R)V.nymin1) - yt;
    }
  }
  if (!(xt >= 0.) || !(xt < (double)(R)V.nxmin1) || !(yt >= 0.) || !(yt <= (double)(R)V.nymin1)) return 3;  // also catches NaN
  if (zt >= hgt[V.nz - 1]) zt = hgt[V.nz - 1] - K(100.) * eps;
  return 0;
}

// horizontal move by (du,dv) metres on grid `ngrid`: advance.f90:750-778 == :923-951
// The move on a polar stereographic map (advance.f90:754-775), OUT OF LINE: a sixth of the particles of a global run take it,
// but inlined its trigonometry (cll2xy, cgszll, cxy2ll) set the register budget of every kernel that moves particles -- the
// polar instances of k_prep / k_pbl_finish ran at two waves per SIMD instead of three.  The map records come from device
// memory (View::polemaps: north | south, 9 values each); the function call keeps its registers to itself.
struct XY { double x, y; };
template <typename R>
__device__ __attribute__((noinline)) XY polar_move(const R *__restrict__ map, R xlon0, R ylat0, R dx, R dy, double xt, double yt, R du, R dv, R fac) {
  R m[9];
#pragma unroll
  for (int i = 0; i < 9; i++) m[i] = map[i];
  R xlon = (R)((double)xlon0 + xt * (double)dx);
  R ylat = (R)((double)ylat0 + yt * (double)dy);
  R xpol, ypol;
  cll2xy(m, ylat, xlon, xpol, ypol);
  R gridsize = K(1000.) * cgszll(m, ylat);
  du = du / gridsize;
  dv = dv / gridsize;
  xpol = xpol + du * fac;
  ypol = ypol + dv * fac;
  cxy2ll(m, xpol, ypol, ylat, xlon);
  XY r;
  r.x = (double)((xlon - xlon0) / dx);
  r.y = (double)((ylat - ylat0) / dy);
  return r;
}
template <typename R, bool POLAR = true>
FPX_DEV void move_xy(const View<R> &V, int ngrid, double &xt, double &yt, R du, R dv, R fac) {
  const R pi180 = FPX_PI_PAR / K(180.);
  if (!POLAR || ngrid >= 0) {
    R cosfact = (R)((double)V.dxconst / m_coslat((yt * (double)V.dy + (double)V.ylat0) * (double)pi180));
    xt = xt + (double)(du * cosfact * fac);
    yt = yt + (double)(dv * V.dyconst * fac);
  } else {
#ifdef FPX_POLAR_INLINE
    const R *map = ngrid == -1 ? V.northpolemap : V.southpolemap;
    R xlon = (R)((double)V.xlon0 + xt * (double)V.dx);
    R ylat = (R)((double)V.ylat0 + yt * (double)V.dy);
    R xpol, ypol;
    cll2xy(map, ylat, xlon, xpol, ypol);
    R gridsize = K(1000.) * cgszll(map, ylat);
    du = du / gridsize;
    dv = dv / gridsize;
    xpol = xpol + du * fac;
    ypol = ypol + dv * fac;
    cxy2ll(map, xpol, ypol, ylat, xlon);
    xt = (double)((xlon - V.xlon0) / V.dx);
    yt = (double)((ylat - V.ylat0) / V.dy);
#else
    const XY r = polar_move<R>(V.polemaps + (ngrid == -1 ? 0 : 9), V.xlon0, V.ylat0, V.dx, V.dy, xt, yt, du, dv, fac);
    xt = r.x; yt = r.y;
#endif
  }
}

struct NoLate { FPX_DEV void operator()() const {} };

// wind at (cell, zt): interpol_wind.f90:75-214 (SIG) / interpol_wind_short.f90:67-140.
// The 48 values are taken corner column by corner column -- the 12 values of a column (2 levels x 2 slots x (u,v,w))
// are one contiguous run of the w3 pack -- and folded into the running sums as they arrive: 12 loaded values are live
// instead of 48 (k_prep's register budget).  The horizontal sums p1*y(ix,jy) + p2*y(ixp,jy) + p3*y(ix,jyp) + p4*y(ixp,jyp)
// keep the reference's left-to-right order; the 16-point sums of the standard deviation (:194-214) are taken in
// corner order instead of (slot, level) order: the same 16 terms, a different rounding.
#ifndef FPX_GATHER_DEPTH
#define FPX_GATHER_DEPTH 2
#endif
template <typename R, bool SIG, typename LATE = NoLate, int DEPTH = FPX_GATHER_DEPTH>
FPX_DEV void interp_wind(const View<R> &V, const R *hgt, const Fld<R> &F, const Cell<R> &C, const TimeW<R> &W, R zt,
                         R &u, R &v, R &w, R &usig, R &vsig, R &wsig, const LATE &late = LATE(), const R *blended = nullptr) {
  const R eps = K(1.0e-30);
  const R *w3 = F.w3;
  const int indz = find_level(hgt, V.nz, zt);
  if (!SIG && blended) {
    // wave-uniform: the pack already carries (y(m1)*dt2 + y(m2)*dt1)*dtt of this call's time (interpol_wind.f90:189-191 taken
    // before the horizontal and vertical sums instead of after: the same numbers up to rounding).  6 values per column
    // instead of 12; all four columns in one round trip.
    R x[4][6];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const long long col = (long long)((c & 2) ? C.jyp : C.jy) * F.nx + ((c & 1) ? C.ixp : C.ix);
      const R *p = blended + (col * V.nz + (indz - 1)) * 3;
#pragma unroll
      for (int i = 0; i < 6; i++) x[c][i] = p[i];
    }
    late();
    __builtin_amdgcn_sched_barrier(0);
    R a[2][3];
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int k = 0; k < 3; k++)
        a[n][k] = ((C.p1 * x[0][n * 3 + k] + C.p2 * x[1][n * 3 + k]) + C.p3 * x[2][n * 3 + k]) + C.p4 * x[3][n * 3 + k];
    const R dz = K(1.) / (hgt[indz] - hgt[indz - 1]);
    const R dz1 = (zt - hgt[indz - 1]) * dz;
    const R dz2 = (hgt[indz] - zt) * dz;
    u = dz2 * a[0][0] + dz1 * a[1][0];
    v = dz2 * a[0][1] + dz1 * a[1][1];
    w = dz2 * a[0][2] + dz1 * a[1][2];
    return;
  }
  R au[2][2], av[2][2], aw[2][2];   // [physical slot][level]
  R usl = 0, vsl = 0, wsl = 0, usq = 0, vsq = 0, wsq = 0;
  // software pipeline over the four columns, DEPTH of them in flight (1: one dependent memory round trip
  // per column, 12 loaded values live; 2: two round trips, 24 live; 4: one round trip, 48 live)
  R x[4][12];
  auto issue = [&](int c) {
    const long long col = (long long)((c & 2) ? C.jyp : C.jy) * F.nx + ((c & 1) ? C.ixp : C.ix);
    const R *p = w3 + (col * V.nz + (indz - 1)) * 6;
#pragma unroll
    for (int i = 0; i < 12; i++) x[c][i] = p[i];
  };
#pragma unroll
  for (int c = 0; c < DEPTH && c < 4; c++) issue(c);
  if (DEPTH >= 4) late();
#pragma unroll
  for (int c = 0; c < 4; c++) {
    __builtin_amdgcn_sched_barrier(0);   // keeps the scheduler from issuing all 24 loads first
    const R pw = c == 0 ? C.p1 : c == 1 ? C.p2 : c == 2 ? C.p3 : C.p4;
#pragma unroll
    for (int n = 0; n < 2; n++) {
#pragma unroll
      for (int sl = 0; sl < 2; sl++) {
        const R uu = x[c][(n * 2 + sl) * 3 + 0], vv = x[c][(n * 2 + sl) * 3 + 1], ww = x[c][(n * 2 + sl) * 3 + 2];
        if (c == 0) { au[sl][n] = pw * uu; av[sl][n] = pw * vv; aw[sl][n] = pw * ww; }
        else { au[sl][n] = au[sl][n] + pw * uu; av[sl][n] = av[sl][n] + pw * vv; aw[sl][n] = aw[sl][n] + pw * ww; }
        if (SIG) {
          usl = usl + uu; vsl = vsl + vv; wsl = wsl + ww;
          usq = m_fma(uu, uu, usq); vsq = m_fma(vv, vv, vsq); wsq = m_fma(ww, ww, wsq);   // spelled out: usq - usl^2/16 cancels to rounding noise
        }
      }
    }
    if (c + DEPTH < 4) issue(c + DEPTH);
    if (c + DEPTH == 3) late();   // with the last column's loads: the caller's late loads share their round trip
  }
  const R dz = K(1.) / (hgt[indz] - hgt[indz - 1]);
  const R dz1 = (zt - hgt[indz - 1]) * dz;
  const R dz2 = (hgt[indz] - zt) * dz;
  R uh[2], vh[2], wh[2];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const bool hi = (m == 0 ? V.m1 : V.m2) != 0;       // wave-uniform
    uh[m] = dz2 * (hi ? au[1][0] : au[0][0]) + dz1 * (hi ? au[1][1] : au[0][1]);
    vh[m] = dz2 * (hi ? av[1][0] : av[0][0]) + dz1 * (hi ? av[1][1] : av[0][1]);
    wh[m] = dz2 * (hi ? aw[1][0] : aw[0][0]) + dz1 * (hi ? aw[1][1] : aw[0][1]);
  }
  u = (uh[0] * W.dt2 + uh[1] * W.dt1) * W.dtt;
  v = (vh[0] * W.dt2 + vh[1] * W.dt1) * W.dtt;
  w = (wh[0] * W.dt2 + wh[1] * W.dt1) * W.dtt;
  if (SIG) {   // 16-point sigma, interpol_wind.f90:194-214
    R xaux = usq - usl * usl / K(16.);
    usig = xaux < eps ? K(0.) : m_sqrt(xaux / K(15.));
    xaux = vsq - vsl * vsl / K(16.);
    vsig = xaux < eps ? K(0.) : m_sqrt(xaux / K(15.));
    xaux = wsq - wsl * wsl / K(16.);
    wsig = xaux < eps ? K(0.) : m_sqrt(xaux / K(15.));
  }
}

// ust, wst, ol at the cell: interpol_all.f90:80-107
template <typename R>
FPX_DEV void interp_surface(const View<R> &V, const Fld<R> &F, const Cell<R> &C, const TimeW<R> &W, Turb<R> &T) {
  const Cols<R> Q = cols_of(F.nx, C);
  R ust1[2], wst1[2], oli1[2];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    int slot = m == 0 ? V.m1 : V.m2;
    const R *a = F.sfc + (Q.c00 * 2 + slot) * 4, *b = F.sfc + (Q.c10 * 2 + slot) * 4;
    const R *c = F.sfc + (Q.c01 * 2 + slot) * 4, *d = F.sfc + (Q.c11 * 2 + slot) * 4;
    ust1[m] = C.p1 * a[0] + C.p2 * b[0] + C.p3 * c[0] + C.p4 * d[0];
    wst1[m] = C.p1 * a[1] + C.p2 * b[1] + C.p3 * c[1] + C.p4 * d[1];
    oli1[m] = C.p1 * a[2] + C.p2 * b[2] + C.p3 * c[2] + C.p4 * d[2];
  }
  T.ust = (ust1[0] * W.dt2 + ust1[1] * W.dt1) * W.dtt;
  T.wst = (wst1[0] * W.dt2 + wst1[1] * W.dt1) * W.dtt;
  R oliaux = (oli1[0] * W.dt2 + oli1[1] * W.dt1) * W.dtt;
  T.ol = oliaux != K(0.) ? K(1.) / oliaux : K(99999.);
}

// interpol_vdep.f90:39-54
template <typename R>
FPX_DEV R interp_vdep(const View<R> &V, const Fld<R> &F, const Cell<R> &C, const TimeW<R> &W, int ks) {
  const Cols<R> Q = cols_of(F.nx, C);
  R y[2];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    int slot = m == 0 ? V.m1 : V.m2;
    y[m] = C.p1 * F.vdep[(Q.c00 * 2 + slot) * V.nspec + ks] + C.p2 * F.vdep[(Q.c10 * 2 + slot) * V.nspec + ks] +
           C.p3 * F.vdep[(Q.c01 * 2 + slot) * V.nspec + ks] + C.p4 * F.vdep[(Q.c11 * 2 + slot) * V.nspec + ks];
  }
  return (y[0] * W.dt2 + y[1] * W.dt1) * W.dtt;
}

// Two-level profile cache of the PBL loop: the reference caches every PBL level it has touched
// (indzindicator, interpol_mod.f90:16); the values are pure functions of the level, so recomputing them
// gives the same numbers (fetch_levels_stash below).
// The 8-point sigmas (usigprof..) are only read when the interval ends (advance.f90:604-606): they are
// evaluated then, for the final level pair (level_pair_sigma).

// usig = 0.5*(usigprof(indzp)+usigprof(indz)) etc., advance.f90:604-606
template <typename R>
FPX_DEV void level_pair_sigma(const View<R> &V, const Fld<R> &F, const Cell<R> &C, const TimeW<R> &W, int indz, R &usig, R &vsig, R &wsig) {
  Level<R> lo, hi;
  level_profile<R, false, false, true>(V, F, C, W, indz, lo);
  level_profile<R, false, false, true>(V, F, C, W, indz + 1, hi);
  usig = K(0.5) * (hi.usig + lo.usig);
  vsig = K(0.5) * (hi.vsig + lo.vsig);
  wsig = K(0.5) * (hi.wsig + lo.wsig);
}

// initialize.f90:66-217.  Returns nothing; fills the turbulent state of a new particle.
template <typename R, typename RNG>
FPX_DEV void initialize_particle(const View<R> &V, const R *hgt, const RNG &G, int nrand, int itime,
                                 PState<R> &P, R cbl_dcas, R cbl_dcas1) {
  P.icbt = 1;
  int ix = (int)P.xt, jy = (int)P.yt;
  ix = min(max(ix, 0), V.nx - 2);           // guard only; in-domain particles are untouched
  jy = min(max(jy, 0), V.ny - 2);
  int ixp = ix + 1, jyp = jy + 1;
  Turb<R> T;
  T.sigw = K(0.); T.dsigw2dz = K(0.); T.dsigwdz = K(0.);
  T.h = V.hcell[(long long)jy * V.nx + ix];   // max of the 8 hmix values, initialize.f90:83-90
  T.zeta = P.zt / T.h;
  Cell<R> C;
  cell_setup(C, ix, jy, ixp, jyp, (R)P.xt, (R)P.yt);
  const TimeW<R> W = time_weights(V, itime);
  // The reference's initialize() reads the module variable ngrid left behind by the
  // previous particle's advance() (interpol_all.f90:144); a parallel engine has no
  // "previous particle", so the particle's own polar/lat-lon choice is used.
  // (both wind packs read before the choice, for the reason given at pick(): `pole ? V.w3pol : V.w3` becomes one load at a
  // selected address and the fp64 instances with a polar or nest table then kept the kernel argument in scratch, 816 B a lane)
  Fld<R> F = fld_of(V, 0);
  {
    const R *w3pol = V.w3pol;
    const bool pole = pick_polar(V, P.yt) != 0;
    F.w3 = pole ? w3pol : F.w3;
    F.w3t0 = pole ? nullptr : F.w3t0; F.w3t1 = pole ? nullptr : F.w3t1; F.r2t0 = pole ? nullptr : F.r2t0;
  }
  R usig, vsig, wsig;
  if (T.zeta <= K(1.)) {
    interp_surface(V, F, C, W, T);
    int indz = find_level(hgt, V.nz, P.zt);
    Level<R> lo, hi;
    level_profile<R, false, false, true>(V, F, C, W, indz, lo);
    level_profile<R, false, false, true>(V, F, C, W, indz + 1, hi);
    // (u,v,w of initialize.f90:116-118 are not used further)
    if (V.turbswitch) hanna(T, P.zt); else hanna1(T, P.zt);
    if (nrand + 2 > V.maxrand) nrand = 1;
    P.up = G.at(nrand) * T.sigu;
    P.vp = G.at(nrand + 1) * T.sigv;
    P.wp = G.at(nrand + 2);
    if (!V.turbswitch) {
      P.wp = P.wp * T.sigw;
    } else if (V.cblflag == 1) {
      if (-T.h / T.ol > K(5)) {   // initialize_cbl_vel.f90:46-83
        R aluarw, sigmawa, sigmawb, wa, wb;
        cbl_pdf(P.zt, T.wst, T.h, T.sigw, T.ol, aluarw, sigmawa, sigmawb, wa, wb);
        R timedir = (R)V.ldirect;
        if (cbl_dcas <= aluarw) P.wp = timedir * (cbl_dcas1 * sigmawa + wa);
        else P.wp = timedir * (cbl_dcas1 * sigmawb - wb);
      } else {
        P.wp = P.wp * T.sigw;
      }
    }
    if (V.turbswitch)
      P.ldt = (int)(m_min(m_min(m_min(T.tlw, T.h / m_max(K(2.) * m_abs(P.wp * T.sigw), K(1.e-5))), K(0.5) / m_abs(T.dsigwdz)), K(600.)) * V.ctl);
    else
      P.ldt = (int)(m_min(m_min(T.tlw, T.h / m_max(K(2.) * m_abs(P.wp), K(1.e-5))), K(600.)) * V.ctl);
    P.ldt = max(P.ldt, V.mintime);
    usig = (hi.usig + lo.usig) / K(2.);
    vsig = (hi.vsig + lo.vsig) / K(2.);
    wsig = (hi.wsig + lo.wsig) / K(2.);
  } else {
    R u, v, w;
    interp_wind<R, true>(V, hgt, F, C, W, P.zt, u, v, w, usig, vsig, wsig);
    P.ldt = abs(V.lsynctime);
    if (nrand + 1 > V.maxrand) nrand = 1;
    P.up = G.at(nrand) * K(0.3);
    P.vp = G.at(nrand + 1) * K(0.3);
    nrand = nrand + 2;
    P.wp = K(0.);
  }
  if (nrand + 2 > V.maxrand) nrand = 1;
  P.usigold = G.at(nrand) * usig;
  P.vsigold = G.at(nrand + 1) * vsig;
  P.wsigold = G.at(nrand + 2) * wsig;
}

// does initialize() take the CBL branch that consumes extra sequential draws?
// (initialize.f90:142-146) -- used only to order the TABLE_SEQ host stream
template <typename R>
FPX_DEV int initialize_needs_cbl_draws(const View<R> &V, const R *hgt, int itime, double xt, double yt, R zt) {
  if (!(V.cblflag == 1 && V.turbswitch)) return 0;
  int ix = (int)xt, jy = (int)yt;
  ix = min(max(ix, 0), V.nx - 2);
  jy = min(max(jy, 0), V.ny - 2);
  Turb<R> T;
  T.h = V.hcell[(long long)jy * V.nx + ix];
  if (!(zt / T.h <= K(1.))) return 0;
  Cell<R> C;
  cell_setup(C, ix, jy, ix + 1, jy + 1, (R)xt, (R)yt);
  interp_surface(V, fld_of(V, 0), C, time_weights(V, itime), T);
  return (-T.h / T.ol > K(5)) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// advance.f90:133-985, decomposed for the GPU:
//   adv_begin      :133-267  grid/cell choice, mixing height, PBL test
//   pbl_begin      :295-302  (first pass) surface-layer parameters at the cell
//   pbl_pass       :282-609  ONE pass of the Langevin loop (label 100 ... goto 100)
//   above_step     :629-708  the single step above the PBL (label 700)
//   adv_finish     :728-985  mesoscale term, wind alignment, move, boundary, Petterssen
// A thread may run these back to back (above-PBL particles) or, in the PBL
// kernel, interleave passes of different particles (lane refill).
// ---------------------------------------------------------------------------
template <typename R>
struct AdvCtx {                 // what advance() keeps between its labelled sections
  int ngrid, ix, jy, ixp, jyp;  // interpol_mod ix..jyp, ngrid
  R xr, yr;                     // position in the index space of that grid: real(xt),real(yt) or xtn,ytn
  R h;
  R dxsave, dysave, dawsave, dcwsave;
  R u, v, w;                    // interpol_mod u, v, w
  int itimec, nrand;
  int nsp;                      // species of the settling pick for this particle's release point (set by the caller)
  R tropop;                     // tropopause height at the nearest grid point (advance.f90:253,263), with TROPO only
};

enum { PBL_CONTINUE = 0, PBL_DONE = 1, PBL_ESCAPED = 2 };


// returns true when the particle starts inside the PBL (zeta <= 1, advance.f90:276)
// TROPO: also fetch the tropopause height of advance.f90:253,263 -- in the same memory round trip as the mixing height
template <typename R, bool MOTHER = false, bool TROPO = false>
FPX_DEV bool adv_begin(const View<R> &V, double xt, double yt, R zt, int itime, int nrand, AdvCtx<R> &A) {
  A.dxsave = K(0.); A.dysave = K(0.); A.dawsave = K(0.); A.dcwsave = K(0.);
  A.u = K(0.); A.v = K(0.); A.w = K(0.);
  A.itimec = itime;
  A.nrand = nrand;
  A.nsp = 0;
  A.ngrid = pick_grid<R, MOTHER>(V, xt, yt);
  int nyrows = V.ny, nxcols = V.nx;
  if (A.ngrid > 0) {   // advance.f90:191-197 nested grid coordinates
    const int l = A.ngrid - 1;
    const NestDesc<R> &N = V.nest[l];
    A.xr = (R)((xt - (double)N.xl) * (double)N.xres);
    A.yr = (R)((yt - (double)N.yl) * (double)N.yres);
    A.ix = (int)A.xr; A.jy = (int)A.yr;
    nyrows = V.nest[l].ny; nxcols = V.nest[l].nx;
  } else {
    A.xr = (R)xt; A.yr = (R)yt;
    A.ix = (int)xt; A.jy = (int)yt;
  }
  A.ixp = A.ix + 1; A.jyp = A.jy + 1;
  if (A.jyp >= nyrows) A.jyp = A.jyp - 1;   // advance.f90:228-231 (device rows are allocated ny, not nymax)
  if (A.ixp >= nxcols) A.ixp = nxcols - 1;  // guard for a non-cyclic domain edge
  {
    const Fld<R> F = fld_of(V, A.ngrid);
    A.h = F.hcell[(long long)A.jy * F.nx + A.ix];        // advance.f90:236-262, interpolhmix = .false.: the maximum over the cell's corners and both times
    if (__builtin_expect(V.interpolhmix != 0 && A.ngrid <= 0, 0)) {
      // interpolhmix = .true. (advance.f90:240-244,266): bilinear in the cell, linear in time; p1..p4 of advance.f90:209-216
      // (xt, yt are double precision there); the reference sets h1 on the mother grid only (nests: refused at fpx_nests_init)
      const R ddx = (R)(xt - (double)(R)A.ix), ddy = (R)(yt - (double)(R)A.jy);
      const R rddx = K(1.) - ddx, rddy = K(1.) - ddy;
      const R p1 = rddx * rddy, p2 = ddx * rddy, p3 = rddx * ddy, p4 = ddx * ddy;
      const TimeW<R> W = time_weights(V, itime);
      R h1[2];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int m = k == 0 ? V.m1 : V.m2;
        const long long c00 = (long long)A.jy * F.nx + A.ix, c10 = (long long)A.jy * F.nx + A.ixp;
        const long long c01 = (long long)A.jyp * F.nx + A.ix, c11 = (long long)A.jyp * F.nx + A.ixp;
        h1[k] = p1 * F.sfc[(c00 * 2 + m) * 4 + 3] + p2 * F.sfc[(c10 * 2 + m) * 4 + 3] + p3 * F.sfc[(c01 * 2 + m) * 4 + 3] + p4 * F.sfc[(c11 * 2 + m) * 4 + 3];
      }
      A.h = (h1[0] * W.dt2 + h1[1] * W.dt1) * W.dtt;
    }
    if (TROPO) {
      // tropopause(nix,njy,1,1) / tropopausen(nix,njy,1,1,ngrid), advance.f90:253,263 (literal slot 1)
      const int nix = A.ngrid > 0 ? (int)lround((double)A.xr) : (int)lround(xt);
      const int njy = A.ngrid > 0 ? (int)lround((double)A.yr) : (int)lround(yt);
      A.tropop = F.tropo[(long long)njy * F.nx + nix];
    }
  }
  return zt / A.h <= K(1.);
}

template <typename R>
struct PblCtx {                 // first-pass set-up handed from k_prep to the PBL loop
  Cell<R> C;
  R ust, wst, ol;               // hanna_mod ust, wst, ol
  R transition;                 // cbl.f90:79-81, constant during the step (depends on h/ol only)
};

template <typename R>
FPX_DEV void pbl_begin(const View<R> &V, double xt, double yt, const TimeW<R> &W, const AdvCtx<R> &A, PblCtx<R> &B) {
  cell_setup(B.C, A.ix, A.jy, A.ixp, A.jyp, A.xr, A.yr);   // interpol_all.f90:57-64 / interpol_all_nests.f90
  Turb<R> T;
  interp_surface(V, fld_of(V, A.ngrid), B.C, W, T);
  B.ust = T.ust; B.wst = T.wst; B.ol = T.ol;
  B.transition = V.cblflag == 1 ? cbl_transition(A.h, T.ol) : K(1.);
}

// One pass of the loop advance.f90:282-609.  DRYDEP: the time below 2*href is summed in the stash (S_TDEP).
// indz_last receives the level pair of this pass: when the pass ends the interval (PBL_DONE)
// the caller evaluates usig/vsig/wsig for it (advance.f90:604-606, level_pair_sigma).
// TSW / CBLF: -1 = read turbswitch / cblflag at run time, 0/1 = fixed at compile time
// (specialised hot kernels: fewer scalar registers, no dead branches).  SETTLE/DRYDEP false
// compile the aerosol paths out.
template <typename R>
struct LoopCtx {                // register-resident state of a lane across passes
  int ngrid, ix, jy, ixp, jyp;  // interpol_mod ix..jyp, ngrid
  R h;
  int itimec, nrand;
  int nsp;                      // species of the settling pick (aerosol kernels only)
};

// adv_begin for a particle whose grid and mixing height k_prep has already determined and handed over in its record (the
// refill of the Langevin kernel): no dependent memory access on the mother grid, one (the nest's descriptor) inside a nest.
template <typename R>
FPX_DEV void adv_begin_known(const View<R> &V, double xt, double yt, int ngrid, R h, LoopCtx<R> &L, R &ddx, R &ddy) {
  int nyrows = V.ny, nxcols = V.nx;
  R xr, yr;
  L.ngrid = ngrid;
  if (ngrid > 0) {   // advance.f90:191-197 nested grid coordinates
    const NestDesc<R> &N = V.nest[ngrid - 1];
    xr = (R)((xt - (double)N.xl) * (double)N.xres);
    yr = (R)((yt - (double)N.yl) * (double)N.yres);
    L.ix = (int)xr; L.jy = (int)yr;
    nyrows = N.ny; nxcols = N.nx;
  } else {
    xr = (R)xt; yr = (R)yt;
    L.ix = (int)xt; L.jy = (int)yt;
  }
  L.ixp = L.ix + 1; L.jyp = L.jy + 1;
  if (L.jyp >= nyrows) L.jyp = L.jyp - 1;   // advance.f90:228-231
  if (L.ixp >= nxcols) L.ixp = nxcols - 1;
  L.h = h;
  ddx = xr - (R)L.ix; ddy = yr - (R)L.jy;   // interpol_all.f90:57-58
}

template <typename R>
FPX_DEV Cell<R> stash_cell(const LoopCtx<R> &L, const Stash<R> &S) {
  Cell<R> C;
  const R ddx = S.get(S_DDX), ddy = S.get(S_DDY);          // as cell_setup, interpol_all.f90:59-64
  const R rddx = K(1.) - ddx, rddy = K(1.) - ddy;
  C.p1 = rddx * rddy; C.p2 = ddx * rddy; C.p3 = rddx * ddy; C.p4 = ddx * ddy;
  C.ix = L.ix; C.jy = L.jy; C.ixp = L.ixp; C.jyp = L.jyp;
  return C;
}

// The two profile levels around the particle, computed every pass and parked in the stash
// (interpol_all.f90:135-240 / interpol_misslev.f90).  The reference caches every level it has touched
// (indzindicator); a per-lane cache of the last level pair (re-using one level when the particle moved to the
// neighbouring pair) was built first and measured slower than recomputing: a lane crosses a level in ~20 % of
// its passes, so nearly every pass of a wave ran the miss path anyway, with a fifth of its lanes active.
template <typename R>
FPX_DEV void fetch_level_pair(const View<R> &V, const Fld<R> &F, const TimeW<R> &W, const LoopCtx<R> &L, const Stash<R> &S, int indz, R (&lv)[2][5] /* [lo|hi][u v w rho rhograd] */) {
  // Both levels of the pair at once: with z fastest in the packs the 2 levels x 2 slots of a corner column are ONE run of
  // 12 (u, v, w) and one of 8 (rho, drhodz) values, so a corner costs one address (32-bit cell index, one 64-bit scale per
  // pack) instead of eight; the horizontal sums p1*y(ix,jy) + p2*y(ixp,jy) + p3*y(ix,jyp) + p4*y(ixp,jyp) are taken corner
  // by corner in the reference's left-to-right order (interpol_all.f90:147-187), 20 running sums live.
  const R ddx = S.get(S_DDX), ddy = S.get(S_DDY);          // as cell_setup, interpol_all.f90:59-64
  const R rddx = K(1.) - ddx, rddy = K(1.) - ddy;
  typedef const R __attribute__((address_space(1))) *gptr;  // the packs live in device memory: global, not flat, loads
  if (F.w3t0 && F.r2t0) {
    // wave-uniform (large clouds on the mother grid): the packs blended in time once per step (View::w3t0, r2t0) -- 10 values
    // per corner instead of 20, no time interpolation per particle; the blend is taken before the horizontal sums instead of
    // after (interpol_all.f90:189-198): the same numbers up to rounding
#ifndef FPX_BLEND_CORNERS
#define FPX_BLEND_CORNERS 4   // all four corners in one round trip: 80 values in flight fit (161 VGPRs, no spill); 2: +0.8 % time
#endif
    constexpr int NB = FPX_BLEND_CORNERS;     // corners per memory round trip
    R acc[2][5];
#pragma unroll
    for (int cb = 0; cb < 4; cb += NB) {
      R y3[NB][6], y2[NB][4];
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const int c = cb + j;
        const int jyc = (c & 2) ? L.jyp : L.jy, ixc = (c & 1) ? L.ixp : L.ix;
        const unsigned int cell = (unsigned int)(jyc * F.nx + ixc) * (unsigned int)V.nz + (unsigned int)(indz - 1);
        const gptr p = (gptr)(F.w3t0 + (size_t)cell * 3);
        const gptr q = (gptr)(F.r2t0 + (size_t)cell * 2);
#pragma unroll
        for (int k = 0; k < 6; k++) y3[j][k] = p[k];
#pragma unroll
        for (int k = 0; k < 4; k++) y2[j][k] = q[k];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const int c = cb + j;
        const R pw = c == 0 ? rddx * rddy : c == 1 ? ddx * rddy : c == 2 ? rddx * ddy : ddx * ddy;
#pragma unroll
        for (int lev = 0; lev < 2; lev++) {
#pragma unroll
          for (int k = 0; k < 3; k++) acc[lev][k] = c == 0 ? pw * y3[j][lev * 3 + k] : m_fma(pw, y3[j][lev * 3 + k], acc[lev][k]);
#pragma unroll
          for (int k = 0; k < 2; k++) acc[lev][3 + k] = c == 0 ? pw * y2[j][lev * 2 + k] : m_fma(pw, y2[j][lev * 2 + k], acc[lev][3 + k]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int lev = 0; lev < 2; lev++)
#pragma unroll
      for (int k = 0; k < 5; k++) lv[lev][k] = acc[lev][k];
    return;
  }
  R a3[2][2][3], a2[2][2][2];                               // [level][physical slot][variable]
#ifndef FPX_FETCH_CORNERS
#define FPX_FETCH_CORNERS 1
#endif
  // NC corners per batch: the 20 values of each are requested together and used after ONE wait (left to itself the
  // compiler, short of registers, waited after every second load: 17 dependent memory round trips per pass)
  constexpr int NC = FPX_FETCH_CORNERS;
#pragma unroll
  for (int cb = 0; cb < 4; cb += NC) {
    R x3[NC][12], x2[NC][8];
#pragma unroll
    for (int j = 0; j < NC; j++) {
      const int c = cb + j;
      const int jyc = (c & 2) ? L.jyp : L.jy, ixc = (c & 1) ? L.ixp : L.ix;
      const unsigned int cell = (unsigned int)(jyc * F.nx + ixc) * (unsigned int)V.nz + (unsigned int)(indz - 1);   // < nx*ny*nz
      const gptr p = (gptr)(F.w3 + (size_t)cell * 6);
      const gptr q = (gptr)(F.r2 + (size_t)cell * 4);
#pragma unroll
      for (int k = 0; k < 12; k++) x3[j][k] = p[k];
#pragma unroll
      for (int k = 0; k < 8; k++) x2[j][k] = q[k];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NC; j++) {
      const int c = cb + j;
      const R pw = c == 0 ? rddx * rddy : c == 1 ? ddx * rddy : c == 2 ? rddx * ddy : ddx * ddy;
#pragma unroll
      for (int lev = 0; lev < 2; lev++) {
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const R x = x3[j][(lev * 2 + sl) * 3 + k];
            a3[lev][sl][k] = c == 0 ? pw * x : m_fma(pw, x, a3[lev][sl][k]);
          }
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const R x = x2[j][(lev * 2 + sl) * 2 + k];
            a2[lev][sl][k] = c == 0 ? pw * x : m_fma(pw, x, a2[lev][sl][k]);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");   // the next batch's loads stay behind this batch's sums (f32: 80 registers in flight otherwise)
  }
  const bool h1 = V.m1 != 0, h2 = V.m2 != 0;               // wave-uniform: physical slot of memind(1) / memind(2)
#pragma unroll
  for (int lev = 0; lev < 2; lev++) {
    R y[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const R y1 = k < 3 ? (h1 ? a3[lev][1][k] : a3[lev][0][k]) : (h1 ? a2[lev][1][k - 3] : a2[lev][0][k - 3]);
      const R y2 = k < 3 ? (h2 ? a3[lev][1][k] : a3[lev][0][k]) : (h2 ? a2[lev][1][k - 3] : a2[lev][0][k - 3]);
      y[k] = (y1 * W.dt2 + y2 * W.dt1) * W.dtt;           // interpol_all.f90:189-198
    }
#pragma unroll
    for (int k = 0; k < 5; k++) lv[lev][k] = y[k];
  }
}

template <int T>
FPX_DEV bool sw(int runtime) { return T < 0 ? runtime != 0 : T != 0; }

template <typename R, bool DRYDEP, bool SETTLE, int TSW, int CBLF, typename RNG>
FPX_DEV int pbl_pass(const View<R> &V, const R *hgt, const RNG &G, const TimeW<R> &W, int itime, double xt, double yt,
                     R &zt, R &wp, int &ldt, short &icbt, LoopCtx<R> &A, const Stash<R> &S,
                     int &indz_last, Stats *st) {
  const R eps = V.eps;
  const R eps2 = K(1.e-9);
  const R href = K(15.);            // par_mod.f90:76
  const R h = A.h;
  const Fld<R> F = fld_of(V, A.ngrid);
  int nrand = A.nrand;
  const bool turbswitch = sw<TSW>(V.turbswitch);
  const bool cblflag = sw<CBLF>(V.cblflag == 1);
  // turboff (com_mod.f90:778, .false. as shipped) exists in the general instance only: in the specialised kernels the two
  // selects per fine sub-step it compiles to cost 1 % of the kernel; a run with turboff takes the general instance (loop_table)
  const bool turboff = TSW < 0 && V.turboff != 0;
  Turb<R> T;
  T.ust = S.get(S_UST); T.wst = S.get(S_WST); T.ol = S.get(S_OL); T.h = h;
  T.sigw = K(0.); T.dsigw2dz = K(0.); T.dsigwdz = K(0.);   // only read if hanna1 meets zeta >= 1 (see hanna1)

  if (V.method == 1) {
    ldt = min(ldt, abs(V.lsynctime - A.itimec + itime));
    A.itimec = A.itimec + ldt * V.ldirect;
  } else {
    ldt = abs(V.lsynctime);
    A.itimec = itime + V.lsynctime;
  }
  const R dt = (R)ldt;
  T.zeta = zt / h;

  const int indz = find_level(hgt, V.nz, zt);
  const int indzp = indz + 1;
  indz_last = indz;
  R lv[2][5];
  fetch_level_pair(V, F, W, A, S, indz, lv);

  // advance.f90:342-350
  const R dz = m_rcp(hgt[indzp - 1] - hgt[indz - 1]);
  const R dz1 = (zt - hgt[indz - 1]) * dz;
  const R dz2 = (hgt[indzp - 1] - zt) * dz;
  {
    // grid-scale wind of this pass and its contribution to the displacement sums (advance.f90:539-540;
    // the sums do not depend on the fine loop, so they are taken here and stay out of registers)
    const R u = dz1 * lv[1][0] + dz2 * lv[0][0];
    const R v = dz1 * lv[1][1] + dz2 * lv[0][1];
    S.put(S_U, u); S.put(S_V, v);      // handed over with the particle when it leaves the loop
    S.put(S_W, dz1 * lv[1][2] + dz2 * lv[0][2]);
    S.add(S_DX, u * dt);
    S.add(S_DY, v * dt);
  }
  {
    const R rhoa = dz1 * lv[1][3] + dz2 * lv[0][3];
    const R rhograd = dz1 * lv[1][4] + dz2 * lv[0][4];
    S.put(S_RHOAUX, rhograd * m_rcp(rhoa));
  }

  if (turbswitch) hanna(T, zt, S); else hanna1(T, zt);
  S.put(S_UST, T.ust);   // hanna may floor ust at 1.e-4 (hanna.f90:43) and the module variable keeps it
  T.isigw = m_rcp(T.sigw);

  // counter mode: a pass starts on a block boundary of the generator, so that all lanes of a wave
  // renew their block in the same fine sub-step (the draws of a pass are indexed alike in every lane)
  if (RNG::kCounter) nrand = (nrand + 3) & ~3;
  // horizontal Langevin, advance.f90:371-384
  if (nrand + 1 > V.maxrand) nrand = 1;
  {
    const R g1 = G.at(nrand), g2 = G.at(nrand + 1);
    const R dttlu = dt * m_rcp(T.tlu), dttlv = dt * m_rcp(T.tlv);
    R up = S.get(S_UP), vp = S.get(S_VP);
    if (dttlu < K(.5)) {
      up = (K(1.) - dttlu) * up + g1 * T.sigu * m_sqrtp(K(2.) * dttlu);
    } else {
      R ru = S.expt(-dttlu);
      up = ru * up + g1 * T.sigu * m_sqrtp(K(1.) - ru * ru);
    }
    if (dttlv < K(.5)) {
      vp = (K(1.) - dttlv) * vp + g2 * T.sigv * m_sqrtp(K(2.) * dttlv);
    } else {
      R rv = S.expt(-dttlv);
      vp = rv * vp + g2 * T.sigv * m_sqrtp(K(1.) - rv * rv);
    }
    if (turboff) { up = K(0.); vp = K(0.); }   // advance.f90:464-467: zeroed in the fine loop, before :541-542 read them
    S.put(S_UP, up); S.put(S_VP, vp);
    S.add(S_DAW, up * dt);   // advance.f90:541-542
    S.add(S_DCW, vp * dt);
  }
  nrand = nrand + 2;

  if (nrand + V.ifine > V.maxrand) nrand = 1;
  const R dtf = dt * V.fine;
  const R dtftlw = dtf * m_rcp(T.tlw);
  const bool cbl_on = cblflag && (-h / T.ol > K(5));
  const R sqrt_dtf = m_sqrtp(dtf);
  const HsInv<R> HI = hanna_short_prepare(T);

  // vertical Langevin, ifine sub-steps, advance.f90:396-498
  FPX_LANES(st, 0);
  unsigned int flip = icbt < 0 ? 0x80000000u : 0u;   // icbt as a sign mask inside the loop (m_flip)
  for (int i = 1; i <= V.ifine; i++) {
    R delz;
    FPX_LANES(st, 1);
    if (turbswitch) {
      if (dtftlw < K(.5)) {
        if (cblflag) {
          if (cbl_on) {
            FPX_LANES(st, 2);
            int flagrein = 0;
            nrand = nrand + 1;
            R old_wp_buf = wp, ath, bth;
            cbl(S, V.ldirect, wp, zt, S.get(S_TRANS), HI.ih, S.get(S_RHOAUX), T.sigw, T.isigw, T.dsigwdz, T.tlw, ath, bth, flagrein);
            wp = m_flip(wp + ath * dtf + bth * G.at(nrand) * sqrt_dtf, flip);
            delz = wp * dtf;
            if (__builtin_expect(flagrein == 1, 0)) {
              re_initialize_particle(V.ldirect, G, zt, S.get(S_WST), h, T.sigw, old_wp_buf, nrand, S.get(S_OL));
              wp = old_wp_buf;
              delz = wp * dtf;
              atomicAdd(&st->nan_count, 1ull);
            }
          } else {
            FPX_LANES(st, 3);
            nrand = nrand + 1;
            R ath = -wp * m_rcp(T.tlw) + T.sigw * T.dsigwdz + wp * wp * T.isigw * T.dsigwdz + T.sigw * T.sigw * S.get(S_RHOAUX);
            R bth = T.sigw * G.at(nrand) * m_sqrtp(K(2.) * dtftlw);
            wp = m_flip(wp + ath * dtf + bth, flip);
            delz = wp * dtf;
            // del_test=(1.-wp)/wp is NaN exactly when wp is NaN or infinite (advance.f90:440-441)
            if (__builtin_expect(!isfinite(wp), 0)) {
              nrand = nrand + 1;
              wp = T.sigw * G.at(nrand);
              delz = wp * dtf;
              atomicAdd(&st->nan_count2, 1ull);
            }
          }
        } else {
          wp = m_flip((K(1.) - dtftlw) * wp + G.at(nrand + i) * m_sqrtp(K(2.) * dtftlw) + dtf * (T.dsigwdz + S.get(S_RHOAUX) * T.sigw), flip);
          delz = wp * T.sigw * dtf;
        }
      } else {
        FPX_LANES(st, 4);
        R rw = S.expt(-dtftlw);
        wp = m_flip(rw * wp + G.at(nrand + i) * m_sqrtp(K(1.) - rw * rw) + T.tlw * (K(1.) - rw) * (T.dsigwdz + S.get(S_RHOAUX) * T.sigw), flip);
        delz = wp * T.sigw * dtf;
      }
    } else {
      R rw = m_expp(-dtftlw);
      wp = m_flip(rw * wp + G.at(nrand + i) * m_sqrtp(K(1.) - rw * rw) * T.sigw + T.tlw * (K(1.) - rw) * (T.dsigw2dz + S.get(S_RHOAUX) * (T.sigw * T.sigw)), flip);
      delz = wp * dtf;
    }

    if (turboff) { wp = K(0.); delz = K(0.); }   // advance.f90:464-470 (turboff; the random numbers stay drawn)

    // reflection at the ground / mixing height, advance.f90:476-491
    if (__builtin_expect(m_abs(delz) > h, 0)) delz = m_fmod(delz, h);
    {
      // branch-free (three short arms as divergent branches cost more in exec-mask and branch instructions than in work):
      // the same sums in the same order as the three arms of the reference
      // -zt - delz is the exact negative of zt + delz: a sign flip of the sum; above the mixing height 2h is added to it
      const bool below = delz < -zt, above = !below && delz > (h - zt);
      flip = (below || above) ? 0x80000000u : 0u;
      // (written as `below ? -zt - delz : above ? -zt - delz + 2h : zt + delz` the compiler still made a branch of it: +1.1 %)
      const R base = zt + delz;
      const R lift = above ? K(2.) * h : K(0.);
      zt = m_flip(base, flip) + lift;
    }
    if (i != V.ifine) {
      T.zeta = zt * HI.ih;
#ifdef FPX_LANE_STATS
      if (HI.regime == 0) FPX_LANES(st, 5); else if (HI.regime == 1) FPX_LANES(st, 6); else FPX_LANES(st, 7);
#endif
      hanna_short(T, zt, HI, S);
    }
  }
  icbt = flip ? (short)-1 : (short)1;
  if (!cblflag) nrand = nrand + V.ifine + 1;   // "nrand=nrand+i", i = ifine+1 after the loop (advance.f90:499)
  A.nrand = nrand;

  // next sub-step length, advance.f90:504-510
  if (turbswitch)
    ldt = (int)(m_min(m_min(T.tlw, h * m_rcp(m_max(K(2.) * m_abs(wp * T.sigw), K(1.e-5)))), K(0.5) * m_rcp(m_abs(T.dsigwdz))) * V.ctl);
  else
    ldt = (int)(m_min(T.tlw, h * m_rcp(m_max(K(2.) * m_abs(wp), K(1.e-5)))) * V.ctl);
  ldt = max(ldt, V.mintime);

  R w = S.get(S_W);
  if (SETTLE && V.lsettling) {   // advance.f90:518-531
    // (column and species of get_settling come from the stash: the position itself need not stay in registers for this)
    if (V.mdomainfill == 0 && S.get(S_SET_NUM) > K(0.)) w = w + get_settling<R, true>(V, hgt, (int)S.get(S_SETCELL), zt, 0, indz, &S);
    S.put(S_W, w);
  }

  // advance.f90:543-547 (:539-542 are taken where u, v, up, vp are computed)
  zt = zt + w * dt * (R)V.ldirect;
  if (zt >= hgt[V.nz - 1]) zt = hgt[V.nz - 1] - K(100.) * eps;

  const bool end_of_interval = A.itimec == itime + V.lsynctime;
  if (zt > h) {   // advance.f90:549-552
    if (end_of_interval) {
      // -> 99.  The reference reaches label 99 here with usig/vsig/wsig still holding
      // whatever the previous particle left in interpol_mod; the caller uses this particle's
      // own profile values, as the regular exit :603-606 does.
      return PBL_DONE;
    }
    return PBL_ESCAPED;   // -> 700
  }

  // dry-deposition probability, advance.f90:582-599: prob(ks) = 1 + (prob(ks) - 1)*exp(-vdepo(ks)*|dt|/(2*href)) in every pass
  // that ends below 2*href, from prob = 0.  vdepo is the same in every pass of the step (the cell and the time weights do not
  // change: the reference's depoindicator cache), so 1 - prob is the product of the exponentials = exp(-vdepo*T/(2*href)) with
  // T the sum of |dt| over those passes: the loop keeps T (integer seconds, exact) and k_pbl_finish takes one exponential per
  // species -- the same number up to the rounding of the product.
  if (DRYDEP && zt < K(2.) * href) S.add(S_TDEP, m_abs(dt));

  if (zt < K(0.)) zt = m_min(h - eps2, K(-1.) * zt);   // advance.f90:601

  if (end_of_interval) return PBL_DONE;   // advance.f90:603-608 (sigmas: see level_pair_sigma)
  return PBL_CONTINUE;
}

// the single step above the PBL, advance.f90:629-708 (label 700)
// HAVE_TROPO: adv_begin<.., TROPO = true> has fetched A.tropop already; LATE: a callable that issues the caller's loads
// which are only needed after the gather (they then travel with the gather's last round trip)
template <typename R, typename RNG, bool HAVE_TROPO = false, typename LATE = NoLate>
FPX_DEV void above_step(const View<R> &V, const R *hgt, const RNG &G, const TimeW<R> &W, int itime, double xt, double yt,
                        R &zt, R &wp, int &ldt, AdvCtx<R> &A, R &usig, R &vsig, R &wsig, const LATE &late = LATE()) {
  const R eps2 = K(1.e-9);
  const Fld<R> F = fld_of(V, A.ngrid);
  R tropop;
  if (HAVE_TROPO) tropop = A.tropop;
  else {
    // tropopause(nix,njy,1,1) / tropopausen(nix,njy,1,1,ngrid), advance.f90:253,263 (literal slot 1)
    const int nix = A.ngrid > 0 ? (int)lround((double)A.xr) : (int)lround(xt);
    const int njy = A.ngrid > 0 ? (int)lround((double)A.yr) : (int)lround(yt);
    tropop = F.tropo[(long long)njy * F.nx + nix];
  }
  Cell<R> C;
  cell_setup(C, A.ix, A.jy, A.ixp, A.jyp, A.xr, A.yr);
  if (V.turbmesoscale == K(0.)) {   // the standard deviations only feed the mesoscale term (advance.f90:728-739)
    usig = K(0.); vsig = K(0.); wsig = K(0.);
#ifndef FPX_ABOVE_DEPTH
#define FPX_ABOVE_DEPTH 4
#endif
    interp_wind<R, false, LATE, FPX_ABOVE_DEPTH>(V, hgt, F, C, W, zt, A.u, A.v, A.w, usig, vsig, wsig, late, F.w3t0);
  } else {
    interp_wind<R, true>(V, hgt, F, C, W, zt, A.u, A.v, A.w, usig, vsig, wsig, late);
  }
  ldt = abs(V.lsynctime - A.itimec + itime);
  const R dt = (R)ldt;
  int nrand = A.nrand;
  R ux, vy;
  if ((V.d_trop == K(0.) && V.d_strat == K(0.)) || V.turboff != 0) {
    // diffusivities switched off (wave-uniform): every random term below is multiplied by zero -- or turboff zeroes ux, vy, wp
    // after the draws (advance.f90:675-679) -- so no random numbers are generated; nrand advances as in the three branches
    ux = K(0.); vy = K(0.); wp = K(0.);
    if (zt < tropop) { if (nrand + 1 > V.maxrand) nrand = 1; nrand = nrand + 2; }
    else if (zt < tropop + K(1000.)) { if (nrand + 2 > V.maxrand) nrand = 1; nrand = nrand + 3; }
    else { if (nrand > V.maxrand) nrand = 1; nrand = nrand + 1; }
  } else if (zt < tropop) {
    R uxscale = m_sqrt(K(2.) * V.d_trop / dt);
    if (nrand + 1 > V.maxrand) nrand = 1;
    ux = G.at(nrand) * uxscale;
    vy = G.at(nrand + 1) * uxscale;
    nrand = nrand + 2;
    wp = K(0.);
  } else if (zt < tropop + K(1000.)) {
    R weight = (zt - tropop) / K(1000.);
    R uxscale = m_sqrt(K(2.) * V.d_trop / dt * (K(1.) - weight));
    if (nrand + 2 > V.maxrand) nrand = 1;
    ux = G.at(nrand) * uxscale;
    vy = G.at(nrand + 1) * uxscale;
    R wpscale = m_sqrt(K(2.) * V.d_strat / dt * weight);
    wp = G.at(nrand + 2) * wpscale + V.d_strat / K(1000.);
    nrand = nrand + 3;
  } else {
    if (nrand > V.maxrand) nrand = 1;
    ux = K(0.);
    vy = K(0.);
    R wpscale = m_sqrt(K(2.) * V.d_strat / dt);
    wp = G.at(nrand) * wpscale;
    nrand = nrand + 1;
  }
  A.nrand = nrand;
  if (V.lsettling) A.w = A.w + settling_velocity<R, true>(V, hgt, xt, yt, zt, A.nsp);   // advance.f90:686-699
  A.dxsave = A.dxsave + (A.u + ux) * dt;
  A.dysave = A.dysave + (A.v + vy) * dt;
  zt = zt + (A.w + wp) * dt * (R)V.ldirect;
  if (zt < K(0.)) zt = m_min(A.h - eps2, K(-1.) * zt);
}

// label 99 to the end: advance.f90:728-985.  Returns nstop (0 or 3).
// POLAR = false compiles the stereographic-map branch out (grids without poles)
// LATE: a callable that issues the caller's loads with the last column of the Petterssen gather (see interp_wind)
// CAPCHECK (with MOTHER): the grid has polar caps, this particle started outside them -- the grid test after the move
// still has to see a particle that has entered one
template <typename R, typename RNG, bool POLAR = true, bool MOTHER = false, typename LATE = NoLate, bool CAPCHECK = false>
FPX_DEV int adv_finish(const View<R> &V, const R *hgt, const RNG &G, int itime, PState<R> &P, AdvCtx<R> &A,
                       R usig, R vsig, R wsig, const LATE &late = LATE()) {
  const R eps = V.eps;
  const R eps2 = K(1.e-9);
  int nrand = A.nrand;
  // mesoscale fluctuations, advance.f90:728-739
  {
    const R r = V.meso_r, rs = V.meso_rs;
    if (nrand + 2 > V.maxrand) nrand = 1;
    if (V.turbmesoscale == K(0.)) {   // mesoscale fluctuations switched off (wave-uniform): the random terms vanish
      P.usigold = r * P.usigold; P.vsigold = r * P.vsigold; P.wsigold = r * P.wsigold;
    } else {
      P.usigold = m_fma(r, P.usigold, rs * G.at(nrand) * usig * V.turbmesoscale);
      P.vsigold = m_fma(r, P.vsigold, rs * G.at(nrand + 1) * vsig * V.turbmesoscale);
      P.wsigold = m_fma(r, P.wsigold, rs * G.at(nrand + 2) * wsig * V.turbmesoscale);
    }
    A.dxsave = A.dxsave + P.usigold * (R)V.lsynctime;
    A.dysave = A.dysave + P.vsigold * (R)V.lsynctime;
    P.zt = P.zt + P.wsigold * (R)V.lsynctime;
    if (P.zt < K(0.)) P.zt = K(-1.) * P.zt;
  }
  // wind alignment and position update, advance.f90:747-778
  {
    R ux, vy;
    windalign(A.dxsave, A.dysave, A.dawsave, A.dcwsave, ux, vy);
    A.dxsave = A.dxsave + ux;
    A.dysave = A.dysave + vy;
    move_xy<R, POLAR>(V, A.ngrid, P.xt, P.yt, A.dxsave, A.dysave, (R)V.ldirect);
  }
  // (every way out of this function runs the caller's late loads exactly once)
  if (boundary(V, hgt, P.xt, P.yt, P.zt, eps)) { late(); return 3; }   // advance.f90:784-813

  // Petterssen correction, advance.f90:829-985
  if (P.ldt != abs(V.lsynctime)) { late(); return 0; }
  if (abs(itime + P.ldt * V.ldirect) > abs(V.memtime1)) { late(); return 0; }
  if ((MOTHER && CAPCHECK ? pick_polar(V, P.yt) : pick_grid<R, MOTHER>(V, P.xt, P.yt)) != A.ngrid) { late(); return 0; }
  R xr, yr;
  int nyrows = V.ny, nxcols = V.nx;
  if (A.ngrid > 0) {   // advance.f90:862-866
    const int l = A.ngrid - 1;
    const NestDesc<R> &N = V.nest[l];
    xr = (R)((P.xt - (double)N.xl) * (double)N.xres);
    yr = (R)((P.yt - (double)N.yl) * (double)N.yres);
    nyrows = V.nest[l].ny; nxcols = V.nest[l].nx;
  } else {
    xr = (R)P.xt; yr = (R)P.yt;
  }
  int ix = A.ngrid > 0 ? (int)xr : (int)P.xt, jy = A.ngrid > 0 ? (int)yr : (int)P.yt;
  int ixp = ix + 1, jyp = jy + 1;
  if (jyp >= nyrows) jyp = nyrows - 1;   // guard: the reference would read the padding row nymax here
  if (ixp >= nxcols) ixp = nxcols - 1;
  R u, v, w;
  {
    R d0, d1, d2;
    Cell<R> C;
    cell_setup(C, ix, jy, ixp, jyp, xr, yr);
#ifndef FPX_PETTERSSEN_DEPTH
#define FPX_PETTERSSEN_DEPTH 4
#endif
    // (P.ldt = |lsynctime| here, :829: the blended pack of itime + lsynctime*ldirect is this call's)
    const Fld<R> FP = fld_of(V, A.ngrid);
    interp_wind<R, false, LATE, FPX_PETTERSSEN_DEPTH>(V, hgt, FP, C, time_weights(V, itime + P.ldt * V.ldirect), P.zt, u, v, w, d0, d1, d2, late, FP.w3t1);
  }
  if (V.lsettling) w = w + settling_velocity<R, true>(V, hgt, P.xt, P.yt, P.zt, A.nsp);   // advance.f90:893-906
  u = (u - A.u) / K(2.);
  v = (v - A.v) / K(2.);
  w = (w - A.w) / K(2.);
  P.zt = P.zt + w * (R)(P.ldt * V.ldirect);
  if (P.zt < K(0.)) P.zt = m_min(A.h - eps2, K(-1.) * P.zt);
  move_xy<R, POLAR>(V, A.ngrid, P.xt, P.yt, u, v, (R)(P.ldt * V.ldirect));
  if (boundary(V, hgt, P.xt, P.yt, P.zt, eps)) return 3;   // advance.f90:956-985
  return 0;
}

// ---------------------------------------------------------------------------
// output-grid sampling: conccalc.f90 (mother output grid) and drydepokernel.f90
// ---------------------------------------------------------------------------
constexpr int kMaxAge = 8;

template <typename R>
struct GridP {
  int on;                                  // an output grid is configured
  int numxgrid, numygrid, numzgrid, maxspec, maxpointspec_act, nclassunc, nageclass;
  int lage[kMaxAge];
  R dxout, dyout, xoutshift, youtshift;
  int ind_samp, ioutputforeachrelease, lusekerneloutput;
  int loutnext, loutstep;
  const R *outheight;                      // [numzgrid]
  R *gridunc;                              // (x, y, z, spec, pointspec, classunc, age), x fastest
  float *drygridunc;                       // (x, y, spec, pointspec, classunc, age): real(dep_prec)
  float *wetgridunc;                       // same shape as drygridunc
  // nested output grid (com_mod.f90:585-586, unc_mod.f90:24-28): same levels and trailing dimensions
  int nested;                              // nested_output == 1
  int numxgridn, numygridn;
  R dxoutn, dyoutn, xoutshiftn, youtshiftn;
  R *griduncn;
  float *drygriduncn, *wetgriduncn;
  // receptor points (com_mod.f90:658-663)
  int numreceptor;
  const R *receptor;                       // [3][numreceptor]: xreceptor, yreceptor, receptorarea
  R *creceptor;                            // [nspec][numreceptor]
};

// one horizontal output grid: the mother grid or the nested one
template <typename R>
struct OutGeom {
  int numx, numy;
  R dxout, dyout, xshift, yshift;
  R *grid;
  float *dry, *wet;
};
template <typename R>
FPX_DEV OutGeom<R> out_geom(const GridP<R> &Gp, bool nest) {
  OutGeom<R> g;
  if (nest) {
    g.numx = Gp.numxgridn; g.numy = Gp.numygridn; g.dxout = Gp.dxoutn; g.dyout = Gp.dyoutn;
    g.xshift = Gp.xoutshiftn; g.yshift = Gp.youtshiftn; g.grid = Gp.griduncn; g.dry = Gp.drygriduncn; g.wet = Gp.wetgriduncn;
  } else {
    g.numx = Gp.numxgrid; g.numy = Gp.numygrid; g.dxout = Gp.dxout; g.dyout = Gp.dyout;
    g.xshift = Gp.xoutshift; g.yshift = Gp.youtshift; g.grid = Gp.gridunc; g.dry = Gp.drygridunc; g.wet = Gp.wetgridunc;
  }
  return g;
}

template <typename R>
FPX_DEV int ageclass(const GridP<R> &Gp, int itage) {   // conccalc.f90:54-58, timemanager.f90:545-548
  int nage = 1;
  for (; nage <= Gp.nageclass; nage++)
    if (itage < Gp.lage[nage - 1]) break;
  return nage;
}

FPX_DEV double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
FPX_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Add val to base[idx] for the lanes with valid set -- one atomic per RUN of neighbouring lanes that target the same
// element.  After a locality sort the slots of a wave are ordered by met-grid cell, so lanes that hit the same output cell
// sit next to each other (a 2-D deposition cell collects a whole column of particles, a point release puts everything
// into a handful of cells): their contributions are summed in registers with a segmented reduction over the wave
// (six shuffle steps) and the first lane of every run issues the atomic -- instead of up to 64 atomics colliding on one
// address in L2.  May be called from divergent code: lanes that are not executing split the runs.
template <typename T>
FPX_DEV void wave_run_add(T *base, long long idx, T val, bool valid) {
  const unsigned long long vm = __ballot(valid);
  if (vm == 0ull) return;                                   // nothing to add in this wave
  const int lane = (int)(threadIdx.x & 63);
  if (__popcll(vm) <= 2) {                                  // not worth the shuffles
    if (valid) atomicAdd(base + idx, val);
    return;
  }
  const unsigned long long act = __ballot(1);               // the lanes that execute this call
  const long long key = valid ? idx : (long long)(-1 - lane);   // a lane without a contribution is a run of its own
  const long long pkey = __shfl_up(key, 1);
  const bool head = lane == 0 || !((act >> (lane - 1)) & 1ull) || pkey != key;
  const unsigned long long hm = __ballot(head);
  const int rid = __popcll(hm & (~0ull >> (63 - lane)));    // number of the run this lane belongs to
  T v = valid ? val : (T)0;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const T v2 = __shfl_down(v, o);
    const int r2 = __shfl_down(rid, o);
    if (lane + o < 64 && ((act >> ((lane + o) & 63)) & 1ull) && r2 == rid) v += v2;
  }
  if (head && valid) atomicAdd(base + idx, v);
}

// The uniform kernel of conccalc / wetdepokernel / drydepokernel spreads a particle over its own output cell c0 = (ix, jy)
// and three neighbours (ix +- 1, jy +- 1: which side depends on where in the cell it sits).  Every lane of the wave calls this
// (convergent code): c0 = kNoCell = nothing to add; dix, djy = +-1; w00 -> c0, w10 -> c0 + dix, w01 -> c0 + djy*numx,
// w11 -> c0 + djy*numx + dix (a weight is zero where the target lies outside the grid).
// The lanes that share c0 (same cell, level, species, point, class, age) -- after a locality sort that is most of a wave,
// since the output grids are no finer than the met grid and a deposition grid collects whole columns -- are summed into the
// 3 x 3 neighbourhood of c0 with nine wave reductions, and nine lanes issue ONE atomic wave instruction.
// Why: float atomics run at the memory side (rocprofv3: TCC_EA0_ATOMIC = TCC_ATOMIC for these kernels, 1800 cycles in flight
// each), scattered dwords at about 2e10 per second for the whole chip; the run-merging of wave_run_add only catches
// NEIGHBOURING lanes with the same target, and the three neighbour cells alternate from lane to lane with the particle's
// quadrant: 0.86 atomics per particle in k_wetdepo, 0.43 in k_conccalc before this.
// Sum over the 64 lanes of a wave through DPP (data-parallel primitives: the adds read their second operand from another lane
// of the row / the previous rows directly, no LDS crossbar as with ds_bpermute): two quad permutes and two mirrors leave every
// lane of a 16-lane row with its row's sum, row_bcast15 / row_bcast31 carry the sums on to the rows behind; lane 63 holds the
// total.  Every lane of the wave must be active.  (The neighbourhood sums of wave_kernel_add are nine of these per group of
// lanes: with __shfl_xor they were 54 cross-lane moves through LDS for f32 and 108 for fp64, and k_conccalc<double> took 6.8
// instead of 4.5 ms.)
template <int CTRL, int ROW_MASK = 0xf>
FPX_DEV int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true); }
template <int CTRL, int ROW_MASK = 0xf>
FPX_DEV float dpp_get(float v) { return __int_as_float(dpp_mov<CTRL, ROW_MASK>(__float_as_int(v))); }
template <int CTRL, int ROW_MASK = 0xf>
FPX_DEV double dpp_get(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = dpp_mov<CTRL, ROW_MASK>((int)(b & 0xffffffffll)), hi = dpp_mov<CTRL, ROW_MASK>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}
template <typename T>
FPX_DEV T wave_total(T v) {
  v += dpp_get<0xB1>(v);          // quad_perm [1,0,3,2]
  v += dpp_get<0x4E>(v);          // quad_perm [2,3,0,1]
  v += dpp_get<0x141>(v);         // row_half_mirror
  v += dpp_get<0x140>(v);         // row_mirror: every lane holds the sum of its row of 16
  v += dpp_get<0x142, 0xa>(v);    // row_bcast15 into rows 1 and 3: + the row before
  v += dpp_get<0x143, 0xc>(v);    // row_bcast31 into rows 2 and 3: + rows 0 and 1
  if (sizeof(T) == 8) {
    const long long b = __double_as_longlong((double)v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return (T)__longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
  }
  return (T)__int_as_float(__builtin_amdgcn_readlane(__float_as_int((float)v), 63));
}

constexpr long long kNoCell = (long long)0x8000000000000000ull;   // "this lane adds nothing" (a cell index may be negative: the own cell can lie outside the grid)
template <typename T>
FPX_DEV void wave_kernel_add(T *base, long long c0, int numx, int dix, int djy, T w00, T w10, T w01, T w11) {
  const bool valid = c0 != kNoCell;
  unsigned long long rem = __ballot(valid);
  if (rem == 0ull) return;
  const int lane = (int)(threadIdx.x & 63);
  for (int round = 0; rem != 0ull; round++) {      // wave-uniform
    const int lead = __ffsll((unsigned long long)rem) - 1;
    const long long ref = __shfl(c0, lead);
    // (a cloud that is not cell-sorted has a wave's lanes in as many cells: after a few groups the rest goes one by one)
    const bool in = valid && ((rem >> lane) & 1ull) && (c0 == ref || round >= 6);
    const unsigned long long gm = __ballot(in);
    rem &= ~gm;
    if (__popcll(gm) <= 2 || round >= 6) {   // not worth the reductions
      if (in) {
        if (w00 != (T)0) atomicAdd(base + c0, w00);
        if (w10 != (T)0) atomicAdd(base + c0 + dix, w10);
        if (w01 != (T)0) atomicAdd(base + c0 + (long long)djy * numx, w01);
        if (w11 != (T)0) atomicAdd(base + c0 + (long long)djy * numx + dix, w11);
      }
      continue;
    }
    T mine = (T)0;             // lane b < 9: the sum of bin b = (bx + 1) + 3 (by + 1)
#pragma unroll
    for (int b = 0; b < 9; b++) {
      const int bx = b % 3 - 1, by = b / 3 - 1;
      T v = (T)0;
      if (in) {
        if (bx == 0 && by == 0) v = w00;
        else if (by == 0) v = bx == dix ? w10 : (T)0;
        else if (bx == 0) v = by == djy ? w01 : (T)0;
        else v = (bx == dix && by == djy) ? w11 : (T)0;
      }
      v = wave_total(v);
      if (lane == b) mine = v;
    }
    if (lane < 9 && mine != (T)0) atomicAdd(base + ref + (lane % 3 - 1) + (long long)(lane / 3 - 1) * numx, mine);
  }
}

// Guard without a counterpart in the reference: a particle older than lage(nageclass) (nage = nageclass+1 after the
// loop; possible only before its first epilogue, e.g. after a warm start), an uncertainty class outside 1..nclassunc
// or a release point outside 1..maxpointspec_act would address planes outside the grids; such a particle is not sampled.
template <typename R>
FPX_DEV bool grid_planes_ok(const GridP<R> &Gp, int nage, int nclass, int kp) {
  return nage >= 1 && nage <= Gp.nageclass && nclass >= 1 && nclass <= Gp.nclassunc && kp >= 1 && kp <= Gp.maxpointspec_act;
}

// conccalc.f90:50-295 for one particle (active == this lane holds a particle that is due)
template <typename R>
FPX_DEV void conccalc_particle(const View<R> &V, const GridP<R> &Gp, const R *hgt, bool active, double xt, double yt, R zt,
                               int itage, int npoint, int nclass, const R *xmass, R weight, const R *scav = nullptr /* max(xscav_frac1, 0) per species, or null */,
                               const R *outh = nullptr /* the block's copy of outheight in LDS, or null */) {
  const int nage = ageclass(Gp, itage);
  const bool planes_ok = grid_planes_ok(Gp, nage, nclass, (Gp.ioutputforeachrelease == 0 || V.mdomainfill == 1) ? 1 : npoint);
  R rhoi = K(1.);
  if (active && Gp.ind_samp == -1) {   // conccalc.f90:80-122, density at the particle
    int ix = (int)xt, jy = (int)yt;
    int ixp = min(ix + 1, V.nx - 1), jyp = jy + 1;
    if (jyp >= V.ny) jyp = jyp - 1;
    R ddx = (R)(xt - (double)(R)ix), ddy = (R)(yt - (double)(R)jy);
    R rddx = K(1.) - ddx, rddy = K(1.) - ddy;
    R p1 = rddx * rddy, p2 = ddx * rddy, p3 = rddx * ddy, p4 = ddx * ddy;
    int indz = find_level(hgt, V.nz, zt);
    R dz1 = zt - hgt[indz - 1], dz2 = hgt[indz] - zt;
    R dz = K(1.) / (dz1 + dz2);
    R rhoprof[2];
    const long long c00 = (long long)jy * V.nx + ix, c10 = (long long)jy * V.nx + ixp;
    const long long c01 = (long long)jyp * V.nx + ix, c11 = (long long)jyp * V.nx + ixp;
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const int lev = indz + n;
      // first corner: slot memind(2); the other three: literal slot 2 (conccalc.f90:117-120)
      rhoprof[n] = p1 * V.r2[((c00 * V.nz + (lev - 1)) * 2 + V.m2) * 2] + p2 * V.r2[((c10 * V.nz + (lev - 1)) * 2 + 1) * 2] +
                   p3 * V.r2[((c01 * V.nz + (lev - 1)) * 2 + 1) * 2] + p4 * V.r2[((c11 * V.nz + (lev - 1)) * 2 + 1) * 2];
    }
    rhoi = (dz1 * rhoprof[1] + dz2 * rhoprof[0]) * dz;
  }
  const int nrelpointer = (Gp.ioutputforeachrelease == 0 || V.mdomainfill == 1) ? 1 : npoint;
  int kz = 1;
  if (outh) {   // (from device memory this search is one dependent round trip per output level)
    for (; kz <= Gp.numzgrid; kz++)
      if (outh[kz - 1] > zt) break;
  } else {
    for (; kz <= Gp.numzgrid; kz++)
      if (Gp.outheight[kz - 1] > zt) break;
  }
  const bool inside = active && planes_ok && kz <= Gp.numzgrid;
  for (int ig = 0; ig < (Gp.nested ? 2 : 1); ig++) {   // mother grid :145-295, nested grid :301-441
    const OutGeom<R> G = out_geom(Gp, ig == 1);
    const R xl = (R)((xt * (double)V.dx + (double)G.xshift) / (double)G.dxout);
    const R yl = (R)((yt * (double)V.dy + (double)G.yshift) / (double)G.dyout);
    int ix = (int)xl; if (xl < K(0.)) ix = ix - 1;
    int jy = (int)yl; if (yl < K(0.)) jy = jy - 1;
    const bool direct = !Gp.lusekerneloutput || itage < 10800 || xl < K(0.5) || yl < K(0.5) ||
                        xl > (R)(G.numx - 1) - K(0.5) || yl > (R)(G.numy - 1) - K(0.5);
    const R ddx = xl - (R)ix, ddy = yl - (R)jy;
    int ixp, jyp;
    R wx, wy;
    if (ddx > K(0.5)) { ixp = ix + 1; wx = K(1.5) - ddx; } else { ixp = ix - 1; wx = K(0.5) + ddx; }
    if (ddy > K(0.5)) { jyp = jy + 1; wy = K(1.5) - ddy; } else { jyp = jy - 1; wy = K(0.5) + ddy; }
    const bool okx = ix >= 0 && ix <= G.numx - 1, oky = jy >= 0 && jy <= G.numy - 1;
    const bool okxp = ixp >= 0 && ixp <= G.numx - 1, okyp = jyp >= 0 && jyp <= G.numy - 1;
    const long long plane = (long long)G.numx * G.numy;
    const long long sstride = plane * Gp.numzgrid;
    // offset of (.., kz, ks=1, nrelpointer, nclass, nage)
    const long long off = plane * (kz - 1) + sstride * ((long long)Gp.maxspec * ((nrelpointer - 1) + (long long)Gp.maxpointspec_act * ((nclass - 1) + (long long)Gp.nclassunc * (nage - 1))));
    for (int ks = 0; ks < V.nspec; ks++) {
      R m = active ? xmass[ks] / rhoi * weight : K(0.);
      if (scav) m = m * scav[ks];      // DRYBKDEP / WETBKDEP: conccalc.f90:177-181,226-230, ...
      // the base is wave-uniform, the whole (cell, level, species, point, class, age) offset is the per-lane index:
      // lanes are merged into one atomic only when all of it agrees
      const long long o = off + sstride * ks;
      // (the weights of targets outside the grid are zero; a particle whose own cell lies outside still feeds the
      // neighbours that lie inside, as in the reference: c0 may then address a cell that is never written)
      const bool any = inside && ((okx && oky) || (!direct && (okxp || okyp)));
      wave_kernel_add<R>(G.grid, any ? o + (long long)jy * G.numx + ix : kNoCell, G.numx, ixp - ix, jyp - jy,
                         (okx && oky) ? (direct ? m : m * (wx * wy)) : K(0.),
                         (!direct && okxp && oky) ? m * ((K(1.) - wx) * wy) : K(0.),
                         (!direct && okx && okyp) ? m * (wx * (K(1.) - wy)) : K(0.),
                         (!direct && okxp && okyp) ? m * ((K(1.) - wx) * (K(1.) - wy)) : K(0.));
    }
  }
  // concentrations at receptor points, parabolic kernel: conccalc.f90:451-498.  The reference sums the
  // particles of one receptor serially; here every particle adds its share 2*weight*xmass*xkern/h/area
  // (the sum is linear), pre-reduced over the wave.
  for (int n = 0; n < Gp.numreceptor; n++) {
    const R factor = K(.596831), hxmax = K(6.0), hymax = K(4.0), hzmax = K(150.);
    const R sq = m_sqrt((R)itage);
    const R hz = m_min(K(50.) + K(0.3) * sq, hzmax);
    const R zd = zt / hz;
    const R hx = m_min((K(0.29) + K(2.222e-3) * sq) * V.dx + (R)itage * K(1.2e-5), hxmax);
    const R xd = (R)((xt - (double)Gp.receptor[n]) / (double)hx);
    const R hy = m_min((K(0.18) + K(1.389e-3) * sq) * V.dy + (R)itage * K(7.5e-6), hymax);
    const R yd = (R)((yt - (double)Gp.receptor[Gp.numreceptor + n]) / (double)hy);
    const R h = hx * hy * hz;
    const R r2 = xd * xd + yd * yd + zd * zd;
    const bool hit = active && !(zd > K(1.)) && !(xd * xd > K(1.)) && !(yd * yd > K(1.)) && r2 < K(1.);
    const R xkern = factor * (K(1.) - r2);
    const R area = Gp.receptor[2 * Gp.numreceptor + n];
    for (int ks = 0; ks < V.nspec; ks++)
      wave_run_add<R>(Gp.creceptor, (long long)ks * Gp.numreceptor + n, hit ? K(2.) * weight * (xmass[ks] * xkern / h) / area : K(0.), hit);
  }
}

// drydepokernel.f90:41-116 for one species (deposit already in dep_prec = float)
// CONV: every lane of the wave is here (k_pbl_finish): the neighbourhood sums of wave_kernel_add instead of the run merging
template <typename R, bool CONV = false>
FPX_DEV void drydepo_particle(const View<R> &V, const GridP<R> &Gp0, int nunc, float deposit, int ks, R x, R y, int nage, int kp, bool nest = false) {
  // every lane that reaches the call takes part in the run-merged adds (wave_run_add); `on` says whether it contributes
  const bool on = fabsf(deposit) > 0.f && grid_planes_ok(Gp0, nage, nunc, kp);
  // the nested variant (drydepokernel_nest.f90:38-100) always uses the kernel
  struct { int numxgrid, numygrid, maxspec, maxpointspec_act, nclassunc, lusekerneloutput; R dxout, dyout, xoutshift, youtshift; float *drygridunc; } Gp;
  {
    const OutGeom<R> G = out_geom(Gp0, nest);
    Gp.numxgrid = G.numx; Gp.numygrid = G.numy; Gp.dxout = G.dxout; Gp.dyout = G.dyout; Gp.xoutshift = G.xshift; Gp.youtshift = G.yshift;
    Gp.drygridunc = G.dry; Gp.maxspec = Gp0.maxspec; Gp.maxpointspec_act = Gp0.maxpointspec_act; Gp.nclassunc = Gp0.nclassunc;
    Gp.lusekerneloutput = nest ? 1 : Gp0.lusekerneloutput;
  }
  const R xl = (x * V.dx + Gp.xoutshift) / Gp.dxout;
  const R yl = (y * V.dy + Gp.youtshift) / Gp.dyout;
  const int ix = (int)xl, jy = (int)yl;   // no correction for negative xl here, as in the reference
  const R ddx = xl - (R)ix, ddy = yl - (R)jy;
  int ixp, jyp;
  R wx, wy;
  if (ddx > K(0.5)) { ixp = ix + 1; wx = K(1.5) - ddx; } else { ixp = ix - 1; wx = K(0.5) + ddx; }
  if (ddy > K(0.5)) { jyp = jy + 1; wy = K(1.5) - ddy; } else { jyp = jy - 1; wy = K(0.5) + ddy; }
  const long long plane = (long long)Gp.numxgrid * Gp.numygrid;
  const long long g = on ? plane * (ks + (long long)Gp.maxspec * ((kp - 1) + (long long)Gp.maxpointspec_act * ((nunc - 1) + (long long)Gp.nclassunc * (nage - 1)))) : 0;
  const bool okx = ix >= 0 && ix <= Gp.numxgrid - 1, oky = jy >= 0 && jy <= Gp.numygrid - 1;
  const bool okxp = ixp >= 0 && ixp <= Gp.numxgrid - 1, okyp = jyp >= 0 && jyp <= Gp.numygrid - 1;
  if (CONV) {
    const bool kern = Gp.lusekerneloutput != 0;   // wave-uniform
    const bool any = on && ((okx && oky) || (kern && (okxp || okyp)));
    wave_kernel_add<float>(Gp.drygridunc, any ? g + (long long)jy * Gp.numxgrid + ix : kNoCell, Gp.numxgrid, ixp - ix, jyp - jy,
                           (okx && oky) ? (kern ? (float)((R)deposit * (wx * wy)) : deposit) : 0.f,
                           (kern && okxp && oky) ? (float)((R)deposit * ((K(1.) - wx) * wy)) : 0.f,
                           (kern && okx && okyp) ? (float)((R)deposit * (wx * (K(1.) - wy))) : 0.f,
                           (kern && okxp && okyp) ? (float)((R)deposit * ((K(1.) - wx) * (K(1.) - wy))) : 0.f);
    return;
  }
  if (!Gp.lusekerneloutput) {   // wave-uniform
    wave_run_add<float>(Gp.drygridunc, g + (long long)jy * Gp.numxgrid + ix, deposit, on && okx && oky);
    return;
  }
  wave_run_add<float>(Gp.drygridunc, g + (long long)jy * Gp.numxgrid + ix, (float)((R)deposit * (wx * wy)), on && okx && oky);
  wave_run_add<float>(Gp.drygridunc, g + (long long)jyp * Gp.numxgrid + ixp, (float)((R)deposit * ((K(1.) - wx) * (K(1.) - wy))), on && okxp && okyp);
  wave_run_add<float>(Gp.drygridunc, g + (long long)jy * Gp.numxgrid + ixp, (float)((R)deposit * ((K(1.) - wx) * wy)), on && okxp && oky);
  wave_run_add<float>(Gp.drygridunc, g + (long long)jyp * Gp.numxgrid + ix, (float)((R)deposit * (wx * (K(1.) - wy))), on && okx && okyp);
}

// ---------------------------------------------------------------------------
// wet deposition: wetdepo.f90, get_wetscav.f90, interpol_rain.f90, wetdepokernel.f90
// ---------------------------------------------------------------------------
template <typename R>
struct WetNest {             // the same packs for one nested grid (lsprecn, convprecn, tccn, ctwcn, ttn, cloudsn)
  const R *prec, *ctwc, *ttw;
  const signed char *clouds;
  int readclouds;            // readclouds_nest(ngrid), com_mod.f90:146
};
template <typename R>
struct WetP {
  const WetNest<R> *nest;    // [numbnests] in device memory (subscripted per lane), or null
  int wetdepspec[kMaxSpec], readclouds;
  R weta_gas[kMaxSpec], wetb_gas[kMaxSpec], crain_aero[kMaxSpec], csnow_aero[kMaxSpec];
  R ccn_aero[kMaxSpec], in_aero[kMaxSpec], henry[kMaxSpec];
  const R *prec;             // [ny][nx][2 slots][3]   (lsprec, convprec, tcc)
  const R *ctwc;             // [ny][nx][2 slots]
  const R *ttw;              // [ny][nx][nz][2 slots]
  const signed char *clouds; // [ny][nx][nz][2 slots]
};

FPX_DEV float m_log10(float x) { return log10f(x); }
FPX_DEV double m_log10(double x) { return log10(x); }

// get_wetscav.f90:78-314 (mother grid); returns wetscav, grfr = grfraction(1)
template <typename R>
FPX_DEV R get_wetscav(const View<R> &V, const WetP<R> &Wp, const R *hgt, int itime, int ltsample, double xtra1, double ytra1, R ztra1,
                      int ks, R &grfr) {
  const R lfr[5] = {K(0.5), K(0.65), K(0.8), K(0.9), K(0.95)};
  const R cfr[5] = {K(0.4), K(0.55), K(0.7), K(0.8), K(0.9)};
  const R bclr[6] = {K(274.35758), K(332839.59273), K(226656.57259), K(58005.91340), K(6588.38582), K(0.244984)};
  const R bcls[6] = {K(22.7), K(0.0), K(0.0), K(1321.0), K(381.0), K(0.0)};
  const R incloud_ratio = K(6.2), r_air = K(287.05);   // par_mod.f90:59,82
  R wetscav = K(0.);
  // nesting level, get_wetscav.f90:82-90: the plain nest bounds (no eps margin as in advance.f90:167-173)
  int ngrid = 0;
  for (int j = V.numbnests; j >= 1; j--) {
    const NestDesc<R> &N = V.nest[j - 1];
    if (xtra1 > (double)N.xl && xtra1 < (double)N.xr && ytra1 > (double)N.yl && ytra1 < (double)N.yr) { ngrid = j; break; }
  }
  int ix, jy, gnx = V.nx, gny = V.ny;
  R xtn = (R)xtra1, ytn = (R)ytra1;
  const R *prec = Wp.prec, *ctwc = Wp.ctwc, *ttw = Wp.ttw;
  const signed char *clouds = Wp.clouds;
  int readclouds = Wp.readclouds;
  if (ngrid > 0) {   // :97-101
    const NestDesc<R> &N = V.nest[ngrid - 1];
    const WetNest<R> &W = Wp.nest[ngrid - 1];
    xtn = (R)((xtra1 - (double)N.xl) * (double)N.xres);
    ytn = (R)((ytra1 - (double)N.yl) * (double)N.yres);
    ix = (int)xtn; jy = (int)ytn;
    gnx = N.nx; gny = N.ny;
    prec = W.prec; ctwc = W.ctwc; ttw = W.ttw; clouds = W.clouds; readclouds = W.readclouds;
  } else {
    ix = (int)xtra1; jy = (int)ytra1;
  }
  const int interp_time = (int)lround((double)((R)itime - K(0.5) * (R)ltsample));
  int slot = V.m2;   // n = memind(2) unless memtime(1) is nearer (get_wetscav.f90:114-116)
  if (abs(V.memtime0 - interp_time) < abs(V.memtime1 - interp_time)) slot = V.m1;
  // interpol_rain.f90:68-130 / interpol_rain_nests.f90:68-142 (no time interpolation: the nearer slot only)
  // cloud flag and temperature at the particle's level are requested WITH the precipitation gather (they depend on the
  // position only): one memory round trip instead of three for a particle under a raining cloud
  const int hz = find_level(hgt, V.nz, ztra1);
  const long long cidx = (((long long)jy * gnx + ix) * V.nz + (hz - 1)) * 2 + slot;
  int clouds_v = (int)clouds[cidx];
  R act_temp = ttw[cidx];
  R lsp, convp, cc;
  {
    R xt = xtn, yt = ytn;
    if (xt >= (R)(gnx - 1)) xt = (R)(gnx - 1) - K(0.00001);
    if (yt >= (R)(gny - 1)) yt = (R)(gny - 1) - K(0.00001);
    const int ixr = (int)xt, jyr = (int)yt, ixp = ixr + 1, jyp = jyr + 1;
    const R ddx = xt - (R)ixr, ddy = yt - (R)jyr, rddx = K(1.) - ddx, rddy = K(1.) - ddy;
    const R p1 = rddx * rddy, p2 = ddx * rddy, p3 = rddx * ddy, p4 = ddx * ddy;
    const R *a = prec + (((long long)jyr * gnx + ixr) * 2 + slot) * 3, *b = prec + (((long long)jyr * gnx + ixp) * 2 + slot) * 3;
    const R *c = prec + (((long long)jyp * gnx + ixr) * 2 + slot) * 3, *d = prec + (((long long)jyp * gnx + ixp) * 2 + slot) * 3;
    lsp = p1 * a[0] + p2 * b[0] + p3 * c[0] + p4 * d[0];
    convp = p1 * a[1] + p2 * b[1] + p3 * c[1] + p4 * d[1];
    cc = p1 * a[2] + p2 * b[2] + p3 * c[2] + p4 * d[2];
  }
  asm volatile("" : "+v"(lsp), "+v"(convp), "+v"(cc), "+v"(clouds_v), "+v"(act_temp));
  if (lsp < K(0.01) && convp < K(0.01)) return wetscav;
  if (clouds_v <= 1) return wetscav;
  int i, j;
  if (lsp > K(20.)) i = 4; else if (lsp > K(8.)) i = 3; else if (lsp > K(3.)) i = 2; else if (lsp > K(1.)) i = 1; else i = 0;
  if (convp > K(20.)) j = 4; else if (convp > K(8.)) j = 3; else if (convp > K(3.)) j = 2; else if (convp > K(1.)) j = 1; else j = 0;
  const R lf = i == 4 ? lfr[4] : i == 3 ? lfr[3] : i == 2 ? lfr[2] : i == 1 ? lfr[1] : lfr[0];
  const R cf = j == 4 ? cfr[4] : j == 3 ? cfr[3] : j == 2 ? cfr[2] : j == 1 ? cfr[1] : cfr[0];
  grfr = m_max(K(0.05), cc * (lsp * lf + convp * cf) / (lsp + convp));
  const R prec1 = (lsp + convp) / grfr;
  if (clouds_v >= 4) {   // below cloud, get_wetscav.f90:206-246
    if (V.dquer[ks] <= K(0.) && (Wp.weta_gas[ks] > K(0.) || Wp.wetb_gas[ks] > K(0.))) {
      wetscav = Wp.weta_gas[ks] * m_pow(prec1, Wp.wetb_gas[ks]);
    } else if (V.dquer[ks] > K(0.) && (Wp.crain_aero[ks] > K(0.) || Wp.csnow_aero[ks] > K(0.))) {
      const R dquer_m = m_min(K(10.), V.dquer[ks]) / K(1000000.);
      const R l10 = m_log10(dquer_m);
      const R i4 = K(1.) / ((l10 * l10) * (l10 * l10)), i3 = K(1.) / (l10 * (l10 * l10)), i2 = K(1.) / (l10 * l10), i1 = K(1.) / l10;
      if (act_temp >= K(273.) && Wp.crain_aero[ks] > K(0.))
        wetscav = Wp.crain_aero[ks] * m_pow(K(10.), bclr[0] + (bclr[1] * i4) + (bclr[2] * i3) + (bclr[3] * i2) + (bclr[4] * i1) + bclr[5] * m_pow(prec1, K(0.5)));
      else if (act_temp < K(273.) && Wp.csnow_aero[ks] > K(0.))
        wetscav = Wp.csnow_aero[ks] * m_pow(K(10.), bcls[0] + (bcls[1] * i4) + (bcls[2] * i3) + (bcls[3] * i2) + (bcls[4] * i1) + bcls[5] * m_pow(prec1, K(0.5)));
    }
  }
  if (clouds_v < 4) {   // in cloud, get_wetscav.f90:251-311
    if ((Wp.ccn_aero[ks] > K(0.) || Wp.in_aero[ks] > K(0.)) || (Wp.henry[ks] > K(0.) && V.dquer[ks] <= K(0.))) {
      R cl;
      if (readclouds) cl = ctwc[((long long)jy * gnx + ix) * 2 + slot] * (grfr / cc);   // ctwc / ctwcn, :257-262
      else cl = K(1E6) * K(2E-7) * m_pow(prec1, K(0.36));
      R liq_frac, ice_frac;
      if (act_temp <= K(253.)) { liq_frac = K(0); ice_frac = K(1); }
      else if (act_temp >= K(273.)) { liq_frac = K(1); ice_frac = K(0); }
      else {
        const R t = (act_temp - K(273.)) / (K(273.) - K(253.));
        ice_frac = t * t;
        liq_frac = m_max(K(0.), K(1.) - ice_frac);
      }
      const R frac_act = liq_frac * Wp.ccn_aero[ks] + ice_frac * Wp.in_aero[ks];
      R S_i;
      if (V.dquer[ks] > K(0.)) S_i = frac_act / cl;
      else {
        const R cle = (K(1) - cl) / (Wp.henry[ks] * (r_air / K(3500.)) * act_temp) + cl;
        S_i = K(1) / cle;
      }
      wetscav = incloud_ratio * S_i * (prec1 / K(3.6E6));
    }
  }
  return wetscav;
}

// wetdepokernel.f90:38-108 for one species (deposit is a default real, the grid is dep_prec)
template <typename R>
FPX_DEV void wetdepo_scatter(const View<R> &V, const GridP<R> &Gp0, int nunc, R deposit, int ks, R x, R y, int nage, int kp, bool nest = false) {
  // every lane that reaches the call takes part in the run-merged adds (wave_run_add); adding zero changes nothing
  const bool on = grid_planes_ok(Gp0, nage, nunc, kp) && deposit != K(0.);
  // the nested variant (wetdepokernel_nest.f90:38-107) always uses the kernel and truncates with floor()
  struct { int numxgrid, numygrid, maxspec, maxpointspec_act, nclassunc, lusekerneloutput; R dxout, dyout, xoutshift, youtshift; float *wetgridunc; } Gp;
  {
    const OutGeom<R> G = out_geom(Gp0, nest);
    Gp.numxgrid = G.numx; Gp.numygrid = G.numy; Gp.dxout = G.dxout; Gp.dyout = G.dyout; Gp.xoutshift = G.xshift; Gp.youtshift = G.yshift;
    Gp.wetgridunc = G.wet; Gp.maxspec = Gp0.maxspec; Gp.maxpointspec_act = Gp0.maxpointspec_act; Gp.nclassunc = Gp0.nclassunc;
    Gp.lusekerneloutput = nest ? 1 : Gp0.lusekerneloutput;
  }
  const R xl = (x * V.dx + Gp.xoutshift) / Gp.dxout;
  const R yl = (y * V.dy + Gp.youtshift) / Gp.dyout;
  const int ix = nest ? (int)floor((double)xl) : (int)xl, jy = nest ? (int)floor((double)yl) : (int)yl;
  const R ddx = xl - (R)ix, ddy = yl - (R)jy;
  int ixp, jyp;
  R wx, wy;
  if (ddx > K(0.5)) { ixp = ix + 1; wx = K(1.5) - ddx; } else { ixp = ix - 1; wx = K(0.5) + ddx; }
  if (ddy > K(0.5)) { jyp = jy + 1; wy = K(1.5) - ddy; } else { jyp = jy - 1; wy = K(0.5) + ddy; }
  const long long plane = (long long)Gp.numxgrid * Gp.numygrid;
  const long long g = on ? plane * (ks + (long long)Gp.maxspec * ((kp - 1) + (long long)Gp.maxpointspec_act * ((nunc - 1) + (long long)Gp.nclassunc * (nage - 1)))) : 0;
  const bool okx = ix >= 0 && ix <= Gp.numxgrid - 1, oky = jy >= 0 && jy <= Gp.numygrid - 1;
  const bool okxp = ixp >= 0 && ixp <= Gp.numxgrid - 1, okyp = jyp >= 0 && jyp <= Gp.numygrid - 1;
  // (every lane of the wave is here: k_wetdepo keeps its waves convergent for the neighbourhood sums of wave_kernel_add)
  const bool kern = Gp.lusekerneloutput != 0;   // wave-uniform
  const bool any = on && ((okx && oky) || (kern && (okxp || okyp)));
  wave_kernel_add<float>(Gp.wetgridunc, any ? g + (long long)jy * Gp.numxgrid + ix : kNoCell, Gp.numxgrid, ixp - ix, jyp - jy,
                         (okx && oky) ? (kern ? (float)(deposit * (wx * wy)) : (float)deposit) : 0.f,
                         (kern && okxp && oky) ? (float)(deposit * ((K(1.) - wx) * wy)) : 0.f,
                         (kern && okx && okyp) ? (float)(deposit * (wx * (K(1.) - wy))) : 0.f,
                         (kern && okxp && okyp) ? (float)(deposit * ((K(1.) - wx) * (K(1.) - wy))) : 0.f);
}

#undef K
FPX_TU_CLOSE
}  // namespace fpx
