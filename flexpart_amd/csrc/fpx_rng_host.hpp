// fpx_rng_host.hpp -- host replica of the reference's shared random streams,
// needed only by the TABLE_SEQ (bit-parity) mode: the 1e6-entry Gaussian table
// and the sequential ran3 draws that pick each particle-step's start index.
// Integer-exact restatement of random_mod.f90:93-139 (ran3), :70-90 (gasdev1),
// :45-67 (gasdev); table fill as FLEXPART.f90:47,56-59.
#pragma once
#include "fpx_tu.hpp"
#include <cmath>
#include <cstdlib>
#include <vector>

namespace fpx {
FPX_TU_OPEN

template <typename R>
struct HostRng {
  // ran3 saved state (random_mod.f90:102-106)
  int iff = 0, inext = 0, inextp = 0, ma[56] = {0};
  // gasdev saved state (random_mod.f90:51-52)
  int iset = 0;
  R gset = 0;
  // the "idummy" seeds local to initialize.f90:64 and advance.f90:120: both start at -7,
  // so the FIRST call from each routine re-seeds the shared generator.
  int idummy_init = -7, idummy_adv = -7;
  int idummy_redist = -88;   // redist.f90:69 (convection): its first call re-seeds the shared generator as well

  R ran3(int &idum) {
    const int mbig = 1000000000, mseed = 161803398, mz = 0;
    const R fac = (R)1. / (R)mbig;
    if (idum < 0 || iff == 0) {
      iff = 1;
      int mj = mseed - std::abs(idum);
      mj = mj % mbig;
      ma[55] = mj;
      int mk = 1;
      for (int i = 1; i <= 54; i++) {
        int ii = (21 * i) % 55;
        ma[ii] = mk;
        mk = mj - mk;
        if (mk < mz) mk += mbig;
        mj = ma[ii];
      }
      for (int k = 1; k <= 4; k++)
        for (int i = 1; i <= 55; i++) {
          ma[i] -= ma[1 + (i + 30) % 55];
          if (ma[i] < mz) ma[i] += mbig;
        }
      inext = 0;
      inextp = 31;
      idum = 1;
    }
    if (++inext == 56) inext = 1;
    if (++inextp == 56) inextp = 1;
    int mj = ma[inext] - ma[inextp];
    if (mj < mz) mj += mbig;
    ma[inext] = mj;
    return (R)mj * fac;
  }

  void gasdev1(int &idum, R &r1, R &r2) {
    R v1, v2, r;
    do {
      v1 = (R)2. * ran3(idum) - (R)1.;
      v2 = (R)2. * ran3(idum) - (R)1.;
      r = v1 * v1 + v2 * v2;
    } while (r >= (R)1.0 || r == (R)0.0);
    R fac = std::sqrt((R)-2. * std::log(r) / r);
    r1 = v1 * fac;
    r2 = v2 * fac;
    if (r1 < (R)-3.) r1 = (R)-3.;
    if (r2 < (R)-3.) r2 = (R)-3.;
    if (r1 > (R)3.) r1 = (R)3.;
    if (r2 > (R)3.) r2 = (R)3.;
  }

  R gasdev(int &idum) {
    if (iset == 0) {
      R v1, v2, r;
      do {
        v1 = (R)2. * ran3(idum) - (R)1.;
        v2 = (R)2. * ran3(idum) - (R)1.;
        r = v1 * v1 + v2 * v2;
      } while (r >= (R)1.0 || r == (R)0.0);
      R fac = std::sqrt((R)-2. * std::log(r) / r);
      gset = v1 * fac;
      iset = 1;
      return v2 * fac;
    }
    iset = 0;
    return gset;
  }

  void fill_table(std::vector<R> &tab, int maxrand) {
    tab.resize(maxrand);
    int idummy = -320;
    iff = 0;
    for (int i = 1; i <= maxrand - 1; i += 2) gasdev1(idummy, tab[i - 1], tab[i]);
    gasdev1(idummy, tab[maxrand - 1], tab[maxrand - 2]);
  }

  // int(ran3(idummy)*real(maxrand-1))+1   (advance.f90:153, initialize.f90:68)
  int start_index(int &idum, int maxrand) { return (int)(ran3(idum) * (R)(maxrand - 1)) + 1; }
};

FPX_TU_CLOSE
}  // namespace fpx
