#!/bin/bash
# TEST INFRASTRUCTURE ONLY.
# Builds the real reference hot path (unmodified Fortran sources, compiled
# where they lie under /root/reference/src) plus our own driver
# oracle/ref_driver.f90 into oracle/_ref/flexref_r4 (reference precision:
# default real = 4 bytes, as src/makefile compiles it) and
# oracle/_ref/flexref_r8 (-fdefault-real-8: the "fp64" oracle of
# BASELINE.json configs 2-3).  No reference source is copied into the repo;
# only objects/binaries land in oracle/_ref/ (git-ignored).
#
# The reference's own build system (makefile + ecCodes + NetCDF) is NOT used:
# the hot path needs none of those libraries.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${FLEXPART_REFERENCE:-/root/reference}/src"
OUT="$HERE/_ref"
FC="${FC:-/opt/rocm/lib/llvm/bin/flang}"
if [ ! -d "$REF" ]; then echo "build_ref: no reference tree at $REF (skipping)"; exit 0; fi
if [ ! -x "$FC" ]; then echo "build_ref: no flang at $FC (skipping)"; exit 0; fi

MODS="par_mod com_mod interpol_mod hanna_mod cmapf_mod point_mod xmass_mod random_mod unc_mod outg_mod conv_mod"
SUBS="advance initialize interpol_all interpol_wind interpol_wind_short interpol_misslev interpol_vdep \
interpol_all_nests interpol_wind_nests interpol_wind_short_nests interpol_misslev_nests interpol_vdep_nests \
hanna hanna1 hanna_short cbl re_initialize_particle initialize_cbl_vel windalign get_settling dynamic_viscosity \
conccalc drydepokernel drydepokernel_nest wetdepo get_wetscav get_vdep_prob interpol_rain interpol_rain_nests wetdepokernel wetdepokernel_nest"

# The class counts maxageclass and nclassunc are compile-time sizes of par_mod ("maximum number of age classes used
# for output", "number of classes used to calculate the uncertainty", par_mod.f90:187-192), both 1 as shipped -- a
# user who wants age classes edits that line.  The 'c' variants are built with maxageclass=4, nclassunc=3 so that the
# age / class / release-point indexing of conccalc, drydepokernel, wetdepokernel can be pinned: the one assignment is
# rewritten in a pipe into the compiler's stdin (FLEXREF_CLASSES="4 3"); no edited source is ever written anywhere.
build_one() {
  local kind="$1"; shift
  local parmod="$1"; shift      # which of the reference's par_mod files supplies the compile-time sizes
  local flags="$*"
  local obj="$OUT/obj_$kind"
  mkdir -p "$obj"
  ( cd "$obj"
    for m in $MODS; do
      src="$REF/$m.f90"
      [ "$m" = "par_mod" ] && src="$REF/$parmod"
      [ "$obj/$m.o" -nt "$src" ] && continue
      # compile-time parameters of the reference that a user changes by editing one line (class counts: FLEXREF_CLASSES;
      # any other: FLEXREF_EDIT="module|text as shipped|replacement", e.g. "com_mod|turboff=.false.|turboff=.true."): the
      # one assignment is rewritten in a pipe into the compiler's stdin -- no edited source is ever written anywhere
      local edit=""
      if [ "$m" = "par_mod" ] && [ -n "${FLEXREF_CLASSES:-}" ]; then
        set -- $FLEXREF_CLASSES
        grep -q 'maxageclass=1,nclassunc=1$' "$src" || { echo "build_ref: par_mod has no 'maxageclass=1,nclassunc=1' line"; exit 1; }
        edit="s/maxageclass=1,nclassunc=1\$/maxageclass=$1,nclassunc=$2/"
      fi
      if [ -n "${FLEXREF_EDIT:-}" ] && [ "${FLEXREF_EDIT%%|*}" = "$m" ]; then
        local rest="${FLEXREF_EDIT#*|}"
        local from="${rest%%|*}" to="${rest#*|}"
        [ "$(grep -cF "$from" "$src")" = "1" ] || { echo "build_ref: $m has not exactly one '$from'"; exit 1; }
        edit="s/$from/$to/"
      fi
      if [ -n "$edit" ]; then
        sed "$edit" "$src" | "$FC" -c -cpp -O2 -mcmodel=medium $flags -x f95-cpp-input - -o "$m.o"
      else
        "$FC" -c -cpp -O2 -mcmodel=medium $flags "$src" -o "$m.o"
      fi
    done
    for s in $SUBS; do
      [ "$obj/$s.o" -nt "$REF/$s.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/$s.f90" -o "$s.o"
    done
    # the product's ISO_C_BINDING shim (compiled against the reference's modules) + our driver
    "$FC" -c -cpp -O2 -mcmodel=medium $flags "$HERE/../flexpart_amd/fortran/flexgpu_mod.f90" -o flexgpu_mod.o
    "$FC" -c -cpp -O2 -mcmodel=medium $flags "$HERE/ref_driver.f90" -o ref_driver.o
    objs=""
    for m in $MODS $SUBS; do objs="$objs $m.o"; done
    # links libflexpart_amd.so (the HIP engine) for the drop-in mode; found at run time
    # relative to the binary, which also lives inside the repo snapshot on the GPU box
    for s in caldate juldate; do
      [ "$obj/$s.o" -nt "$REF/$s.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/$s.f90" -o "$s.o"
    done
    "$FC" -O2 -mcmodel=medium $flags ref_driver.o flexgpu_mod.o caldate.o juldate.o $objs \
        -L"$HERE/../flexpart_amd/csrc" -lflexpart_amd \
        -Wl,-rpath,'$ORIGIN/../../flexpart_amd/csrc' -Wl,-rpath,/opt/rocm/lib \
        -o "$OUT/flexref_$kind"
  )
  echo "build_ref: built $OUT/flexref_$kind"
}

# verttransform_ecmwf behind its own small driver oracle/ref_vt_driver.f90 -> vtref_rK;
# reuses the module objects build_one left in obj_K
build_vt() {
  local kind="$1"; shift
  local flags="$*"
  local obj="$OUT/obj_$kind"
  ( cd "$obj"
    local extra=""
    case "$flags" in *FLEXREF_NESTS*) extra="verttransform_nests";; esac
    for s in verttransform_ecmwf ew qvsat $extra; do
      [ "$obj/$s.o" -nt "$REF/$s.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/$s.f90" -o "$s.o"
    done
    [ -z "$extra" ] || extra="$extra.o"
    "$FC" -c -cpp -O2 -mcmodel=medium $flags "$HERE/ref_vt_driver.f90" -o ref_vt_driver.o
    "$FC" -O2 -mcmodel=medium $flags ref_vt_driver.o flexgpu_mod.o caldate.o juldate.o verttransform_ecmwf.o $extra ew.o qvsat.o \
        par_mod.o com_mod.o cmapf_mod.o point_mod.o xmass_mod.o unc_mod.o outg_mod.o conv_mod.o \
        -L"$HERE/../flexpart_amd/csrc" -lflexpart_amd \
        -Wl,-rpath,'$ORIGIN/../../flexpart_amd/csrc' -Wl,-rpath,/opt/rocm/lib \
        -o "$OUT/vtref_$kind"
  )
  echo "build_ref: built $OUT/vtref_$kind"
}

# partoutput behind oracle/ref_po_driver.f90 -> poref_rK
build_po() {
  local kind="$1"; shift
  local flags="$*"
  local obj="$OUT/obj_$kind"
  ( cd "$obj"
    for s in partoutput caldate juldate; do
      [ "$obj/$s.o" -nt "$REF/$s.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/$s.f90" -o "$s.o"
    done
    "$FC" -c -cpp -O2 -mcmodel=medium $flags "$HERE/ref_po_driver.f90" -o ref_po_driver.o
    "$FC" -O2 -mcmodel=medium $flags ref_po_driver.o flexgpu_mod.o partoutput.o caldate.o juldate.o \
        par_mod.o com_mod.o point_mod.o xmass_mod.o unc_mod.o outg_mod.o conv_mod.o \
        -L"$HERE/../flexpart_amd/csrc" -lflexpart_amd \
        -Wl,-rpath,'$ORIGIN/../../flexpart_amd/csrc' -Wl,-rpath,/opt/rocm/lib \
        -o "$OUT/poref_$kind"
  )
  echo "build_ref: built $OUT/poref_$kind"
}

# readpartpositions (warm start) behind oracle/ref_rp_driver.f90 -> rpref_rK
build_rp() {
  local kind="$1"; shift
  local flags="$*"
  local obj="$OUT/obj_$kind"
  ( cd "$obj"
    for s in readpartpositions caldate juldate; do
      [ "$obj/$s.o" -nt "$REF/$s.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/$s.f90" -o "$s.o"
    done
    "$FC" -c -cpp -O2 -mcmodel=medium $flags "$HERE/ref_rp_driver.f90" -o ref_rp_driver.o
    "$FC" -O2 -mcmodel=medium $flags ref_rp_driver.o flexgpu_mod.o readpartpositions.o caldate.o juldate.o \
        par_mod.o com_mod.o random_mod.o point_mod.o xmass_mod.o unc_mod.o outg_mod.o conv_mod.o \
        -L"$HERE/../flexpart_amd/csrc" -lflexpart_amd \
        -Wl,-rpath,'$ORIGIN/../../flexpart_amd/csrc' -Wl,-rpath,/opt/rocm/lib \
        -o "$OUT/rpref_$kind"
  )
  echo "build_ref: built $OUT/rpref_$kind"
}

# releaseparticles behind oracle/ref_rel_driver.f90 -> relref_rK
build_rel() {
  local kind="$1"; shift
  local flags="$*"
  local obj="$OUT/obj_$kind"
  ( cd "$obj"
    for s in releaseparticles caldate juldate; do
      [ "$obj/$s.o" -nt "$REF/$s.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/$s.f90" -o "$s.o"
    done
    "$FC" -c -cpp -O2 -mcmodel=medium $flags "$HERE/ref_rel_driver.f90" -o ref_rel_driver.o
    "$FC" -O2 -mcmodel=medium $flags ref_rel_driver.o flexgpu_mod.o releaseparticles.o caldate.o juldate.o \
        par_mod.o com_mod.o random_mod.o point_mod.o xmass_mod.o unc_mod.o outg_mod.o conv_mod.o \
        -L"$HERE/../flexpart_amd/csrc" -lflexpart_amd \
        -Wl,-rpath,'$ORIGIN/../../flexpart_amd/csrc' -Wl,-rpath,/opt/rocm/lib \
        -o "$OUT/relref_$kind"
  )
  echo "build_ref: built $OUT/relref_$kind"
}

# the leaf routines of calcpar that compile here: scalev, ew, f_qvsat behind oracle/ref_cp_driver.f90 -> cpref_rK
build_cp() {
  local kind="$1"; shift
  local flags="$*"
  local obj="$OUT/obj_$kind"
  ( cd "$obj"
    for s in scalev ew qvsat; do
      [ "$obj/$s.o" -nt "$REF/$s.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/$s.f90" -o "$s.o"
    done
    "$FC" -c -cpp -O2 -mcmodel=medium $flags "$HERE/ref_cp_driver.f90" -o ref_cp_driver.o
    "$FC" -O2 -mcmodel=medium $flags ref_cp_driver.o scalev.o ew.o qvsat.o par_mod.o -o "$OUT/cpref_$kind"
  )
  echo "build_ref: built $OUT/cpref_$kind"
}

# concoutput (the sparse grid_conc writer) behind oracle/ref_co_driver.f90 -> coref_rK
build_co() {
  local kind="$1"; shift
  local flags="$*"
  local obj="$OUT/obj_$kind"
  ( cd "$obj"
    [ "$obj/mean_mod.o" -nt "$REF/mean_mod.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/mean_mod.f90" -o mean_mod.o
    for s in concoutput concoutput_nest caldate juldate; do
      [ "$obj/$s.o" -nt "$REF/$s.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/$s.f90" -o "$s.o"
    done
    "$FC" -c -cpp -O2 -mcmodel=medium $flags "$HERE/ref_co_driver.f90" -o ref_co_driver.o
    "$FC" -O2 -mcmodel=medium $flags ref_co_driver.o concoutput.o concoutput_nest.o mean_mod.o caldate.o juldate.o par_mod.o com_mod.o unc_mod.o outg_mod.o point_mod.o -o "$OUT/coref_$kind"
  )
  echo "build_ref: built $OUT/coref_$kind"
}

mkdir -p "$OUT"
build_one r4 par_mod.f90
build_one r8 par_mod.f90 -fdefault-real-8
build_vt r4
build_vt r8 -fdefault-real-8
build_po r4
build_po r8 -fdefault-real-8
build_rp r4
build_rp r8 -fdefault-real-8
build_rel r4
build_rel r8 -fdefault-real-8
# the routines of the convective mixing that compile here: CONVECT/TLIFT, redist, sort2, f_qvsat, ew, ran3
# behind oracle/ref_conv_driver.f90 -> convref_rK (convmix.f90 and calcmatrix.f90 need ecCodes' grib_api: not built)
build_conv() {
  local kind="$1"; shift
  local flags="$*"
  local obj="$OUT/obj_$kind"
  ( cd "$obj"
    for s in conv_mod convect43c redist sort2 qvsat ew caldate juldate; do
      [ "$obj/$s.o" -nt "$REF/$s.f90" ] || "$FC" -c -cpp -O2 -mcmodel=medium $flags "$REF/$s.f90" -o "$s.o"
    done
    "$FC" -c -cpp -O2 -mcmodel=medium $flags "$HERE/ref_conv_driver.f90" -o ref_conv_driver.o
    "$FC" -O2 -mcmodel=medium $flags ref_conv_driver.o flexgpu_mod.o conv_mod.o convect43c.o redist.o sort2.o qvsat.o ew.o caldate.o juldate.o \
        par_mod.o com_mod.o random_mod.o point_mod.o xmass_mod.o unc_mod.o outg_mod.o \
        -L"$HERE/../flexpart_amd/csrc" -lflexpart_amd \
        -Wl,-rpath,'$ORIGIN/../../flexpart_amd/csrc' -Wl,-rpath,/opt/rocm/lib \
        -o "$OUT/convref_$kind"
  )
  echo "build_ref: built $OUT/convref_$kind"
}
build_cp r4
build_cp r8 -fdefault-real-8
build_conv r4
build_conv r8 -fdefault-real-8
build_co r4      # (with -fdefault-real-8 concoutput.f90 itself does not compile: no specific of mean_mod's generic matches)
# nested-grid variant: the stock par_mod.f90 has maxnests=0; the reference's own
# par_mod_meteoswiss.f90 (nxmax=721, maxnests=1, nxmaxn=571, nymaxn=301) enables the *_nests path
build_one r8n par_mod_meteoswiss.f90 -fdefault-real-8 -DFLEXREF_NESTS -DFLEXGPU_NESTS
build_vt r8n -fdefault-real-8 -DFLEXREF_NESTS -DFLEXGPU_NESTS
build_conv r8n -fdefault-real-8 -DFLEXREF_NESTS -DFLEXGPU_NESTS
build_rel r8n -fdefault-real-8 -DFLEXREF_NESTS -DFLEXGPU_NESTS
# ... and in the reference's own precision (BASELINE config 5: nests + deposition in f32)
build_one r4n par_mod_meteoswiss.f90 -DFLEXREF_NESTS -DFLEXGPU_NESTS
build_conv r4n -DFLEXREF_NESTS -DFLEXGPU_NESTS
build_rel r4n -DFLEXREF_NESTS -DFLEXGPU_NESTS
# several age classes / uncertainty classes (see build_one)
FLEXREF_CLASSES="4 3" build_one r4c par_mod.f90
FLEXREF_CLASSES="4 3" build_one r8c par_mod.f90 -fdefault-real-8
# the reference's other compile-time switches of the path, one at a time (see build_one): turbulence off and the
# time-interpolated mixing height (com_mod.f90:777-778), the output grid without the kernel (par_mod.f90:39)
FLEXREF_EDIT="com_mod|turboff=.false.|turboff=.true." build_one r8t par_mod.f90 -fdefault-real-8
FLEXREF_EDIT="com_mod|turboff=.false.|turboff=.true." build_one r4t par_mod.f90
FLEXREF_EDIT="com_mod|interpolhmix=.false.|interpolhmix=.true." build_one r8h par_mod.f90 -fdefault-real-8
FLEXREF_EDIT="com_mod|interpolhmix=.false.|interpolhmix=.true." build_one r4h par_mod.f90
FLEXREF_EDIT="par_mod|lusekerneloutput=.true.|lusekerneloutput=.false." build_one r8k par_mod.f90 -fdefault-real-8
FLEXREF_EDIT="par_mod|lusekerneloutput=.true.|lusekerneloutput=.false." build_one r4k par_mod.f90
# ... the same compile-time sizes for the class mean of concoutput (mean_mod over nclassunc = 3) and for the uncertainty
# class readpartpositions draws for every particle of a warm start (ran1, readpartpositions.f90:142-143)
build_co r4c
build_rp r4c
build_rp r8c -fdefault-real-8
