#!/usr/bin/env python3
"""Measures fpx_verttransform_ecmwf on one MI355X at the BASELINE grid
(361x181x138, fp64) and times the reference's own routine beside it on one host core.

Prints ONE JSON line: device time of the transform kernels (HIP events on the engine's stream),
its HBM roofline fraction under the compulsory-traffic model (every input array read once, every
output array written once), the wall time of the whole call (H2D of the model-level arrays +
transform + repack into the gather layout), and the CPU baseline.
    python tools/bench_verttransform.py [--nx 361 --ny 181 --nz 138 --real 8 --reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _traffic():
    """HBM bytes per call from the committed PMC summary (profiles/r*/vt_361x181x138_pmc.json), None if absent."""
    import glob
    f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "vt_361x181x138_pmc.json")))
    try:
        return json.load(open(f[-1]))["hbm_bytes_per_call"] if f else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=361)
    ap.add_argument("--ny", type=int, default=181)
    ap.add_argument("--nz", type=int, default=138)
    ap.add_argument("--real", type=int, default=8, choices=(4, 8))
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    from flexpart_amd import synthetic as syn
    from flexpart_amd.engine import Engine, RNG_PHILOX
    nx, ny, nz = a.nx, a.ny, a.nz
    m = syn.model_levels(nx=nx, ny=ny, nz=nz, global_grid=True, polar=True)
    sc = syn.base_scenario(nx, ny, nz, polar=True, nsteps=1)
    sfc = {k: sc[k][0] for k in ("hmix", "ustar", "wstar", "oli", "tropopause")}
    for k in ("height", "nmixz", "uu", "vv", "ww", "rho", "drhodz", "tt", "uupol", "vvpol", "hmix", "ustar", "wstar", "oli", "tropopause", "vdep"):
        sc.pop(k, None)
    eng = Engine(sc, compute_real_bytes=a.real, host_real_bytes=a.real, rng_mode=RNG_PHILOX, options={"vt_unfused": os.environ.get("FPX_VT_UNFUSED", "0")})   # the tool reads the variable, the library reads none
    eng.verttransform(1, m, sfc, init=True, want=())          # warm-up, allocations, z levels
    dev, wall, wall_pinned = [], [], []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        r = eng.verttransform(2, m, sfc, want=())
        wall.append(r["call_ms"] * 1e-3)
        dev.append(r["device_ms"])
    # the same call with host arrays that stay put (the Fortran host's static com_mod arrays): registered for DMA once.
    # The Python mirror copies the scenario into those arrays first; that host-side copy is timed out of the call.
    held = {}
    eng.verttransform(2, m, sfc, want=(), host_arrays=held)
    for _ in range(a.reps):
        t0 = time.perf_counter()
        wall_pinned.append(eng.verttransform(2, m, sfc, want=(), host_arrays=held)["call_ms"] * 1e-3)
    eng.close()
    rb = a.real
    arr = nx * ny * nz * rb
    rows_pol = (ny - 1 - (int((75.0 + 90.0) / (180.0 / (ny - 1))) - 2) + 1) + (int((-75.0 + 90.0) / (180.0 / (ny - 1))) + 3 + 1)
    alg = (6 + 8 + 2 * rows_pol / ny) * arr          # 6 inputs read, 8 outputs written, uupol/vvpol on the polar rows
    dms = float(np.median(dev))
    out = {
        "metric": "verttransform_ecmwf, one wind field", "value": dms, "unit": "ms (device, transform kernels)",
        "higher_is_better": False, "dtype": "f64" if rb == 8 else "f32", "data": "synthetic",
        "config": {"workload": f"{nx}x{ny}x{nz} hybrid-level input -> z levels, polar caps on", "reps": a.reps},
        "wall_ms_whole_call": float(np.median(wall)) * 1e3,
        "wall_ms_whole_call_pinned_host_arrays": float(np.median(wall_pinned)) * 1e3,
        "roofline": {"bound": "hbm", "achieved": alg / (dms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (dms * 1e-3) / 1e9 / 8000.0, "traffic": _traffic(), "kernel": "k_vt_levels + k_vt_fused + k_vt_polar + k_vt_polerow" if os.environ.get("FPX_VT_UNFUSED", "0") != "1" else "k_vt_inc + k_vt_column + k_vt_search + k_vt_fill + k_vt_post + k_vt_polar + k_vt_polerow",
                     "alg_bytes": alg},
    }
    if not a.no_cpu_baseline:
        from oracle import scenario_io as sio
        kind = "r8" if rb == 8 else "r4"
        if sio.have_vt_ref(kind) and (nx, ny, nz) <= (361, 181, 138):
            ref = sio.run_vt_reference(m, kind, workdir=os.environ.get("TMPDIR", "/tmp"), ncalls=3)
            out["cpu_baseline"] = {"value": float(ref["timing"][0]) * 1e3, "unit": "ms per wind field", "cores": 1, "kind": "reference",
                                   "sample": "3 calls of the unmodified verttransform_ecmwf (flang -O2) on the same input, whole routine incl. its cloud diagnostics"}
        else:
            from oracle import oracle as orc
            t0 = time.perf_counter()
            orc.vt_oracle(m, kind)
            out["cpu_baseline"] = {"value": (time.perf_counter() - t0) * 1e3, "unit": "ms per wind field", "cores": 1, "kind": "port",
                                   "sample": "1 call of oracle/verttransform_oracle.c on the same input (includes f64 <-> real conversion of the arrays)"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
