#!/usr/bin/env python3
"""Measures fpx_convmix on one MI355X at the BASELINE grid (361x181 columns, 138 levels, fp64) and times
the reference's own routines beside it on one host core.

Prints ONE JSON line: device time of one convmix call (HIP events on the engine's stream: column marking, scan, CONVECT per
column in batches, redist per particle), how many columns held particles / convected, the scratch batch, and the CPU baseline:
the unmodified CONVECT / redist / sort2 behind oracle/_ref/convref_r8 on a bounded sample (a sub-grid of the same soundings),
scaled per column.
    python tools/bench_convmix.py [--nx 361 --ny 181 --nuvz 138 --particles 1e7 --reps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=361)
    ap.add_argument("--ny", type=int, default=181)
    ap.add_argument("--nuvz", type=int, default=138)
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--real", type=int, default=8, choices=(4, 8))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backward", action="store_true", help="a backward run (ldirect = -1): redist walks down the columns of fmassfrac")
    ap.add_argument("--unsorted", action="store_true", help="leave the particles in their random seeding order (default: fpx_sort_particles "
                    "first -- the order a run keeps them in)")
    a = ap.parse_args()
    from flexpart_amd import synthetic as syn
    from flexpart_amd.engine import Engine, RNG_PHILOX
    n = int(a.particles)
    cs = syn.convection_case(nx=a.nx, ny=a.ny, nuvz=a.nuvz, n=1000, ncalls=1, ldirect=-1 if a.backward else 1)
    sc = syn.base_scenario(a.nx, a.ny, 30, global_grid=False, nsteps=1, ldirect=-1 if a.backward else 1)
    eng = Engine(sc, compute_real_bytes=a.real, host_real_bytes=a.real, rng_mode=RNG_PHILOX, max_particles=n)
    eng.seed_particles(n, zmax=16000.0, itime0=0)
    eng.set_windtime(cs["memtime"], (1, 2))
    eng.conv_init(cs)
    eng.cbaseflux(cs["cbaseflux"])
    for slot in (1, 2):
        eng.upload_conv_fields(slot, *(np.asarray(cs[k])[slot - 1] for k in ("ps", "tt2", "td2", "tth", "qvh")))
    cb0 = eng.cbaseflux()
    if not a.unsorted:
        eng.sort()
    ms, moved = [], 0
    for r in range(a.reps + 1):
        eng.cbaseflux(cb0)                 # every repetition does the same work
        t0 = time.perf_counter()
        moved = eng.convmix(0)
        wall = time.perf_counter() - t0
        if r:
            ms.append((eng.convmix_device_ms, wall * 1e3))
    cb1 = eng.cbaseflux()
    eng.close()
    dms = float(np.median([m[0] for m in ms]))
    nconv = int((cb1 > 0).sum())
    out = {"metric": "convmix, one call", "value": dms, "unit": "ms (device)", "higher_is_better": False,
           "dtype": "f64" if a.real == 8 else "f32", "data": "synthetic",
           "config": {"workload": f"{a.nx}x{a.ny} columns x {a.nuvz} levels, {n:.0e} particles, every column holds particles, " +
                                  ("random storage order" if a.unsorted else "cell-sorted storage order") + (", backward run" if a.backward else ""), "reps": a.reps},
           "wall_ms_whole_call": float(np.median([m[1] for m in ms])), "particles_moved": int(moved), "columns": a.nx * a.ny,
           "columns_with_mass_flux_after": nconv, "us_per_column": dms * 1e3 / (a.nx * a.ny)}
    # the dominant kernel against the HBM roofline, from the committed counter summary of this workload (tools/pmc_conv.sh)
    import glob
    pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"conv_{a.nx}x{a.ny}x{a.nuvz}_pmc.json")))
    if pm and a.real == 8 and n == 10000000 and not a.unsorted and not a.backward:
        ks = json.load(open(pm[-1]))["kernels"]
        name = max(ks, key=lambda k: ks[k]["avg_us"] or 0.0)
        e = ks[name]
        out["roofline"] = {"bound": "hbm", "kernel": name, "traffic": e["hbm_bytes"], "achieved": e["hbm_GBps"], "peak": 8000.0, "unit": "GB/s",
                           "frac": e["hbm_frac_of_8TBps"], "valu_busy": e["valu_busy"], "avg_launch_us": e["avg_us"],
                           "source": os.path.relpath(pm[-1], ROOT) + " (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command; achieved = measured HBM bytes / "
                                     "launch time: the kernel's algorithmic traffic, 24 B per matrix entry, is 88 % of it)",
                           "all_kernels_hbm_bytes_per_call": sum(v["hbm_bytes"] for v in ks.values())}
    if not a.no_cpu_baseline:
        from oracle import scenario_io as sio
        kind = "r8" if a.real == 8 else "r4"
        if sio.have_conv_ref(kind):
            sub = syn.convection_case(nx=48, ny=32, nuvz=min(a.nuvz, 137), n=60000, ncalls=1)   # par_mod: nuvzmax = 138, maxpart = 1e5
            t0 = time.perf_counter()
            sio.run_conv_reference(sub, kind, workdir=os.environ.get("TMPDIR", "/tmp"))
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": dt * 1e3 / (48 * 32) * (a.nx * a.ny), "unit": "ms per call, scaled per column", "cores": 1, "kind": "reference",
                                   "sample": f"convref_{kind}: the unmodified CONVECT / TLIFT / redist / sort2 on 48x32 columns x {min(a.nuvz, 137)} levels, "
                                             f"60000 particles, {dt:.2f} s including the driver's file I/O"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
