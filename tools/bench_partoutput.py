#!/usr/bin/env python3
"""Measures fpx_partoutput on one MI355X: the dump of N particles resident on the
device on the BASELINE grid (361x181x138), fp64, and the CPU restatement of the reference's routine
beside it on one host core (bounded sample).  Prints ONE JSON line.
    python tools/bench_partoutput.py [--particles 1e7 --real 8 --reps 3 --out /dev/shm/partposit_end]
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
    """HBM bytes per call from the committed PMC summary (profiles/r*/po_1e+07_pmc.json), None if absent."""
    import glob
    f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "po_1e+07_pmc.json")))
    try:
        return json.load(open(f[-1]))["hbm_bytes_per_call_raw"] if f else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--real", type=int, default=8, choices=(4, 8))
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="/dev/shm/fpx_partposit_end")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    a = ap.parse_args()
    from flexpart_amd import synthetic as syn
    from flexpart_amd.engine import Engine, RNG_PHILOX
    n = int(a.particles)
    sc = syn.base_scenario(ctl=5.0, ifine=4, cblflag=1, nsteps=1)
    sc["npart"] = 1
    syn.add_partoutput_fields(sc, itime=0, dead_every=0)
    del sc["npart"], sc["itra1"], sc["npoint"]
    eng = Engine(sc, compute_real_bytes=a.real, host_real_bytes=a.real, rng_mode=RNG_PHILOX, max_particles=n, sort_interval=4)
    eng.upload_diag_fields_from_scenario(sc)
    eng.seed_particles(n, seed=0x5EED, frac_pbl=0.5)
    eng.sort()
    dev, wall = [], []
    nrec = 0
    for _ in range(a.reps):
        t0 = time.perf_counter()
        nrec = eng.partoutput(0, a.out)
        wall.append(time.perf_counter() - t0)
        dev.append(eng.partoutput_device_ms)
    eng.close()
    size = os.path.getsize(a.out)
    os.remove(a.out)
    rb = a.real
    nx, ny, nz = (int(v) for v in sc["grid"])
    reclen = 8 + 11 * rb + 8
    fields = (3 * 2 * nz + 2 * 2 * nz + 2 + 2 + 1) * nx * ny * rb        # pv,qv,tt (2 slots) + rho,drhodz pack + hmix, tropopause, oro
    b_alg = (16 + rb) + 16 + rb + 12 + reclen + fields / n                 # state read, selection, record written, fields once
    dms = float(np.median(dev))
    out = {"metric": "partoutput: particles dumped per second (device, record building)", "value": nrec / (dms * 1e-3), "unit": "particles/s",
           "higher_is_better": True, "dtype": "f64" if rb == 8 else "f32", "data": "synthetic",
           "config": {"workload": f"{n:.0e} particles on the {nx}x{ny}x{nz} grid, all due, after a locality sort", "records": nrec, "file_bytes": size},
           "device_ms": dms, "wall_ms_whole_call": float(np.median(wall)) * 1e3,
           "roofline": {"bound": "hbm", "achieved": b_alg * nrec / (dms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": b_alg * nrec / (dms * 1e-3) / 1e9 / 8000.0, "traffic": _traffic(), "kernel": "k_po_flags + rocprim scan + k_partoutput",
                        "alg_bytes_per_particle": b_alg}}
    # CPU: the C restatement of the reference's routine (file image in memory, no disk), one core
    from oracle import oracle as orc
    m = a.cpu_sample
    s2 = dict(sc)
    s2.update(syn.make_particles(m, nx, ny, sc["height"], sc["hmix"], seed=0x5EED, frac_pbl=0.5))
    syn.add_partoutput_fields(s2, itime=0, dead_every=0)
    orc.build()
    t0 = time.perf_counter()
    img = orc.po_oracle(s2, "r8" if rb == 8 else "r4", nymax=ny)
    t1 = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": m / t1, "unit": "particles/s", "cores": 1, "kind": "port",
                           "sample": f"{m} particles, oracle/partoutput_oracle.c building the file image in memory ({t1:.1f} s incl. marshalling)"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
