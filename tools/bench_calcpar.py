#!/usr/bin/env python3
"""Device time of fpx_calcpar (second half) at the BASELINE grid: one JSON line.
    python tools/bench_calcpar.py [--nx 361 --ny 181 --nz 138 --real 8]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=361)
    ap.add_argument("--ny", type=int, default=181)
    ap.add_argument("--nz", type=int, default=138)
    ap.add_argument("--real", type=int, default=8, choices=(4, 8))
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from flexpart_amd import synthetic as syn
    from flexpart_amd.engine import Engine
    m = syn.model_levels(nx=a.nx, ny=a.ny, nz=a.nz, polar=False)
    cin = syn.calcpar_inputs(m)
    sc = dict(grid=m["grid"], geom=m["geom"], globalflags=m["globalflags"], nspec=1, npart=0)
    skip = ("height", "nmixz", "uu", "vv", "ww", "rho", "drhodz", "tt", "hmix", "ustar", "wstar", "oli", "tropopause", "vdep")
    sc.update({k: v for k, v in syn.base_scenario(8, 6, 5).items() if k not in sc and k not in skip})
    eng = Engine(sc, compute_real_bytes=a.real, host_real_bytes=a.real)
    eng.verttransform(1, m, None, init=True, want=())
    ms = [eng.calcpar(1, cin)["device_ms"] for _ in range(a.reps + 1)][1:]
    eng.close()
    print(json.dumps({"metric": "calcpar, one wind field", "value": float(np.median(ms)), "unit": "ms (device)", "higher_is_better": False,
                      "dtype": "f64" if a.real == 8 else "f32", "data": "synthetic",
                      "config": {"workload": f"{a.nx}x{a.ny} columns x {a.nz} levels", "reps": a.reps}}), flush=True)


if __name__ == "__main__":
    main()
