#!/usr/bin/env python3
"""Measures fpx_concoutput on one MI355X on BASELINE config 4's output grid (360x180x10):
particles are sampled into the grid by the device, then the grid_conc file is written.  Prints ONE JSON line with
the wall time of the call (device compression + D2H of the compressed dump + file write into tmpfs), the bytes of
the grid it replaces on the PCIe link, and the CPU restatement of the reference's loop beside it.
    python tools/bench_concoutput.py [--particles 2e6 --reps 5]
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
    ap.add_argument("--particles", type=float, default=2e6)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from flexpart_amd import synthetic as syn
    from flexpart_amd.engine import Engine, RNG_PHILOX
    from oracle import oracle as orc
    n = int(a.particles)
    sc = syn.base_scenario(ctl=5.0, ifine=4, cblflag=1, nsteps=3)
    sc["npart"] = 1; sc["itramem"] = np.zeros(1, np.int32); sc["itime0"] = 0
    syn.add_outgrid(sc, nxg=360, nyg=180, nzg=10, outlon0=-180.0, outlat0=-90.0, dxout=1.0, dyout=1.0, ind_samp=-1, old_fraction=0.0)
    del sc["npart"], sc["itramem"]
    eng = Engine(sc, compute_real_bytes=8, host_real_bytes=4, rng_mode=RNG_PHILOX, max_particles=n, sort_interval=4)
    eng.seed_particles(n, seed=0x5EED, frac_pbl=0.5)
    eng.sort()
    for i in range(3):
        eng.set_windtime((0, 10800), (1, 2))
        eng.step_async(i * 900)
        eng.conccalc((i + 1) * 900, 1.0)
    eng.sync()
    case = syn.concoutput_case(nxg=360, nyg=180, nzg=10, nspec=1, wet=False, dry=False)
    case["outheight"] = np.asarray(sc["outheight"], np.float64)
    prefix = "/dev/shm/fpx_grid_conc_"
    wall = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        eng.concoutput(2700, prefix, case["area"], case["volume"], outnum=3.0)
        wall.append(time.perf_counter() - t0)
    g, d = eng.grids()
    eng.close()
    size = os.path.getsize(prefix + "001")
    got = open(prefix + "001", "rb").read()
    os.remove(prefix + "001")
    co = dict(outgrid=np.array([360, 180, 10, 1, 0, 0, 2700], np.int32), outgeom=np.array([1.0, 1.0, -180.0, -90.0, 3.0]),
              outheight=case["outheight"], area=case["area"], volume=case["volume"], gridunc=g[0, 0, 0])
    t0 = time.perf_counter()
    want = orc.co_oracle(co)["_001"]
    t1 = time.perf_counter() - t0
    out = {"metric": "concoutput: one grid_conc file, 360x180x10 output grid", "value": float(np.median(wall)) * 1e3, "unit": "ms (whole call)",
           "higher_is_better": False, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{n:.0e} particles sampled for 3 steps, 1 species", "cells": 648000, "nonzero_cells": int((g > 0).sum()),
                      "file_bytes": size, "identical_to_oracle": got == want},
           "pcie_bytes_replaced": 648000 * 4, "pcie_bytes_now": size,
           "cpu_baseline": {"value": t1 * 1e3, "unit": "ms", "cores": 1, "kind": "port",
                            "sample": "oracle/concoutput_oracle.c on the same grid (incl. marshalling), without the file write"}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
