#!/usr/bin/env python3
"""tests/golden/conv_<case>_<kind>.npz: what the flang build of the unmodified CONVECT / TLIFT, redist, sort2, f_qvsat, ew, ran3
(behind oracle/ref_conv_driver.f90, oracle/_ref/convref_rK) leaves after each of three convmix passes over
flexpart_amd.synthetic.convection_case(): particle heights, cloud-base mass flux per column, which columns convect, their
nconvtop, and the redistribution matrices of the first eight convective columns.  Run in the build container."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from flexpart_amd import synthetic as syn          # noqa: E402
from oracle import scenario_io as sio              # noqa: E402

CASES = {"forward": dict(), "backward": dict(ldirect=-1, seed=23), "nested": dict(nest=True, seed=31)}

if __name__ == "__main__":
    for name, kw in CASES.items():
        cs = syn.convection_case(**kw)
        for kind in ("r8", "r4"):
            calls = sio.run_conv_reference(cs, kind)
            out = {"ncalls": len(calls)}
            for i, c in enumerate(calls):
                for k in ("ztra1", "cbaseflux", "lconv", "nconvtop", "fm_col", "fmassfrac") + (("cbasefluxn",) if "cbasefluxn" in c else ()):
                    out[f"c{i}_{k}"] = c[k] if k != "fmassfrac" else c[k].astype(np.float64)
            path = os.path.join(HERE, f"conv_{name}_{kind}.npz")
            np.savez_compressed(path, **out)
            print(path, os.path.getsize(path), "convective columns per call:", [int((c["lconv"] == 1).sum()) for c in calls])
