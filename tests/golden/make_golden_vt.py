"""Generates tests/golden/vt_<case>_<kind>.npz: outputs of the UNMODIFIED reference routine
verttransform_ecmwf (flang build oracle/_ref/vtref_r4|r8, driven by oracle/ref_vt_driver.f90) on the
synthetic hybrid-level input of flexpart_amd.synthetic.model_levels (regenerated bit-identically
from integer hashes, so only the outputs are stored).  Run in the build container:
    python tests/golden/make_golden_vt.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from flexpart_amd import synthetic as syn  # noqa: E402
from oracle import scenario_io as sio  # noqa: E402
from test_verttransform import CASES, FIELDS  # noqa: E402

for case, kw in CASES.items():
    m = syn.model_levels(**kw)
    for kind in ("r8", "r4"):
        ref = sio.run_vt_reference(m, kind)
        out = {k: ref[k].astype(np.float64 if kind == "r8" else np.float32) for k in FIELDS if k in ref}
        if "uupol" not in out:
            z = np.zeros_like(out["uu"])
            out["uupol"] = z; out["vvpol"] = z
        out["height"] = ref["height"]
        out["nmixz"] = np.int32(ref["nmixz"])
        np.savez_compressed(os.path.join(HERE, f"vt_{case}_{kind}.npz"), **out)
        print(case, kind, {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim == 3}.popitem())

# nested grid: the unmodified verttransform_nests (reference built with par_mod_meteoswiss.f90, maxnests = 1)
from test_verttransform import nest_case  # noqa: E402
m, n = nest_case()
ref = sio.run_vt_reference(m, "r8n", nest=n)
np.savez_compressed(os.path.join(HERE, "vt_nest_r8.npz"), **{k: ref[k] for k in ("uun", "vvn", "wwn", "ttn", "qvn", "pvn", "rhon", "drhodzn")})
print("nest", ref["uun"].shape)
