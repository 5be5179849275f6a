"""Generates tests/golden/po_s<nspec>_<kind>.bin and rp_s2_<kind>.npz: the files partposit_end written by the UNMODIFIED
reference routine partoutput (flang build oracle/_ref/poref_r4|r8, driven by oracle/ref_po_driver.f90)
on the scenarios of tests/test_partoutput.py (inputs regenerate bit-identically from
flexpart_amd/synthetic.py, so only the output files are stored).  Run in the build container:
    python tests/golden/make_golden_po.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import scenario_io as sio  # noqa: E402
from test_partoutput import scenario, restart_setup  # noqa: E402
import numpy as np  # noqa: E402

for nspec in (1, 2):
    for kind in ("r4", "r8"):
        b = sio.run_po_reference(scenario(nspec), kind)
        open(os.path.join(HERE, f"po_s{nspec}_{kind}.bin"), "wb").write(b)
        print(nspec, kind, len(b))

# readpartpositions: what the unmodified routine makes of the reference's own dump
for kind in ("r4", "r8"):
    sc = scenario(2)
    dump = open(os.path.join(HERE, f"po_s2_{kind}.bin"), "rb").read()
    ref = sio.run_rp_reference(dump, restart_setup(sc, 2), kind)
    np.savez_compressed(os.path.join(HERE, f"rp_s2_{kind}.npz"), **{k: np.asarray(v) for k, v in ref.items()})
    print("rp", kind, ref["numpart"])

# ... and with three uncertainty classes (the reference built with nclassunc = 3: rpref_r4c / rpref_r8c, oracle/build_ref.sh):
# the class every particle draws from ran1 (readpartpositions.f90:142-143)
for kind in ("r4", "r8"):
    sc = scenario(2)
    dump = open(os.path.join(HERE, f"po_s2_{kind}.bin"), "rb").read()
    rs = restart_setup(sc, 2)
    rs["restart"][7] = 3
    ref = sio.run_rp_reference(dump, rs, kind + "c")
    np.savez_compressed(os.path.join(HERE, f"rp_s2_classes_{kind}.npz"), **{k: np.asarray(v) for k, v in ref.items()})
    print("rp classes", kind, ref["numpart"], np.bincount(ref["nclass"]))
