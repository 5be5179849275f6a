"""Generates tests/golden/co_<case>_<nnn>.bin: the files grid_conc_* written by the UNMODIFIED reference
routine concoutput (flang build oracle/_ref/coref_r4, driven by oracle/ref_co_driver.f90) on the cases of
tests/test_concoutput.py (inputs regenerate bit-identically from flexpart_amd/synthetic.py).
    python tests/golden/make_golden_co.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from flexpart_amd import synthetic as syn  # noqa: E402
from oracle import scenario_io as sio  # noqa: E402
from test_concoutput import CASES  # noqa: E402

for case, kw in CASES.items():
    for name, b in sio.run_co_reference(syn.concoutput_case(**kw), kind="r4c" if kw.get("classes", 1) > 1 else "r4").items():
        open(os.path.join(HERE, f"co_{case}_{name[-3:]}.bin"), "wb").write(b)
        print(case, name, len(b))

# the nested output grid: the unmodified concoutput_nest
for name, b in sio.run_co_reference(syn.concoutput_case(nxg=30, nyg=20, nzg=3, nspec=2, seed=21), nest=True).items():
    open(os.path.join(HERE, f"co_nest_{name[-3:]}.bin"), "wb").write(b)
    print("nest", name, len(b))

# iout = 3: concentration and mixing-ratio files
for name, b in sio.run_co_reference(syn.add_pptv(syn.concoutput_case(nxg=30, nyg=20, nzg=4, nspec=2, seed=8))).items():
    key = ("pptv_" if "pptv" in name else "") + name[-3:]
    open(os.path.join(HERE, f"co_pptv_{key}.bin"), "wb").write(b)
    print("pptv", name, len(b))
