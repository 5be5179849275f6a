"""Generate tests/golden/cp_leaves_*.npz from the UNMODIFIED leaf routines of the reference's calcpar that compile in the
build container -- scalev.f90, ew.f90, qvsat.f90 -- through oracle/ref_cp_driver.f90 (oracle/_ref/cpref_rK).  The fixture is
data: their outputs on the seeded arguments tests/test_calcpar.py:leaf_inputs()."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import scenario_io as sio  # noqa: E402
from test_calcpar import leaf_inputs  # noqa: E402

for kind in ("r8", "r4"):
    out = sio.run_cp_leaf_reference(*leaf_inputs(), kind)
    path = os.path.join(HERE, f"cp_leaves_{kind}.npz")
    np.savez_compressed(path, out=out)
    print("wrote", path, os.path.getsize(path))
