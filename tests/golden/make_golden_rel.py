"""Generate tests/golden/rel_*.npz from the UNMODIFIED reference routine releaseparticles.f90 (flang build,
oracle/_ref/relref_rK through our driver oracle/ref_rel_driver.f90, which also runs the splitting block of
timemanager.f90:473-504 after every call).  Run in the build container only.  The fixtures are data: the particle
arrays 1..numpart, numpart / numparticlecount, xmasssave and rho_rel after each call for the seeded scenarios
tests/test_release.py:case(name); inputs regenerate bit-identically from flexpart_amd/synthetic.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import scenario_io as sio  # noqa: E402
from test_release import CASES, KEYS, case  # noqa: E402


def main():
    for name in sorted(CASES):
        rs = case(name)
        for kind in ("r8", "r4"):
            calls = sio.run_rel_reference(rs, kind)
            out = {"ncalls": np.array(len(calls))}
            for i, c in enumerate(calls):
                for k in KEYS + ("xmasssave", "rho_rel"):
                    a = np.asarray(c[k])
                    if kind == "r4" and a.dtype == np.float64 and k not in ("xtra1", "ytra1"):
                        a = a.astype(np.float32)          # exact: the values are f32 in the r4 build
                    out[f"c{i}_{k}"] = a
            path = os.path.join(HERE, f"rel_{name}_{kind}.npz")
            np.savez_compressed(path, **out)
            print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
