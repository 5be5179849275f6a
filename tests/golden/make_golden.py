"""Generate tests/golden/*.npz from the UNMODIFIED reference (flang build, oracle/_ref).

Run in the build container only (needs /root/reference + flang; see
oracle/build_ref.sh).  The fixtures are data: per synchronisation step the
particle SoA the reference's initialize()/advance() produced for the seeded
scenario `tests/test_oracle_cpu.py:golden_scenario(name)`.  Inputs are not
stored -- they regenerate bit-identically from flexpart_amd/synthetic.py.
Float64 outputs are kept exactly; nothing of the reference's source is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import scenario_io as sio  # noqa: E402
from test_oracle_cpu import CASES, REF_VARIANT, golden_scenario  # noqa: E402

KEYS = ("xtra1", "ytra1", "ztra1", "uap", "ucp", "uzp", "us", "vs", "ws", "idt", "itra1", "cbt", "xmass1")


def main():
    only = set(sys.argv[1:])
    for name in sorted(CASES):
        if only and name not in only:
            continue
        sc = golden_scenario(name)
        for kind in ("r8", "r4"):
            ref = sio.run_reference(sc, kind + REF_VARIANT.get(name, ""))
            out = {}
            for i, s in enumerate(ref["steps"]):
                for k in KEYS + (("xscav_frac1",) if "xscav_frac1" in s else ()):
                    a = s[k]
                    if kind == "r4" and a.dtype == np.float64 and k not in ("xtra1", "ytra1"):
                        a = a.astype(np.float32)      # exact: the values are f32 in the r4 build
                    out[f"s{i}_{k}"] = a
            for gk in ("gridunc", "drygridunc", "wetgridunc", "griduncn", "drygriduncn", "wetgriduncn", "creceptor"):
                if gk in ref:
                    out[gk] = ref[gk]
            if name == "polar":
                out["northpolemap"] = ref["northpolemap"]
                out["southpolemap"] = ref["southpolemap"]
            if name == "hanna":
                # pin the random table too: first/last entries and a checksum
                t = ref["rannumb"]
                out["rannumb_head"] = t[:64]
                out["rannumb_tail"] = t[-64:]
                out["rannumb_sum"] = np.array([t.sum(), np.abs(t).sum(), (t * np.arange(t.size)).sum()])
            path = os.path.join(HERE, f"{name}_{kind}.npz")
            np.savez_compressed(path, **out)
            print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
