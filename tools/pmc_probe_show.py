#!/usr/bin/env python3
"""Print the last-launch counters of gpurun_out/probe_<tag>/p*/ for kernels matching a substring."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import per_kernel
tag, sub = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"probe_{tag}", "p*"))):
    fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not fs:
        print(d, "no data"); continue
    a = per_kernel(max(fs, key=os.path.getmtime))
    for k, v in a.items():
        if sub in k:
            print(os.path.basename(d), k.split("(")[0][-36:], {c: x[-1] for c, x in v.items()})
