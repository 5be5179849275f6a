#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/collect_tool_profile.sh) -> profiles/<round>/<tag>_{kernel_stats.csv,pmc.json}.

rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB), median over the profiled launches of a kernel;
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 applies the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md for
wide coalesced reads; hbm_bytes_raw = (FETCH_SIZE + WRITE_SIZE)*1024 is kept beside it.
    usage: summarize_tool_profile.py <tag> <round> <kernel name substring> <calls profiled>
"""
import json
import os
import shutil
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import newest, per_kernel


def main():
    tag, rnd, sub, calls = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(root, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(newest(os.path.join(src, "stats", "*", "*kernel_stats.csv")), os.path.join(dst, f"{tag}_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "stats", "bench.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
    fe = per_kernel(newest(os.path.join(src, "fetch", "*", "*counter_collection.csv")))
    wr = per_kernel(newest(os.path.join(src, "write", "*", "*counter_collection.csv")))
    out = {"note": __doc__.strip().splitlines()[2:5], "kernels": {}}
    total = total_raw = 0.0
    for k in fe:
        if sub not in k:
            continue
        f = fe[k]["FETCH_SIZE"]
        w = wr.get(k, {}).get("WRITE_SIZE", [0.0])
        fm, wm = statistics.median(f), statistics.median(w)
        per_call = len(f) / calls
        e = {"launches_profiled": len(f), "launches_per_call": per_call, "FETCH_SIZE_KiB": fm, "WRITE_SIZE_KiB": wm,
             "hbm_bytes": (2 * fm + wm) * 1024.0, "hbm_bytes_raw": (fm + wm) * 1024.0}
        out["kernels"][k.split("(")[0].replace("void ", "")] = e
        total += e["hbm_bytes"] * per_call
        total_raw += e["hbm_bytes_raw"] * per_call
    out["hbm_bytes_per_call"] = total
    out["hbm_bytes_per_call_raw"] = total_raw
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
    print("wrote", dst, {k: round(v["hbm_bytes"] / 1e6, 1) for k, v in out["kernels"].items()}, "total MB", round(total / 1e6, 1))


if __name__ == "__main__":
    main()
