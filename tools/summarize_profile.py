#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ -> profiles/<round>/<tag>_{kernel_stats.csv,pmc.json}.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are
in KiB and come from separate passes; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced
reads, so it is doubled (our field gathers are 8-24 B per lane -- an uncalibrated width; the factor
is applied as the guide prescribes and the raw counters are kept next to the corrected figure).
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def per_kernel(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def newest(pattern):
    """gpurun merges every call's files into the same directory: take the latest run's."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(root, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    st = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
    shutil.copy(st, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "stats", "bench.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
    out = {"note": __doc__.strip().splitlines()[2:], "kernels": {}}
    sq = per_kernel(newest(os.path.join(src, "sq", "*", "*counter_collection.csv")))
    fe = per_kernel(newest(os.path.join(src, "fetch", "*", "*counter_collection.csv")))
    wr = per_kernel(newest(os.path.join(src, "write", "*", "*counter_collection.csv")))
    for k in sq:
        if "fpx::" not in k or "::k_" not in k:
            continue
        short = re.sub(r"tu_r\d+_p\w+::", "", k.split("(")[0].replace("void ", ""))   # drop the translation unit's inline namespace
        # a timed step: the launch before the last (the last dispatch of a kernel in a counter pass can report SQ_WAVES
        # of two dispatches -- seen for k_pbl_loop: 6144 instead of 3072 with every other counter in line)
        n_l = len(next(iter(sq[k].values())))
        pick = -2 if n_l >= 2 else -1
        d = {c: v[pick] for c, v in sq[k].items()}
        e = {"launches_profiled": n_l, "sq_last_launch": d}
        if d.get("SQ_ACTIVE_INST_VALU"):
            e["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64)
            e["valu_active_per_wave_cycle"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
        fl, wl = fe.get(k, {}).get("FETCH_SIZE", [0]), wr.get(k, {}).get("WRITE_SIZE", [0])
        f = fl[-2] if len(fl) >= 2 else fl[-1]
        w = wl[-2] if len(wl) >= 2 else wl[-1]
        e["FETCH_SIZE_KiB"] = f
        e["WRITE_SIZE_KiB"] = w
        e["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        out["kernels"][short] = e
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
    print("wrote", dst, list(out["kernels"]))


if __name__ == "__main__":
    main()
