#!/usr/bin/env python3
"""gpurun_out/probe_<tag>/p1..p5 (tools/pmc_conv.sh) + the kernel trace -> profiles/<round>/conv_361x181x138_pmc.json.

Per kernel of fpx_convmix, last launch of the run: FETCH_SIZE / WRITE_SIZE (KiB, separate passes), hbm_bytes =
(2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md for wide coalesced
reads (hbm_bytes_raw without it: k_conv_redist's reads are 8-byte gathers, for which the correction does not apply),
VALU busy = 4*SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs), L2 hits / misses.
    usage: summarize_conv_pmc.py <tag> <round> <kernel_stats.csv written by conv_kernel_stats.py>"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import per_kernel


def main():
    tag, rnd, stats = sys.argv[1], sys.argv[2], sys.argv[3]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cnt = {}
    for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"probe_{tag}", "p*"))):
        fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
        if not fs:
            continue
        for k, v in per_kernel(max(fs, key=os.path.getmtime)).items():
            if "k_conv_" not in k:
                continue
            name = k.split("<")[0].split("::")[-1]
            cnt.setdefault(name, {}).update({c: x[-1] for c, x in v.items()})
    avg_us = {}
    for r in csv.DictReader(open(stats)):
        for name in cnt:
            if name in r["kernel"] and name not in avg_us:
                avg_us[name] = float(r["avg_us"])
    out = {"note": __doc__.strip().splitlines()[2:7], "kernels": {}}
    for name, c in sorted(cnt.items(), key=lambda kv: -avg_us.get(kv[0], 0)):
        if "FETCH_SIZE" not in c:
            continue
        t = avg_us.get(name)
        hb = (2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0
        raw = (c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0
        e = {"avg_us": t, "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c.get("WRITE_SIZE"), "hbm_bytes": hb, "hbm_bytes_raw": raw,
             "hbm_GBps": hb / (t * 1e-6) / 1e9 if t else None, "hbm_frac_of_8TBps": hb / (t * 1e-6) / 8e12 if t else None,
             "valu_busy": 4 * c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024) if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c else None,
             "waves": c.get("SQ_WAVES"), "l2_hit": c.get("TCC_HIT_sum"), "l2_miss": c.get("TCC_MISS_sum")}
        out["kernels"][name] = e
    dst = os.path.join(root, "profiles", rnd, "conv_361x181x138_pmc.json")
    json.dump(out, open(dst, "w"), indent=1)
    for k, e in out["kernels"].items():
        if e["avg_us"] is None or e["valu_busy"] is None:
            continue
        print(f"{k:18s} {e['avg_us']:8.1f} us  {e['hbm_bytes'] / 1e9:6.2f} GB ({e['hbm_bytes_raw'] / 1e9:5.2f} raw)  {e['hbm_GBps']:7.0f} GB/s  frac {e['hbm_frac_of_8TBps']:.2f}  valu {e['valu_busy']:.2f}")


if __name__ == "__main__":
    main()
