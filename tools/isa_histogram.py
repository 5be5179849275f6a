#!/usr/bin/env python3
"""Opcode-class histogram of one kernel of a gfx950 assembly listing (hipcc --save-temps .s).

usage: isa_histogram.py file.s 'kernel-name-regex' [--blocks] [--from LABEL --to LABEL]

Classes follow the measured issue costs of tools/valu_rates.hip (cycles per wave64 instruction per SIMD):
f64 arithmetic 4.4, f64 transcendental seeds (v_rcp/rsq/sqrt_f64) 16.3, 32-bit integer multiplies 4.1,
f32 transcendentals 8.2, every other VALU instruction 2.4.  Static counts: a block's weight in the run is
its trip count, which the listing does not carry -- use --blocks to see the basic blocks of the loop nest.
"""
import re
import sys
import collections

COST = {"f64_fma": 4.4, "f64_mul": 4.4, "f64_add": 4.4, "f64_other": 4.4, "f64_trans": 16.3, "cmp_f64": 4.4, "cvt": 4.4,
        "imul32": 4.1, "f32_trans": 8.2, "cndmask": 2.4, "mov": 2.4, "int32": 2.4, "f32": 2.4, "lane": 2.4, "other_valu": 2.4}


def classify(op):
    if not op.startswith("v_"):
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            return "vmem"
        if op.startswith("s_waitcnt"):
            return "waitcnt"
        if op.startswith("s_"):
            return "salu"
        return "misc"
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return "f64_trans"
    if re.match(r"v_cmp\w*_f64|v_cmpx\w*_f64", op):
        return "cmp_f64"
    if re.match(r"v_cvt_", op):
        return "cvt"
    if re.match(r"v_fma_f64|v_fmac_f64", op):
        return "f64_fma"
    if re.match(r"v_mul_f64", op):
        return "f64_mul"
    if re.match(r"v_add_f64", op):
        return "f64_add"
    if "_f64" in op:
        return "f64_other"
    if re.match(r"v_(mul_lo|mul_hi|mad_u64|mad_i64)", op):
        return "imul32"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", op):
        return "f32_trans"
    if re.match(r"v_cndmask", op):
        return "cndmask"
    if re.match(r"v_mov|v_accvgpr|v_pk_mov", op):
        return "mov"
    if re.match(r"v_(readlane|readfirstlane|writelane|permlane|mbcnt|bpermute)", op):
        return "lane"
    if re.search(r"_f32|_f16", op):
        return "f32"
    if re.search(r"_[ui](32|16|24)|_b32|_b64|v_xor|v_and|v_or|v_not|v_lshl|v_lshr|v_ashr|v_bfe|v_bfi|v_alignbit|v_xad|v_add3|v_lshl_add|v_add_co|v_addc|v_sub|v_cmp|v_perm|v_min|v_max", op):
        return "int32"
    return "other_valu"


def main():
    args = sys.argv[1:]
    if len(args) < 2:
        print(__doc__)
        return 1
    path, pat = args[0], re.compile(args[1])
    blocks = "--blocks" in args
    lab_from = args[args.index("--from") + 1] if "--from" in args else None
    lab_to = args[args.index("--to") + 1] if "--to" in args else None
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and pat.search(m.group(1)):
            start = i
            break
    if start is None:
        print("kernel not found")
        return 1
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    cur, depth = "entry", 0
    per_block = collections.OrderedDict()
    active = lab_from is None
    for l in body:
        m = re.match(r"^(\.LBB\w+):|^; %bb\.(\d+):", l)
        if m:
            cur = m.group(1) or ("bb." + m.group(2))
            if lab_from and cur == lab_from:
                active = True
            if lab_to and cur == lab_to:
                active = False
            d = re.search(r"Depth=(\d+)", l)
            depth = int(d.group(1)) if d else depth
            if "Depth=" not in l and "in Loop" not in l and not l.startswith("; %bb") and "Flow" not in l and "exit" not in l:
                pass
            per_block.setdefault(cur, {"depth": depth, "c": collections.Counter(), "note": l.split(";")[-1].strip()[:70]})
            continue
        d = re.search(r";\s+(?:=>)?\s*(?:This|in|Parent).*Depth=(\d+)", l)
        if d and cur in per_block and not per_block[cur]["c"]:
            per_block[cur]["depth"] = int(d.group(1))
            depth = int(d.group(1))
        mm = re.match(r"\s+([a-z_0-9]+)", l)
        if not mm or not active:
            continue
        op = mm.group(1)
        per_block.setdefault(cur, {"depth": depth, "c": collections.Counter(), "note": ""})
        per_block[cur]["c"][classify(op)] += 1
    tot = collections.Counter()
    for b in per_block.values():
        tot.update(b["c"])
    valu = {k: v for k, v in tot.items() if k in COST}
    nv = sum(valu.values())
    cyc = sum(COST[k] * v for k, v in valu.items())
    print(f"kernel lines {start}-{end}; VALU instructions {nv}, static issue cycles {cyc:.0f}")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        share = f"{100.0 * COST[k] * v / cyc:5.1f} % of VALU cycles" if k in COST else ""
        print(f"  {k:11s} {v:6d}  {share}")
    if blocks:
        print("\nblocks (label, loop depth, VALU count, VALU cycles, note)")
        for name, b in per_block.items():
            v = sum(n for k, n in b["c"].items() if k in COST)
            if v == 0:
                continue
            c = sum(COST[k] * n for k, n in b["c"].items() if k in COST)
            top = ", ".join(f"{k}={n}" for k, n in b["c"].most_common(5))
            print(f"  {name:14s} d{b['depth']} {v:5d} {c:7.0f}  {top}  | {b['note']}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
