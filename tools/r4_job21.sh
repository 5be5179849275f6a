#!/bin/bash
# (needs the round-3 worktree of tools/r4_job20.sh)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
rm -rf gpurun_out/r4_j21_r3 gpurun_out/r4_j21_r4
(cd _r3tmp && FPX_BLEND_MIN=1 timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_j21_r3 -- python bench.py --no-cpu-baseline --no-pmc --steps 3 --warmup 1 > /dev/null 2>&1); echo "r3 rc=$?"
timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_j21_r4 -- python bench.py --no-cpu-baseline --no-pmc --steps 3 --warmup 1 > /dev/null 2>&1; echo "r4 rc=$?"
python - <<'PY'
import csv, glob, collections
for t in ("r3", "r4"):
    f = glob.glob(f"gpurun_out/r4_j21_{t}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_pbl_loop" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            per[int(r["Dispatch_Id"])]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    k = sorted(per)[-1]
    e = per[k]
    print(t, {a: "%.4e" % b for a, b in e.items()}, "util %.4f" % (e["SQ_THREAD_CYCLES_VALU"] / (64 * e["SQ_ACTIVE_INST_VALU"])))
PY
