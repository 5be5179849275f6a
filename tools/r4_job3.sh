#!/bin/bash
# round 4, GPU job 3: do the class streams overlap (kernel trace)?  where do the lanes go (lane-stats build)?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="bench.py --particles 12500000 --global-particles 1e8 --no-cpu-baseline --no-pmc"
rm -rf gpurun_out/r4_j3_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_j3_trace -- python $B --steps 2 --warmup 1 --opt pbl_slices=0 --opt pbl_class_streams=1 > gpurun_out/r4_j3_trace.json 2> gpurun_out/r4_j3_trace.err; echo "trace rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4_j3_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_pbl_loop" in r["Kernel_Name"] or "k_pbl_finish" in r["Kernel_Name"] or "k_prep" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-8:]:
    print(r["Kernel_Name"][:40], r.get("Queue_Id"), "start %.3f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6), "end %.3f ms" % ((int(r["End_Timestamp"]) - t0) / 1e6))
PY
export FPX_LIBRARY=$GRAFT_REPO_ROOT/flexpart_amd/csrc/libflexpart_amd_lanestats.so
i=0
for S in "0:0" "0:1" "64,64,64,0:1" "64,64,64,0:0"; do
  i=$((i+1))
  timeout -k 10 300 python $B --steps 4 --warmup 2 --opt pbl_slices=${S%%:*} --opt pbl_class_streams=${S##*:} > gpurun_out/r4_j3_lanes_$i.json 2> gpurun_out/r4_j3_lanes_$i.err; echo "lanes $S rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j3_lanes_*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f, d["config"]["options"], "%.2f ms" % r["step_kernels_ms"]["k_pbl_loop"])
    for k, v in r.get("lane_stats", {}).items():
        print("   ", k, v)
PY
