#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for V in settle1 settle2; do
  FPX_LIBRARY=$GRAFT_REPO_ROOT/flexpart_amd/csrc/libflexpart_amd_$V.so timeout -k 10 400 python bench.py --config 5 --real 4 --without nest,wet --no-cpu-baseline --no-pmc --steps 6 --warmup 3 > gpurun_out/r4_j13_$V.json 2> gpurun_out/r4_j13_$V.err; echo "$V rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_j13_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in r["step_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
