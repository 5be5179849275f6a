# A/B of compile-time variants of the fused verttransform kernels: tools/ab_vt.sh <variant>...  (libflexpart_amd_<variant>.so)
for v in "$@"; do
  FPX_LIBRARY=$PWD/flexpart_amd/csrc/libflexpart_amd_$v.so python tools/bench_verttransform.py --no-cpu-baseline --reps 7 > gpurun_out/abvt_$v.json 2>>gpurun_out/err.log
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/abvt_*.json")):
    try:
        d=json.load(open(f)); print(f, "%.4f ms"%d["value"], "frac %.3f"%d["roofline"]["frac"])
    except Exception as e: print(f, "failed", e)
PY
