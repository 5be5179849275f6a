for v in "$@"; do for i in 1 2; do FPX_LIBRARY=$PWD/flexpart_amd/csrc/libflexpart_amd_$v.so python bench.py --config 2 --no-cpu-baseline --no-pmc --steps 16 --warmup 4 > gpurun_out/ab_${v}_$i.json 2>>gpurun_out/err.log; done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, "%.3g"%d["value"], {k:round(v,4) for k,v in r["step_kernels_ms"].items()}, "frac %.3f"%r["frac"])
    except Exception as e: print(f, "failed", e)
PY
