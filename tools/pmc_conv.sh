#!/bin/bash
# counter passes for the convection kernels: tools/pmc_conv.sh <tag> [bench_convmix args]  -> gpurun_out/probe_<tag>/pN/
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, TCC counter budget)
set -uo pipefail
TAG="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; OUT="$ROOT/gpurun_out/probe_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); mkdir -p "$OUT/p$i"
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/tools/bench_convmix.py" --no-cpu-baseline --reps 1 "$@" > "$OUT/p$i/out.json" 2> "$OUT/p$i/err.log" || echo "pass $i failed"
  echo "pass $i done"
done
echo done
