#!/bin/bash
# counter passes for the scatter kernels (k_wetdepo, k_conccalc) of config 5: which unit limits them?
#   tools/pmc_scatter.sh <tag> [bench.py args]  -> gpurun_out/probe_<tag>/pN/
set -uo pipefail
TAG="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; OUT="$ROOT/gpurun_out/probe_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || rocprofv3 -L > "$OUT/avail.txt" 2>&1
grep -o "TCC_[A-Z0-9_]*ATOMIC[A-Z0-9_]*\|TCP_TCC_ATOMIC[A-Z0-9_]*\|TCC_EA0_WRREQ[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*\|TCC_BUSY[A-Z0-9_]*\|TCC_TAG_STALL[A-Z0-9_]*\|TCC_WRITE[A-Z0-9_]*\|TCC_CYCLE[A-Z0-9_]*" "$OUT/avail.txt" | sort -u > "$OUT/names.txt"
i=0
for set in "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1)); mkdir -p "$OUT/p$i"
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python "$ROOT/bench.py" "$@" --no-cpu-baseline --no-pmc > "$OUT/p$i/out.json" 2> "$OUT/p$i/err.log" || echo "pass $i failed: $(tail -2 $OUT/p$i/err.log)"
done
echo done
