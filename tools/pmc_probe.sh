#!/bin/bash
# diagnostic counter passes for one tool run: tools/pmc_probe.sh <tag> <script.py> [args]  -> gpurun_out/probe_<tag>/pN/
set -uo pipefail
TAG="$1"; shift; SCRIPT="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; OUT="$ROOT/gpurun_out/probe_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1)); mkdir -p "$OUT/p$i"
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python "$ROOT/$SCRIPT" "$@" > "$OUT/p$i/out.json" 2> "$OUT/p$i/err.log" || echo "pass $i failed"
done
echo done
