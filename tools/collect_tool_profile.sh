#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes (separate runs)
# for one of the tools/bench_*.py programs; results under gpurun_out/prof_<tag>/, summarised by
# tools/summarize_tool_profile.py into profiles/<round>/.
#   usage: tools/collect_tool_profile.sh <tag> <script.py> [args...]
set -uo pipefail
TAG="$1"; shift
SCRIPT="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"/{stats,fetch,write,sq}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$ROOT/$SCRIPT" "$@" > "$OUT/stats/bench.json" 2> "$OUT/stats/err.log" || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python "$ROOT/$SCRIPT" "$@" > "$OUT/fetch/bench.json" 2> "$OUT/fetch/err.log" || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python "$ROOT/$SCRIPT" "$@" > "$OUT/write/bench.json" 2> "$OUT/write/err.log" || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq" -- python "$ROOT/$SCRIPT" "$@" > "$OUT/sq/bench.json" 2> "$OUT/sq/err.log" || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq2" -- python "$ROOT/$SCRIPT" "$@" > "$OUT/sq/bench2.json" 2> "$OUT/sq/err2.log" || true
echo "collected $OUT"
