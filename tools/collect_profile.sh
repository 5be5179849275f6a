#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + PMC passes for bench.py,
# results under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into the files
# kept under profiles/.  PMC passes run separately from the trace (gpurun refuses mixing them).
#   usage: tools/collect_profile.sh <tag> [bench.py args...]
set -uo pipefail
TAG="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"/{stats,sq,fetch,write}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$ROOT/bench.py" "$@" --no-cpu-baseline --no-pmc > "$OUT/stats/bench.json" 2> "$OUT/stats/err.log" || exit 1
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq" -- python "$ROOT/bench.py" "$@" --no-cpu-baseline --no-pmc > "$OUT/sq/bench.json" 2> "$OUT/sq/err.log" || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python "$ROOT/bench.py" "$@" --no-cpu-baseline --no-pmc > "$OUT/fetch/bench.json" 2> "$OUT/fetch/err.log" || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python "$ROOT/bench.py" "$@" --no-cpu-baseline --no-pmc > "$OUT/write/bench.json" 2> "$OUT/write/err.log" || exit 1
echo "collected $OUT"
