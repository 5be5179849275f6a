#!/bin/bash
# SQ-level diagnostic counter passes for bench.py: where do the waves of a kernel wait?  -> gpurun_out/probe_<tag>/pN/
#   usage: tools/pmc_sq_probe.sh <tag> [bench.py args]      (counters only: no trace domains)
set -uo pipefail
TAG="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; OUT="$ROOT/gpurun_out/probe_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1)); mkdir -p "$OUT/p$i"
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python "$ROOT/bench.py" "$@" --no-cpu-baseline --no-pmc --steps 2 --warmup 1 > "$OUT/p$i/out.json" 2> "$OUT/p$i/err.log" || echo "pass $i failed: $(tail -2 $OUT/p$i/err.log)"
done
echo done
