#!/bin/bash
# Register / scratch / LDS use of every kernel of the HIP library (device-only compile, no GPU needed).
#   usage: tools/kernel_resources.sh [-DFPX_PREP_WAVES=3 ...] | grep -A12 k_prep
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm "$@" \
  --cuda-device-only -c "$ROOT/flexpart_amd/csrc/fpx_engine.hip" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed -e "s/.*remark: [^ ]* //" -e "s/ \[-Rpass.*//" | paste - - - - - - - | c++filt | sed -e "s/(fpx::View.*)/()/"
