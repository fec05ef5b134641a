#!/bin/bash
# Development aid: register / spill report of urnn_train.hip kernels (optionally filtered by a grep pattern).
cd "$(dirname "$0")/../u-rnn_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DURNN_NO_PACKED_F32=1 -c urnn_train.hip -o /tmp/urnn_train_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|VGPRs Spill|Occupancy|AGPRs|ScratchSize" | paste - - - - - - \
  | sed -e 's/urnn_train.hip:[0-9]*:[0-9]*: remark: //g' -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | grep -E "error|${1:-.}"
