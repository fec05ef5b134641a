#!/bin/bash
# Development aid: per-instantiation register / spill report of urnn_gemm.hip (optionally filtered by a grep pattern).
cd "$(dirname "$0")/../u-rnn_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DURNN_NO_PACKED_F32=1 -c urnn_gemm.hip -o /tmp/urnn_gemm_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|VGPRs Spill|Occupancy|AGPRs" | paste - - - - - \
  | sed -e 's/urnn_gemm.hip:[0-9]*:1: remark: //g' -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | grep -E "error|${1:-.}"
