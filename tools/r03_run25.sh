#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03y; mkdir -p $O; cd $R
URNN_TUNE_FUSED_R=0 python tools/parity_slice.py --t0 0 --n 100 --cache /tmp/s0.npz > $O/slice0_f16.txt 2>&1
URNN_LIB=$R/u-rnn_amd/liburnn_hip_bf6.so URNN_TUNE_F16=0 python tools/parity_slice.py --t0 0 --n 100 --cache /tmp/s0.npz > $O/slice0_bf16x6.txt 2>&1
for f in f16 bf16x6; do echo "== $f"; grep -v amdgpu $O/slice0_$f.txt | grep -E "^ +[0-9]+ \||max over|frames where|^#" | cut -c1-200; done
