#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n8; mkdir -p $O; cd $R
python tools/parity_slice.py --t0 0 --n 100 --cache /tmp/s0.npz > $O/a_default.txt 2>&1
URNN_TUNE_SPLIT=0 python tools/parity_slice.py --t0 0 --n 100 --cache /tmp/s0.npz > $O/b_fp32mfma.txt 2>&1
URNN_TUNE_SPLIT=0 python tools/parity_slice.py --t0 0 --n 100 --cache /tmp/s0.npz --overlap 0 > $O/c_fp32mfma_onechain.txt 2>&1
for f in a_default b_fp32mfma c_fp32mfma_onechain; do echo "== $f"; grep -v amdgpu $O/$f.txt | grep -E "^ +[0-9]+ \||max over|frames where|^#" | cut -c1-130; done
