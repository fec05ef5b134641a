#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n12; mkdir -p $O; cd $R
export URNN_LIB=$R/u-rnn_amd/liburnn_hip_v5.so
URNN_TUNE_FUSED_R=0 timeout 900 python tools/noise_floor.py --n 100 --k 5 --skip-torch > $O/nf_cand_k8.txt 2>&1
grep -v amdgpu $O/nf_cand_k8.txt | grep "^hip\|^HIP  " | cut -c1-200
timeout 600 python -m pytest tests/test_hip_rollout.py -m gpu -x -q -k "full_size_cell or full_size_rollout" 2>&1 | tail -2
URNN_TUNE_FUSED_R=0 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('three-pass, cand K=8 MFMAs', round(r['value'],1))"
