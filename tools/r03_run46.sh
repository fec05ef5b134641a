#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n14; mkdir -p $O; cd $R
export URNN_LIB=$R/u-rnn_amd/liburnn_hip_v6.so
timeout 600 python -m pytest tests/test_hip_rollout.py -m gpu -x -q -s -k "fused_reset_gate" 2>&1 | grep -E "fused vs|passed|failed" | cut -c1-200
timeout 900 python tools/noise_floor.py --n 100 --k 5 --skip-torch > $O/nf_phase2_f16_scale10.txt 2>&1
grep -v amdgpu $O/nf_phase2_f16_scale10.txt | grep "^hip\|^HIP  " | cut -c1-200
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('variant default', round(r['value'],1))"
unset URNN_LIB
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('product default', round(r['value'],1))"
