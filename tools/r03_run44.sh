#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n13; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_rollout.py -m gpu -x -q -s -k "fused_reset_gate or full_size_rollout" > $O/pytest_fused.log 2>&1; echo "rc $?"; grep -v amdgpu $O/pytest_fused.log | grep -E "fused vs|passed|failed|Error" | cut -c1-200
timeout 900 python tools/noise_floor.py --n 100 --k 5 --skip-torch > $O/nf_fused_fp32_phase2.txt 2>&1
grep -v amdgpu $O/nf_fused_fp32_phase2.txt | grep "^hip\|^HIP  " | cut -c1-200
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('default', round(r['value'],1))"
python bench.py --no-cpu-baseline --overlap 0 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('one chain', round(r['value'],1))"
URNN_TUNE_FUSED_R=0 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('three-pass default', round(r['value'],1))"
