#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
URNN_REPEAT_LAUNCHES=400 timeout 900 python -m pytest tests/test_hip_rollout.py -m gpu -q -x -k "repeated_launches" > $O/repeat_main.log 2>&1
echo "== repeated launches main: rc $?"; grep -v amdgpu $O/repeat_main.log | grep -E "AssertionError|passed|failed" | cut -c1-900
DIAG_N=30 timeout 600 python tools/stress_overlap.py > $O/stress_main.txt 2>&1; grep -v amdgpu $O/stress_main.txt | cut -c1-200 | tail -4
DIAG_N=30 DIAG_OVERLAP=0 timeout 600 python tools/stress_overlap.py > $O/stress_main_seq.txt 2>&1; grep -v amdgpu $O/stress_main_seq.txt | cut -c1-200 | tail -4
URNN_TUNE_F16=0 URNN_REPEAT_LAUNCHES=400 URNN_LIB=$R/u-rnn_amd/liburnn_hip_bf6.so timeout 900 python -m pytest tests/test_hip_rollout.py -m gpu -q -x -k "repeated_launches" > $O/repeat_bf6.log 2>&1
echo "== repeated launches bf6 (old act1): rc $?"; grep -v amdgpu $O/repeat_bf6.log | grep -E "AssertionError|passed|failed" | cut -c1-900
