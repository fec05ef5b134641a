#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03f3; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v amdgpu $O/pytest_gpu.log | tail -2 | cut -c1-300
URNN_LONG_T=360 timeout 3000 python -m pytest tests/test_hip_rollout.py -m gpu -x -q -s -k "whole_event" > $O/parity_T360.log 2>&1; echo "T360 rc $?"
grep -v amdgpu $O/parity_T360.log | tail -5 | cut -c1-300
bash tools/r03_run36.sh
