#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03w2; mkdir -p $O; cd $R
URNN_LONG_T=360 URNN_LONG_MODE=fp32_cand timeout 3000 python -m pytest tests/test_hip_rollout.py -m gpu -x -q -s -k "whole_event" > $O/parity_T360_fp32_cand.log 2>&1; echo "rc $?"
grep -v amdgpu $O/parity_T360_fp32_cand.log | tail -26 | cut -c1-300
