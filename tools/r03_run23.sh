#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03w; mkdir -p $O; cd $R
URNN_LONG_T=360 timeout 3000 python -m pytest tests/test_hip_rollout.py -m gpu -x -q -s -k "whole_event" > $O/parity_T360.log 2>&1; echo "rc $?"
grep -v amdgpu $O/parity_T360.log | tail -30 | cut -c1-300
