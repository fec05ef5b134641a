#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03f4; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v amdgpu $O/pytest_gpu.log | grep -E "501x499|passed|failed" | cut -c1-300
bash tools/r03_run36.sh
