#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v amdgpu $O/pytest_gpu.log | tail -3 | cut -c1-300
bash tools/collect_profiles.sh r03 > /dev/null 2>&1
ls $O | head -40
