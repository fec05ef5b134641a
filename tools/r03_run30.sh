#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n4; mkdir -p $O; cd $R
timeout 600 python tools/check_gn_stats.py 70 > $O/gn_stats.txt 2>&1; grep -v amdgpu $O/gn_stats.txt | tail -12 | cut -c1-300
