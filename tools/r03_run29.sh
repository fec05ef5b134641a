#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n3; mkdir -p $O; cd $R
timeout 1700 python tools/hybrid_bisect.py --n 100 --variants x3 > $O/hybrid.txt 2>&1; grep -v amdgpu $O/hybrid.txt | tail -14 | cut -c1-250
