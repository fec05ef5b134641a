#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n2; mkdir -p $O; cd $R
timeout 1500 python tools/noise_floor.py --n 100 --k 3 > $O/noise_floor.txt 2>&1; grep -v amdgpu $O/noise_floor.txt | tail -14 | cut -c1-250
