#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n2; mkdir -p $O; cd $R
timeout 1500 python tools/noise_floor.py --n 100 --k 6 > $O/noise_floor2.txt 2>&1; grep -v amdgpu $O/noise_floor2.txt | tail -18 | cut -c1-250
URNN_TUNE_SPLIT=0 timeout 1500 python tools/noise_floor.py --n 100 --k 3 > $O/noise_floor2_fp32mfma.txt 2>&1; grep -v amdgpu $O/noise_floor2_fp32mfma.txt | grep "^HIP\|^hip" | cut -c1-250
