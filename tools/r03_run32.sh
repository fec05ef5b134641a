#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n6; mkdir -p $O; cd $R
timeout 1500 python tools/truth64.py --n 100 > $O/truth64.txt 2>&1; grep -v amdgpu $O/truth64.txt | tail -24 | cut -c1-330
