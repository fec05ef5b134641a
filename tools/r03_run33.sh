#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n7; mkdir -p $O; cd $R
timeout 900 python tools/per_step_cell_error.py 100 > $O/per_step.txt 2>&1; grep -v amdgpu $O/per_step.txt | tail -24 | cut -c1-330
