#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n5; mkdir -p $O; cd $R
timeout 600 python tools/error_field_coherence.py 70 > $O/coherence.txt 2>&1; grep -v amdgpu $O/coherence.txt | tail -8 | cut -c1-300
