#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
DIAG_N=50 timeout 900 python tools/diag_dec1.py > $O/dec1.txt 2>&1; grep -v amdgpu $O/dec1.txt | cut -c1-900
DIAG_N=50 URNN_TUNE_GATE_ALLN=0 timeout 900 python tools/diag_dec1.py > $O/dec1_nogroup.txt 2>&1; grep -v amdgpu $O/dec1_nogroup.txt | cut -c1-900
DIAG_N=50 URNN_LIB=$R/u-rnn_amd/liburnn_hip_act0.so timeout 900 python tools/diag_dec1.py > $O/dec1_act0.txt 2>&1; grep -v amdgpu $O/dec1_act0.txt | cut -c1-900
