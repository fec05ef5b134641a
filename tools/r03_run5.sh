#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
timeout 600 python tools/stress_overlap.py > $O/stress_grouped.txt 2>&1; grep -v amdgpu $O/stress_grouped.txt | cut -c1-300 | tail -25
URNN_TUNE_GATE_ALLN=0 timeout 600 python tools/stress_overlap.py > $O/stress_nogroup.txt 2>&1; grep -v amdgpu $O/stress_nogroup.txt | cut -c1-300 | tail -25
DIAG_OVERLAP=0 timeout 600 python tools/stress_overlap.py > $O/stress_seq.txt 2>&1; grep -v amdgpu $O/stress_seq.txt | cut -c1-300 | tail -12
URNN_LIB=$R/u-rnn_amd/liburnn_hip_act0.so timeout 600 python tools/stress_overlap.py > $O/stress_act0.txt 2>&1; grep -v amdgpu $O/stress_act0.txt | cut -c1-300 | tail -12
URNN_TUNE_F16=0 URNN_LIB=$R/u-rnn_amd/liburnn_hip_bf6.so timeout 600 python tools/stress_overlap.py > $O/stress_bf6.txt 2>&1; grep -v amdgpu $O/stress_bf6.txt | cut -c1-300 | tail -12
