#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
for v in noslp; do
  DIAG_N=5 DIAG_M=8000 URNN_LIB=$R/u-rnn_amd/liburnn_hip_$v.so timeout 900 python tools/diag_dec1.py > $O/dec1_$v.txt 2>&1
  echo "== $v"; grep -v amdgpu $O/dec1_$v.txt | cut -c1-330 | tail -5
  DIAG_N=30 URNN_LIB=$R/u-rnn_amd/liburnn_hip_$v.so timeout 600 python tools/stress_overlap.py > $O/stress_$v.txt 2>&1; grep -v amdgpu $O/stress_$v.txt | cut -c1-200 | tail -2
  URNN_LIB=$R/u-rnn_amd/liburnn_hip_$v.so python bench.py --no-cpu-baseline --overlap 0 --no-long-run 2>/dev/null | grep -o '"value": [0-9.]*'
done
