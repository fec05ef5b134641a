#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n10; mkdir -p $O; cd $R
export URNN_LIB=$R/u-rnn_amd/liburnn_hip_v3.so
URNN_TUNE_FUSED_R=0 timeout 900 python tools/noise_floor.py --n 100 --k 4 --skip-torch > $O/nf_threepass.txt 2>&1
URNN_TUNE_FUSED_R=0 URNN_TUNE_FP32_EPI=1 timeout 900 python tools/noise_floor.py --n 100 --k 4 --skip-torch > $O/nf_cand_fp32.txt 2>&1
URNN_TUNE_FUSED_R=0 URNN_TUNE_FP32_EPI=2 timeout 900 python tools/noise_floor.py --n 100 --k 4 --skip-torch > $O/nf_gates_fp32.txt 2>&1
URNN_TUNE_FUSED_R=0 URNN_TUNE_FP32_EPI=3 timeout 900 python tools/noise_floor.py --n 100 --k 4 --skip-torch > $O/nf_both_fp32.txt 2>&1
for f in nf_threepass nf_cand_fp32 nf_gates_fp32 nf_both_fp32; do echo "== $f"; grep -v amdgpu $O/$f.txt | grep "^hip" | cut -c1-200; done
for e in 0 1 3; do URNN_TUNE_FUSED_R=0 URNN_TUNE_FP32_EPI=$e python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('FP32_EPI=$e default', round(r['value'],1))"; done
