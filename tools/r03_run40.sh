#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n11; mkdir -p $O; cd $R
export URNN_LIB=$R/u-rnn_amd/liburnn_hip_v4.so
URNN_TUNE_FUSED_R=0 URNN_TUNE_FP32_EPI=4 timeout 900 python tools/noise_floor.py --n 100 --k 5 --skip-torch > $O/nf_cand_bf16x6.txt 2>&1
URNN_TUNE_FUSED_R=0 URNN_TUNE_FP32_EPI=1 timeout 900 python tools/noise_floor.py --n 100 --k 5 --skip-torch > $O/nf_cand_fp32.txt 2>&1
for f in nf_cand_bf16x6 nf_cand_fp32; do echo "== $f"; grep -v amdgpu $O/$f.txt | grep "^hip" | cut -c1-200; done
for e in 0 4 1; do URNN_TUNE_FUSED_R=0 URNN_TUNE_FP32_EPI=$e python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('three-pass FP32_EPI=$e default', round(r['value'],1))"; done
URNN_TUNE_FP32_EPI=0 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('fused default', round(r['value'],1))"
