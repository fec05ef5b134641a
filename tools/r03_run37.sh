#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n9; mkdir -p $O; cd $R
timeout 900 python tools/noise_floor.py --n 100 --k 6 --skip-torch > $O/nf_default.txt 2>&1
URNN_LIB=$R/u-rnn_amd/liburnn_hip_v1.so URNN_TUNE_FUSED_R=1 timeout 900 python tools/noise_floor.py --n 100 --k 6 --skip-torch > $O/nf_v1_hh_first.txt 2>&1
URNN_LIB=$R/u-rnn_amd/liburnn_hip_v2.so timeout 900 python tools/noise_floor.py --n 100 --k 6 --skip-torch > $O/nf_v2_four_mfma.txt 2>&1
for f in nf_default nf_v1_hh_first nf_v2_four_mfma; do echo "== $f"; grep -v amdgpu $O/$f.txt | grep "^hip\|^HIP  " | cut -c1-200; done
for v in 1 2; do URNN_LIB=$R/u-rnn_amd/liburnn_hip_v$v.so python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('v$v default', round(r['value'],1))"; done
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('product default', round(r['value'],1))"
