#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03x; mkdir -p $O; cd $R
python tools/parity_slice.py --t0 0 --n 100 --cache /tmp/s0.npz > $O/slice0_fused.txt 2>&1
URNN_TUNE_FUSED_R=0 python tools/parity_slice.py --t0 0 --n 100 --cache /tmp/s0.npz > $O/slice0_threepass.txt 2>&1
URNN_TUNE_HEAD_V=1 python tools/parity_slice.py --t0 0 --n 100 --cache /tmp/s0.npz > $O/slice0_headv1.txt 2>&1
python tools/parity_slice.py --t0 0 --n 100 --event-seed 43 --cache /tmp/s1.npz > $O/slice0_ev43.txt 2>&1
for f in fused threepass headv1 ev43; do echo "== $f"; grep -v amdgpu $O/slice0_$f.txt | grep -E "^ +[0-9]+ \||max over|frames where" | cut -c1-200; done
for v in 2 1; do URNN_TUNE_HEAD_V=$v python bench.py --no-cpu-baseline --overlap 0 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('head V=$v one-chain', round(r['value'],1))"; URNN_TUNE_HEAD_V=$v python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('head V=$v default', round(r['value'],1))"; done
timeout 1200 python -m pytest tests/test_hip_train.py tests/test_hip_train_fullsize.py -m gpu -x -q > $O/pytest_train.log 2>&1; echo "train tests rc $?"; grep -v amdgpu $O/pytest_train.log | tail -3 | cut -c1-300
python bench.py --mode train 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train', round(r['value'],1), r['unit'], r['ms_per_step'])"
python bench.py --mode train --seq-num 12 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train seq12', round(r['value'],1), r['unit'], r['ms_per_step'])"
