#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
DIAG_N=20 DIAG_M=6000 timeout 900 python tools/diag_dec1.py > $O/dec1.txt 2>&1; grep -v amdgpu $O/dec1.txt | cut -c1-400 | tail -8
DIAG_N=40 timeout 600 python tools/stress_overlap.py > $O/stress_main.txt 2>&1; grep -v amdgpu $O/stress_main.txt | cut -c1-200 | tail -3
DIAG_N=40 DIAG_OVERLAP=0 timeout 600 python tools/stress_overlap.py > $O/stress_main_seq.txt 2>&1; grep -v amdgpu $O/stress_main_seq.txt | cut -c1-200 | tail -3
URNN_REPEAT_LAUNCHES=400 timeout 900 python -m pytest tests/test_hip_rollout.py -m gpu -q -x -k "repeated_launches or bit_stable" > $O/repeat_main.log 2>&1
echo "== determinism tests: rc $?"; grep -v amdgpu $O/repeat_main.log | grep -E "AssertionError|passed|failed" | cut -c1-900
timeout 600 python tools/parity_slice.py > $O/slice_main.txt 2>&1; grep -v "amdgpu.ids" $O/slice_main.txt | tail -4 | cut -c1-300
python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0.log 2>&1
python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03h/bench_*.log')):
    for line in open(f):
        if line.startswith('{'):
            r=json.loads(line); ro=r['roofline']
            print(f, round(r['value'],1), 'frac',round(ro.get('frac',0),3),{k:round(v,1) for k,v in ro.get('launch_us',{}).items()})
PY
