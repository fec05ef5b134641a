#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03l; mkdir -p $O
cd $R
DIAG_N=5 DIAG_M=8000 timeout 900 python tools/diag_dec1.py > $O/dec1.txt 2>&1; grep -v amdgpu $O/dec1.txt | cut -c1-300 | tail -2
DIAG_N=5 DIAG_M=4000 DIAG_CELL=enc1 timeout 900 python tools/diag_dec1.py > $O/enc1.txt 2>&1; grep -v amdgpu $O/enc1.txt | cut -c1-300 | tail -1
DIAG_N=5 DIAG_M=4000 DIAG_CELL=dec2 timeout 900 python tools/diag_dec1.py > $O/dec2.txt 2>&1; grep -v amdgpu $O/dec2.txt | cut -c1-300 | tail -1
DIAG_N=40 timeout 600 python tools/stress_overlap.py > $O/stress.txt 2>&1; grep -v amdgpu $O/stress.txt | cut -c1-200 | tail -1
DIAG_N=40 DIAG_OVERLAP=0 timeout 600 python tools/stress_overlap.py > $O/stress_seq.txt 2>&1; grep -v amdgpu $O/stress_seq.txt | cut -c1-200 | tail -1
URNN_REPEAT_LAUNCHES=300 timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v amdgpu $O/pytest_gpu.log | tail -4
python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0.log 2>&1
python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03l/bench_*.log')):
    for line in open(f):
        if line.startswith('{'):
            r=json.loads(line); ro=r['roofline']
            print(f, round(r['value'],1), 'frac',round(ro.get('frac',0),3),{k:round(v,1) for k,v in ro.get('launch_us',{}).items()})
PY
