#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
timeout 600 python tools/diag_overlap.py > $O/diag_main.txt 2>&1; grep -v amdgpu $O/diag_main.txt | cut -c1-300
URNN_LIB=$R/u-rnn_amd/liburnn_hip_act0.so timeout 600 python tools/diag_overlap.py > $O/diag_act0.txt 2>&1; grep -v amdgpu $O/diag_act0.txt | cut -c1-300
URNN_TUNE_GATE_ALLN=0 timeout 600 python tools/diag_overlap.py > $O/diag_main_nogroup.txt 2>&1; grep -v amdgpu $O/diag_main_nogroup.txt | cut -c1-300
timeout 900 python -m pytest tests -m gpu -x -q -k "not whole_event and not mid_event and not bit_stable and not full_size" > $O/pytest_subset.log 2>&1; echo "pytest subset rc $?"; tail -5 $O/pytest_subset.log
python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0.log 2>&1
URNN_TUNE_GATE_ALLN=0 python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0_nogroup.log 2>&1
python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c/bench_*.log')):
    for line in open(f):
        if line.startswith('{'):
            r=json.loads(line); ro=r['roofline']
            print(f, round(r['value'],1), 'frac',round(ro.get('frac',0),3),{k:round(v,1) for k,v in ro.get('launch_us',{}).items()})
PY
