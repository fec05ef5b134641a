#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03o; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "not whole_event and not mid_event and not full_size" > $O/pytest_subset.log 2>&1; echo "pytest subset rc $?"; grep -v amdgpu $O/pytest_subset.log | tail -3 | cut -c1-300
DIAG_N=3 DIAG_M=3000 timeout 900 python tools/diag_dec1.py > $O/dec1.txt 2>&1; grep -v amdgpu $O/dec1.txt | cut -c1-300 | tail -1
python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0.log 2>&1
python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03o/bench_*.log')):
    for line in open(f):
        if line.startswith('{'):
            r=json.loads(line); ro=r.get('roofline') or {}
            print(f, round(r['value'],1), r['unit'], 'frac',round(ro.get('frac',0),3),{k:round(v,1) for k,v in (ro.get('launch_us') or {}).items()})
PY
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o o -- python $R/bench.py --no-cpu-baseline --overlap 0 --no-long-run > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p2/o_results.db > $O/kernel_stats_ov0.txt 2>&1
head -24 $O/kernel_stats_ov0.txt | cut -c1-160
