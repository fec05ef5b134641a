#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v amdgpu $O/pytest_gpu.log | tail -3 | cut -c1-300
timeout 600 python tools/stress_overlap.py > $O/stress.txt 2>&1; tail -3 $O/stress.txt | cut -c1-300
python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0.log 2>&1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03v/bench_*.log')):
    for line in open(f):
        if line.startswith('{"metric'):
            r=json.loads(line); ro=r.get('roofline') or {}
            print(os.path.basename(f), round(r['value'],1), 'frac',round(ro.get('frac',0) or 0,3),{k:round(v,1) for k,v in (ro.get('launch_us') or {}).items()}, ro.get('reset_gate_recomputed_in_candidate_kernel'), ro.get('error'))
PY
