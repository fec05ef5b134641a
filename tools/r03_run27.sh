#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v amdgpu $O/pytest_gpu.log | tail -2 | cut -c1-300
bash tools/collect_profiles.sh r03 > /dev/null 2>&1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03/bench_*.log')):
    for line in open(f):
        if line.startswith('{"metric'):
            r=json.loads(line); ro=r.get('roofline') or {}
            print(os.path.basename(f), round(r['value'],1), r['unit'], 'frac',round(ro.get('frac',0) or 0,3))
PY
