#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/collect_profiles.sh r03 > /dev/null 2>&1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03/bench_*.log')):
    for line in open(f):
        if line.startswith('{"metric'):
            r=json.loads(line); ro=r.get('roofline') or {}
            print(os.path.basename(f), round(r['value'],1), r['unit'], 'frac',round(ro.get('frac',0) or 0,3))
PY
