#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
for c in lite64 ukea futian lite128; do
  python bench.py --config $c --no-cpu-baseline > $O/cfg_$c.log 2>&1
  python bench.py --config $c --no-cpu-baseline --overlap 0 > $O/cfg_${c}_ov0.log 2>&1
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03q/cfg_*.log')):
    for line in open(f):
        if line.startswith('{"metric'):
            r=json.loads(line); print(os.path.basename(f), round(r['value'],1), r['unit'], r['ms_per_step'])
PY
