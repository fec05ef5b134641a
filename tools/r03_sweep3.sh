#!/bin/bash
# ring-depth sweep for the full-resolution gate / candidate GEMMs (development knobs URNN_TUNE_RING, URNN_TUNE_RING_CAND)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03p; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --overlap 0 --no-long-run > $O/b_$name.log 2>&1; env "$@" python bench.py --no-cpu-baseline --no-long-run > $O/d_$name.log 2>&1; }
run base A=1
run ring12 URNN_TUNE_RING=12
run ring16 URNN_TUNE_RING=16
run cand12 URNN_TUNE_RING_CAND=12
run both URNN_TUNE_RING=16 URNN_TUNE_RING_CAND=12
run base2 A=1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03p/*.log'), key=os.path.getmtime):
    for line in open(f):
        if line.startswith('{"metric'):
            r=json.loads(line); ro=r.get('roofline') or {}
            print(os.path.basename(f), round(r['value'],1), 'frac',round(ro.get('frac',0) or 0,3),{k:round(v,1) for k,v in (ro.get('launch_us') or {}).items()})
PY
