#!/bin/bash
# do the two kernel chains overlap better when every persistent GEMM takes only part of the chip?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
run() { name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-long-run > $O/b_$name.log 2>&1
  python - "$O/b_$name.log" "$name" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        r=json.loads(line); ro=r['roofline']
        print(f"{sys.argv[2]:24s} {r['value']:8.1f} frames/s  gate launches us", {k:round(v,1) for k,v in ro.get('launch_us',{}).items()})
PY
}
run base A=1
run cus128 URNN_TUNE_CUS=128
run cus144 URNN_TUNE_CUS=144
run cus160 URNN_TUNE_CUS=160
run cus192 URNN_TUNE_CUS=192
run cus96 URNN_TUNE_CUS=96
run base2 A=1
