#!/bin/bash
# knob sweep after the f16 x 3 switch (one-chain schedule, same box): which tile / kernel choices moved?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --overlap 0 --no-long-run > $O/b_$name.log 2>&1
  python - "$O/b_$name.log" "$name" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        r=json.loads(line); ro=r['roofline']
        print(f"{sys.argv[2]:34s} {r['value']:8.1f} frames/s  gate launches us", {k:round(v,1) for k,v in ro.get('launch_us',{}).items()})
PY
}
run base A=1
run base_again A=1
run alln0 URNN_TUNE_GATE_ALLN=0
run pbcand2 URNN_TUNE_PB_CAND=2
run pbcand2_alln0 URNN_TUNE_PB_CAND=2 URNN_TUNE_GATE_ALLN=0
run small70k URNN_TUNE_SMALL=70000
run small0 URNN_TUNE_SMALL=0
run wpb4 URNN_TUNE_WPB=4
run ring6 URNN_TUNE_RING=6
run pbconv2 URNN_TUNE_PB_CONV=2
run candnb1 URNN_TUNE_CAND_NB=1
run convnb3 URNN_TUNE_CONV_NB3=3
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o o -- env URNN_TUNE_PB_CAND=2 python $R/bench.py --no-cpu-baseline --overlap 0 --no-long-run > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p1/o_results.db > $O/kernel_stats_pbcand2.txt 2>&1
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o o -- python $R/bench.py --no-cpu-baseline --overlap 0 --no-long-run > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p2/o_results.db > $O/kernel_stats_base.txt 2>&1
head -24 $O/kernel_stats_base.txt | cut -c1-160
