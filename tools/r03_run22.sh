#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/collect_profiles.sh r03 > /dev/null 2>&1
O=$R/gpurun_out/r03
for c in lite64 ukea futian lite128; do
  python bench.py --config $c --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$c', round(r['value'],1), r['unit'], 'ms/frame', round(r['ms_per_step'],4))"
done > $O/bench_configs.txt 2>&1
python bench.py --config mixed --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('mixed', round(r['value'],1), r['unit'])" >> $O/bench_configs.txt 2>&1
URNN_TUNE_FUSED_R=0 python bench.py --no-cpu-baseline > $O/bench_default_three_pass_cell.log 2>&1
URNN_TUNE_FUSED_R=0 python bench.py --no-cpu-baseline --overlap 0 > $O/bench_overlap0_three_pass_cell.log 2>&1
for f in $O/*.log; do grep -v "amdgpu.ids\|^W2026\|^E2026\|simple_timer" $f > $f.tmp; mv $f.tmp $f; done
ls $O; cat $O/bench_configs.txt
