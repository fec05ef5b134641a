#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03s; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "not whole_event and not mid_event and not full_size and not train" > $O/pytest_subset.log 2>&1; echo "pytest subset rc $?"; grep -v amdgpu $O/pytest_subset.log | tail -3 | cut -c1-300
for c in location1 lite64 ukea lite128; do
  python bench.py --config $c --no-cpu-baseline > $O/cfg_$c.log 2>&1
  python bench.py --config $c --no-cpu-baseline --overlap 0 > $O/cfg_${c}_ov0.log 2>&1
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03s/cfg_*.log')):
    for line in open(f):
        if line.startswith('{"metric'):
            r=json.loads(line); print(os.path.basename(f), round(r['value'],1), r['unit'], round(r['ms_per_step'],4))
PY
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o o -- python $R/bench.py --no-cpu-baseline --overlap 0 --no-long-run > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p2/o_results.db > $O/kernel_stats_ov0.txt 2>&1
grep "small_cell\|blend" $O/kernel_stats_ov0.txt | cut -c1-160
