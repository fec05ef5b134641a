#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_rollout.py -m gpu -x -q -s -k "fused_reset_gate or full_size_rollout" > $O/pytest_fused.log 2>&1; echo "fused test rc $?"; grep -v amdgpu $O/pytest_fused.log | grep -E "fused vs|passed|failed|Error|assert" | head -20 | cut -c1-300
for v in 1 0; do
  URNN_TUNE_FUSED_R=$v python bench.py --no-cpu-baseline > $O/bench_default_f$v.log 2>&1
  URNN_TUNE_FUSED_R=$v python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0_f$v.log 2>&1
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03u/bench_*.log')):
    for line in open(f):
        if line.startswith('{"metric'):
            r=json.loads(line); ro=r.get('roofline') or {}
            print(os.path.basename(f), round(r['value'],1), 'frac',round(ro.get('frac',0) or 0,3),{k:round(v,1) for k,v in (ro.get('launch_us') or {}).items()})
PY
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o o -- python $R/bench.py --no-cpu-baseline --overlap 0 --no-long-run > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p2/o_results.db > $O/kernel_stats_ov0.txt 2>&1
head -8 $O/kernel_stats_ov0.txt | cut -c1-160
