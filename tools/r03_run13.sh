#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v amdgpu $O/pytest_gpu.log | tail -6 | cut -c1-300
python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0.log 2>&1
URNN_TUNE_FUSE_HEAD=0 python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0_nofusehead.log 2>&1
python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
URNN_TUNE_FUSE_HEAD=0 python bench.py --no-cpu-baseline > $O/bench_default_nofusehead.log 2>&1
python bench.py --mode train > $O/bench_train.log 2>&1
python bench.py --mode train --dtype bf16 > $O/bench_train_bf16.log 2>&1
python bench.py --mode train --seq-num 12 > $O/bench_train_seq12.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03m/bench_*.log')):
    for line in open(f):
        if line.startswith('{'):
            r=json.loads(line); ro=r.get('roofline') or {}
            print(f, round(r['value'],1), r['unit'], 'frac',round(ro.get('frac',0),3),{k:(round(v,1) if not isinstance(v,dict) else round(v['us'],1)) for k,v in (ro.get('launch_us') or ro.get('launches') or {}).items()})
PY
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/pt -o t -- python $R/bench.py --mode train > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/pt/t_results.db > $O/train_kernel_stats.txt 2>&1
head -30 $O/train_kernel_stats.txt | cut -c1-150
