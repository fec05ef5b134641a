#!/bin/bash
# kernel statistics + a timeline excerpt of the 64x64 rollout on three chains and on the level pipeline -> gpurun_out/r06/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lv in 0 1; do
  rm -rf /tmp/pl_$lv
  URNN_TUNING=1 URNN_TUNE_LEVELS=$lv timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pl_$lv -o o -- python $R/bench.py --config lite64 --no-cpu-baseline > /tmp/pl_$lv.log 2>&1
  name=$([ $lv = 0 ] && echo three_chains || echo level_pipeline)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config lite64 --no-cpu-baseline   (URNN_TUNING=1 URNN_TUNE_LEVELS=$lv: $name)"
    grep '^{' /tmp/pl_$lv.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('# bench line under the profiler (kernel tracing slows the launch-bound small grids; without it: profiles/r06_bench_configs.txt):', round(r['value'],1), 'frames/s,', round(r['ms_per_step']*1e3,1), 'us per step, kernel chains', r['config']['kernel_chains'])"
    python $R/tools/prof_summary.py /tmp/pl_$lv/o_results.db | head -16 | cut -c1-175
    echo "# timeline of 500 us of the steady state (tools/chain_timeline.py)"
    python $R/tools/chain_timeline.py /tmp/pl_$lv/o_results.db 500 | cut -c1-150; } > $O/kernel_stats_lite64_$name.txt 2>&1
done
tail -3 $O/kernel_stats_lite64_level_pipeline.txt
