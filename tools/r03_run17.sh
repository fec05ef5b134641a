#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03q; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in lite64 ukea; do
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps_$c -o o -- python $R/bench.py --config $c --no-cpu-baseline --overlap 0 --steps 3000 --warmup 100 > $O/prof_$c.log 2>&1
python $R/tools/prof_summary.py /tmp/ps_$c/o_results.db > $O/kernel_stats_$c.txt 2>&1
done
head -40 $O/kernel_stats_lite64.txt | cut -c1-170
grep '^{' $O/prof_lite64.log | cut -c1-400
