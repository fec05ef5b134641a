#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box: raw databases stay in /tmp, text summaries go to gpurun_out/$1.
# usage (through gpurun): bash tools/collect_profiles.sh r06   (then copy gpurun_out/r06/* to profiles/r06_*, pmc_kernels.json + pmc_train.json to profiles/)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}
O=$R/gpurun_out/$TAG; P=/tmp/prof_$TAG
mkdir -p $O $P; cd /tmp; export TMPDIR=/tmp
# one chain, eager (every dispatch attributed) -- with the kernel selection of the benchmarked overlapped schedule (URNN_TUNE_COOP_BIG=0)
export URNN_TUNING=1 URNN_TUNE_COOP_BIG=0
CMD="python $R/bench.py --steps 14 --warmup 2 --no-cpu-baseline --no-long-run --overlap 0 --no-graph"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 420 rocprofv3 --kernel-trace --pmc $pass -d $P/pmc_$name -o p -- $CMD > $P/pmc_$name.log 2>&1
  { echo "# rocprofv3 --kernel-trace --pmc $pass -- $CMD"; python $R/tools/pmc_summary.py $P/pmc_$name/p_results.db "" --frames=-1; } > $O/pmc_$name.txt 2>&1
done
python $R/tools/make_pmc_json.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_kernels.json "$TAG" > /dev/null 2>$O/make_pmc_json.err
unset URNN_TUNE_COOP_BIG
# the ONE-chain schedule as it runs by default (cooperative cells wherever planned: urnn_coop_tiles.hip at half resolution): its byte counters
for name in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --kernel-trace --pmc $name -d $P/pmc1_$name -o p -- python $R/tools/one_chain_frames.py 12 > $P/pmc1_$name.log 2>&1
  { echo "# rocprofv3 --kernel-trace --pmc $name -- python tools/one_chain_frames.py 12   (one chain, eager, default kernel selection: cooperative cells; 12 frames)"; python $R/tools/pmc_summary.py $P/pmc1_$name/p_results.db "" --frames=12; } > $O/pmc_one_chain_$name.txt 2>&1
done
# the training step's byte counters (two windows of 4 timesteps, eager so that every dispatch is attributed)
TCMD="python $R/bench.py --mode train --steps 4 --warmup 4 --no-graph"
for name in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $name -d $P/pmct_$name -o p -- $TCMD > $P/pmct_$name.log 2>&1
  { echo "# rocprofv3 --kernel-trace --pmc $name -- $TCMD   (8 training timesteps: per-'frame' totals below are per timestep)"; python $R/tools/pmc_summary.py $P/pmct_$name/p_results.db "" --frames=8 --all-kernels; } > $O/pmc_train_$name.txt 2>&1
done
python $R/tools/make_pmc_json.py $O/pmc_train_FETCH_SIZE.txt $O/pmc_train_WRITE_SIZE.txt $O/pmc_train.json "$TAG train" > /dev/null 2>>$O/make_pmc_json.err
timeout 420 rocprofv3 --kernel-trace --stats -d $P/stats_default -o d -- python $R/bench.py --no-cpu-baseline > $O/bench_default_under_rocprof.log 2>&1
python $R/tools/prof_summary.py $P/stats_default/d_results.db > $O/kernel_stats.txt 2>&1
timeout 420 rocprofv3 --kernel-trace --stats -d $P/stats_ov0 -o o -- python $R/bench.py --no-cpu-baseline --overlap 0 > $O/bench_overlap0_under_rocprof.log 2>&1
python $R/tools/prof_summary.py $P/stats_ov0/o_results.db > $O/kernel_stats_overlap0.txt 2>&1
timeout 300 python $R/bench.py > $O/bench_default.log 2>&1
timeout 200 python $R/bench.py --overlap 0 --no-cpu-baseline > $O/bench_overlap0.log 2>&1
timeout 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.log 2>&1
# the other BASELINE shapes: one event per GPU and -- configs[4] is BATCHED inference -- eight events per GPU
PYP='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print(sys.argv[1], round(r["value"],1), "frames/s", round(r["ms_per_step"]*1000,1), "us per step (one frame per event)")'
{ for c in lite64 ukea lite128 futian mixed; do
    timeout 300 python $R/bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "$PYP" "$c batch 1"
  done
  for c in lite64 ukea lite128 mixed; do
    timeout 300 python $R/bench.py --config $c --batch 8 --no-cpu-baseline 2>/dev/null | python -c "$PYP" "$c batch 8"
  done; } > $O/bench_configs.txt 2>&1
# the request-shape microbenchmark (tools/ubench/operand_stream.hip)
( cd $R/tools/ubench && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/operand_stream operand_stream.hip && /tmp/operand_stream ) > $O/operand_stream.txt 2>&1
timeout 300 python $R/tools/kernel_bench.py > $O/kernel_bench.txt 2>&1
timeout 300 python $R/bench.py --mode train > $O/bench_train.log 2>&1
timeout 300 python $R/bench.py --mode train --dtype bf16 > $O/bench_train_bf16.log 2>&1
timeout 300 python $R/bench.py --mode train --seq-num 12 > $O/bench_train_seq12.log 2>&1
URNN_TUNING=1 URNN_TUNE_TRAIN_CHAINS=0 URNN_TUNE_TRAIN_BWD_CHAINS=0 timeout 300 python $R/bench.py --mode train > $O/bench_train_one_chain.log 2>&1   # the window step by step on one stream
timeout 420 rocprofv3 --kernel-trace --stats -d $P/stats_train -o t -- python $R/bench.py --mode train > /dev/null 2>&1
python $R/tools/prof_summary.py $P/stats_train/t_results.db > $O/train_kernel_stats.txt 2>&1
python $R/tools/wgrad_trace.py $P/stats_train/t_results.db > $O/train_wgrad_launches.txt 2>&1
# the bf16 arm's kernels beside the fp32 arm's (why bf16 mode is only ~7 % faster)
timeout 420 rocprofv3 --kernel-trace --stats -d $P/stats_train_bf16 -o t -- python $R/bench.py --mode train --dtype bf16 > /dev/null 2>&1
python $R/tools/prof_summary.py $P/stats_train_bf16/t_results.db > $O/train_kernel_stats_bf16.txt 2>&1
for f in $O/*.log; do grep -v "amdgpu.ids\|^W2026\|^E2026\|simple_timer" $f > $f.tmp; mv $f.tmp $f; done
ls -la $O
