#!/bin/bash
# round-3 GPU call 1: f16 x 3 GEMMs -- parity subset, bench (both schedules), per-kernel stats, arithmetic A/B on the mid-event slice
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "not whole_event and not mid_event and not bit_stable and not full_size" > $O/pytest_subset.log 2>&1; echo "pytest subset rc $?"
tail -5 $O/pytest_subset.log
python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1; tail -c 1500 $O/bench_default.log
python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/p_ov0 -o o -- python $R/bench.py --no-cpu-baseline --overlap 0 > $O/bench_overlap0_under_rocprof.log 2>&1
python $R/tools/prof_summary.py /tmp/p_ov0/o_results.db > $O/kernel_stats_overlap0.txt 2>&1
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/p_def -o d -- python $R/bench.py --no-cpu-baseline > $O/bench_default_under_rocprof.log 2>&1
python $R/tools/prof_summary.py /tmp/p_def/d_results.db > $O/kernel_stats.txt 2>&1
cd $R
timeout 900 python tools/parity_slice.py --cache /tmp/slice.npz > $O/slice_main.txt 2>&1
timeout 600 python tools/parity_slice.py --cache /tmp/slice.npz --inject 1 > $O/slice_inj_f16_act1.txt 2>&1
URNN_LIB=$R/u-rnn_amd/liburnn_hip_act0.so timeout 600 python tools/parity_slice.py --cache /tmp/slice.npz --inject 1 > $O/slice_inj_f16_act0.txt 2>&1
URNN_TUNE_F16=0 URNN_LIB=$R/u-rnn_amd/liburnn_hip_bf6.so timeout 600 python tools/parity_slice.py --cache /tmp/slice.npz --inject 1 > $O/slice_inj_bf6_act1.txt 2>&1
URNN_TUNE_F16=0 URNN_LIB=$R/u-rnn_amd/liburnn_hip_r02.so timeout 600 python tools/parity_slice.py --cache /tmp/slice.npz --inject 1 > $O/slice_inj_bf6_act0.txt 2>&1
for f in $O/slice_*.txt; do echo "== $f"; grep -v "amdgpu.ids" $f | tail -4; done
URNN_TUNE_F16=0 URNN_LIB=$R/u-rnn_amd/liburnn_hip_bf6.so python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0_bf6.log 2>&1
for f in $O/*.log; do grep -v "amdgpu.ids\|^W2026\|^E2026\|simple_timer" $f > $f.tmp; mv $f.tmp $f; done
ls -la $O
