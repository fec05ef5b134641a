#!/bin/bash
# round-3 GPU call 2: run-to-run determinism of the f16 GEMMs (which build, which layer), slice parity of the fixed split
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
for v in main asm act0; do
  lib=$R/u-rnn_amd/liburnn_hip.so; [ $v != main ] && lib=$R/u-rnn_amd/liburnn_hip_$v.so
  URNN_LIB=$lib timeout 600 python -m pytest tests/test_hip_rollout.py -m gpu -q -x -k "repeated_launches or bit_stable" > $O/determinism_$v.log 2>&1
  echo "== determinism $v: rc $?"; grep -v amdgpu $O/determinism_$v.log | tail -4
done
timeout 900 python tools/parity_slice.py --cache /tmp/slice.npz > $O/slice_main.txt 2>&1
timeout 600 python tools/parity_slice.py --cache /tmp/slice.npz --inject 1 > $O/slice_inj_main.txt 2>&1
URNN_LIB=$R/u-rnn_amd/liburnn_hip_asm.so timeout 600 python tools/parity_slice.py --cache /tmp/slice.npz --inject 1 > $O/slice_inj_asm.txt 2>&1
URNN_LIB=$R/u-rnn_amd/liburnn_hip_asm.so timeout 600 python tools/parity_slice.py --cache /tmp/slice.npz --inject 1 > $O/slice_inj_asm_again.txt 2>&1
timeout 600 python tools/parity_slice.py --cache /tmp/slice.npz --inject 1 > $O/slice_inj_main_again.txt 2>&1
for f in $O/slice_*.txt; do echo "== $f"; grep -v "amdgpu.ids" $f | tail -4 | cut -c1-400; done
python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0.log 2>&1
URNN_LIB=$R/u-rnn_amd/liburnn_hip_asm.so python bench.py --no-cpu-baseline --overlap 0 > $O/bench_ov0_asm.log 2>&1
grep -o '"value": [0-9.]*' $O/bench_ov0.log $O/bench_ov0_asm.log
