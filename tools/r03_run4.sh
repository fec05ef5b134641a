#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
timeout 600 python tools/diag_overlap2.py > $O/diag2.txt 2>&1; grep -v amdgpu $O/diag2.txt | cut -c1-400
timeout 900 python tools/parity_slice.py --cache /tmp/slice.npz > $O/slice_ovl.txt 2>&1
timeout 600 python tools/parity_slice.py --cache /tmp/slice.npz --inject 1 > $O/slice_seq.txt 2>&1
for f in $O/slice_*.txt; do echo "== $f"; grep -v "amdgpu.ids" $f | tail -4 | cut -c1-300; done
