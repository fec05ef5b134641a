#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
python tools/parity_slice.py --t0 0 --n 100 --cache /tmp/s0.npz > $O/slice0_vtdouble.txt 2>&1
for f in vtdouble; do echo "== $f"; grep -v amdgpu $O/slice0_$f.txt | grep -E "^ +[0-9]+ \||max over|frames where|final states|^#" | cut -c1-600; done
