#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03h2; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -v amdgpu $O/pytest_gpu.log | tail -2 | cut -c1-300
for m in fp32 fp32_cand fp32_mfma; do
  python bench.py --no-cpu-baseline --matrix-mode $m > $O/bench_$m.log 2>&1
  python bench.py --no-cpu-baseline --matrix-mode $m --overlap 0 > $O/bench_${m}_ov0.log 2>&1
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03h2/bench_*.log')):
    for line in open(f):
        if line.startswith('{"metric'):
            r=json.loads(line); ro=r.get('roofline') or {}
            print(os.path.basename(f), round(r['value'],1), r['config'].get('matrix_mode'), 'frac', round(ro.get('frac',0) or 0,3), ro.get('reset_gate_recomputed_in_candidate_kernel'), ro.get('error'))
PY
python - > $O/nf_fp32_cand.txt 2>&1 <<'PY'
import os, subprocess, sys
PY
timeout 900 python tools/noise_floor.py --n 100 --k 5 --skip-torch > $O/nf_default.txt 2>&1
timeout 900 python tools/noise_floor.py --n 100 --k 5 --skip-torch --matrix-mode fp32_cand > $O/nf_fp32_cand.txt 2>&1
for f in nf_default nf_fp32_cand; do echo "== $f"; grep -v amdgpu $O/$f.txt | grep "^hip\|^# matrix" | cut -c1-200; done
