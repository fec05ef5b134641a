#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_rollout.py -m gpu -x -q -s -k "fp32_mfma or full_size_rollout" > $O/pytest_mfma.log 2>&1; echo "rc $?"; grep -v amdgpu $O/pytest_mfma.log | tail -3 | cut -c1-300
python bench.py --no-cpu-baseline --matrix-mode fp32_mfma > $O/bench_fp32_mfma.log 2>&1
python bench.py --no-cpu-baseline --matrix-mode fp32_mfma --overlap 0 > $O/bench_fp32_mfma_ov0.log 2>&1
python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r03g/bench_*.log')):
    for line in open(f):
        if line.startswith('{"metric'):
            r=json.loads(line); ro=r.get('roofline') or {}
            print(os.path.basename(f), round(r['value'],1), r['config'].get('matrix_mode'), ro.get('error'))
PY
