#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for b in 512 384 256 768; do URNN_TUNE_WGRAD_BLOCKS=$b python bench.py --mode train 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('wgrad blocks $b:', round(r['value'],1), r['unit'], round(r['ms_per_step'],3), 'roofline frac', round(r['roofline']['frac'],3))"; done
