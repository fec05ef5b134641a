#!/bin/bash
# Same-box A/B of library variants: tools/ab.sh name1 name2 ...   (name = suffix of u-rnn_amd/liburnn_hip_<name>.so; "cur" = the product library)
# Alternates the variants REPS times: one-chain bench (per-cell launch times) and the default three-chain bench.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
REPS=${REPS:-2}
PY='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); ks=r["roofline"].get("kernels") or []
        print(sys.argv[1], "%7.1f frames/s" % r["value"], " ".join("%s %s" % (d["family"][:5], {c:round(v,1) for c,v in (d.get("launch_us") or {}).items()}) for d in ks))'
for rep in $(seq $REPS); do
  for v in "$@"; do
    lib=$R/u-rnn_amd/liburnn_hip_$v.so; [ "$v" = cur ] && lib=$R/u-rnn_amd/liburnn_hip.so
    URNN_LIB=$lib python $R/bench.py --no-cpu-baseline --no-long-run --overlap 0 2>/dev/null | python -c "$PY" "$v one-chain  "
    URNN_LIB=$lib python $R/bench.py --no-cpu-baseline --no-long-run 2>/dev/null | python -c "$PY" "$v three-chain"
  done
done
