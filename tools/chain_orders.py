"""Development aid: whole-rollout frames/s for every enqueue order of the kernel chains' segments (URNN_TUNE_CHAIN_ORDER).
Result (round 2, two chains -- the head in front of the encoder pass): every order with the head first within noise (1 275-1 283), orders
that start with the decoder lose 4-11 %.  Round 4, the head on a chain of its own: decoder first is the best (profiles/r04_enqueue_order.txt)."""
import itertools, subprocess, os, re, sys
orders=set()
for pos in itertools.combinations(range(6),3):
    s=['D']*6
    for p in pos: s[p]='E'
    base=''.join(s)
    for hp in (0,1,2,3):
        orders.add(base[:hp]+'H'+base[hp:])
res=[]
for o in sorted(orders):
    env=dict(os.environ, URNN_TUNING="1", URNN_TUNE_CHAIN_ORDER=o)
    out=subprocess.run([sys.executable,'bench.py','--no-cpu-baseline','--no-long-run','--steps','360','--warmup','36'],env=env,capture_output=True,text=True).stdout
    m=re.search(r'"value": ([0-9.]+)',out)
    v=float(m.group(1)) if m else 0
    res.append((v,o)); print(o,v,flush=True)
res.sort(reverse=True)
print("BEST",res[:8])
