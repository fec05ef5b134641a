import sys, faulthandler
faulthandler.enable()
import torch
dev = torch.device("cuda:0")
a = torch.zeros(1 << 20, device=dev); b = torch.zeros(1 << 20, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
variant = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
keep = []
def body():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    ev1 = ev2 = None
    for j in range(K):
        with torch.cuda.stream(s1):
            if ev2 is not None and variant in ("both", "s1waits"):
                s1.wait_event(ev2)
            a.add_(1.0); a.mul_(1.0001)
            e1 = torch.cuda.Event(); e1.record(s1); keep.append(e1)
        with torch.cuda.stream(s2):
            if ev1 is not None and variant in ("both", "s2waits"):
                s2.wait_event(ev1)
            b.add_(2.0); b.mul_(1.0001)
            e2 = torch.cuda.Event(); e2.record(s2); keep.append(e2)
        ev1, ev2 = e1, e2
    cur.wait_stream(s1); cur.wait_stream(s2)
body(); torch.cuda.synchronize(); print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
print("captured", variant, K, flush=True)
g.replay(); torch.cuda.synchronize(); print("replayed", float(a[0]), float(b[0]), flush=True)
