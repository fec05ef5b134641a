#!/usr/bin/env python
"""HBM stream rates on this box: pure write (fill), pure read (sum), copy.  Development tool."""
import torch
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters
for mb in (64, 192, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    tw = timeit(lambda: x.fill_(1.5))
    tr = timeit(lambda: x.sum())
    tc = timeit(lambda: y.copy_(x))
    ta = timeit(lambda: torch.add(x, y, out=y))
    print(f"{mb:5d} MB: fill {mb/1024/tw/1.024:6.2f} TB/s ({tw*1e6:7.1f} us)  sum {mb/1024/tr/1.024:6.2f} TB/s  copy {2*mb/1024/tc/1.024:6.2f} TB/s  add(2r1w) {3*mb/1024/ta/1.024:6.2f} TB/s")
