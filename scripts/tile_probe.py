#!/usr/bin/env python3
"""Time fprop / dgrad / wgrad for explicit shapes: tile_probe.py cin,cout,k,stride,hw,N,G [...]
(used to separate main-loop efficiency from tile-quantisation tails)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops
from rot_mvgaze_amd._lib import ConvDesc

dev = torch.device("cuda:0")
iters = int(os.environ.get("ITERS", "10"))

def timeit(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

for a in sys.argv[1:]:
    cin, cout, k, st, hw, N, G = map(int, a.split(","))
    d = ConvDesc.make(G, N, hw, hw, cin, cout, k, st, k // 2)
    x = torch.randn(G, N, hw, hw, cin, device=dev)
    w = torch.randn(cout, k, k, cin, device=dev) * 0.05
    y = torch.empty(G, N, d.ho, d.wo, cout, device=dev)
    P, rpp = ops.conv_stats_partials(d)
    stats = torch.empty(G, P, 2, cout, device=dev)
    gy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * G * N * d.ho * d.wo * cout * k * k * cin
    tf = timeit(lambda: ops.conv_fprop(d, x, w, y, None, False, stats))
    td = timeit(lambda: ops.conv_dgrad(d, gy, w, dx))
    tw = timeit(lambda: ops.conv_wgrad(d, x, gy, dw))
    rows = G * N * d.ho * d.wo
    print(f"{a:28s} rows={rows:7d} fprop {tf*1e3:7.3f} ms {fl/tf/1e12:6.1f} TF | dgrad {td*1e3:7.3f} ms {fl/td/1e12:6.1f} TF | wgrad {tw*1e3:7.3f} ms {fl/tw/1e12:6.1f} TF", flush=True)
