#!/usr/bin/env python3
"""Time the fusion block's Linear GEMMs (fprop / dgrad / wgrad mvg_linear_fprop / mvg_linear_dgrad / mvg_conv_wgrad):
linear_bench.py rows [in,out ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops
from rot_mvgaze_amd._lib import ConvDesc

dev = torch.device("cuda:0")
iters = int(os.environ.get("ITERS", "50"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shapes = [tuple(map(int, a.split(","))) for a in sys.argv[2:]] or [(512, 1536), (1536, 1536), (2048, 2048), (2048, 1536), (2048, 512)]

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

tot = [0.0, 0.0, 0.0]
for cin, cout in shapes:
    d = ConvDesc.make(1, rows, 1, 1, cin, cout, 1, 1, 0)
    x = torch.randn(1, rows, 1, 1, cin, device=dev)
    w = torch.randn(cout, 1, 1, cin, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    y = torch.empty(1, rows, 1, 1, cout, device=dev)
    gy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * rows * cin * cout
    tf = timeit(lambda: ops.linear_fprop(x, w, b, True, y, rows, cin, cout))
    td = timeit(lambda: ops.linear_dgrad(gy, w, None, None, dx, rows, cin, cout))
    tw = timeit(lambda: ops.conv_wgrad(d, x, gy, dw))
    for i, t in enumerate((tf, td, tw)): tot[i] += t
    print(f"{rows}x{cin}->{cout}: fprop {tf*1e6:6.1f} us {fl/tf/1e12:5.1f} TF {cin*cout*4/tf/1e12:4.2f} TB/s | dgrad {td*1e6:6.1f} us {fl/td/1e12:5.1f} TF | wgrad {tw*1e6:6.1f} us {fl/tw/1e12:5.1f} TF", flush=True)
print("total us: fprop %.1f dgrad %.1f wgrad %.1f" % tuple(t * 1e6 for t in tot))
