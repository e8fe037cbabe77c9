#!/usr/bin/env python3
"""Run ONE conv kernel shape repeatedly (for rocprofv3 --pmc).  one_conv.py op cin cout k stride hw N G iters"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops
from rot_mvgaze_amd._lib import ConvDesc
op, cin, cout, k, st, hw, N, G, iters = sys.argv[1], *map(int, sys.argv[2:10])
dev = torch.device("cuda:0")
pad = k // 2
d = ConvDesc.make(G, N, hw, hw, cin, cout, k, st, pad)
x = torch.randn(G, N, hw, hw, cin, device=dev)
w = torch.randn(cout, k, k, cin, device=dev) * 0.05
y = torch.empty(G, N, d.ho, d.wo, cout, device=dev)
P, rpp = ops.conv_stats_partials(d)
stats = torch.empty(G, P, 2, cout, device=dev)
gy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
wp = ops.weight_split(d, w, False) if ops.conv_math() == 1 and cout >= 64 else None
xp = ops.split_planes(x) if wp is not None and cin >= 64 else None
for _ in range(iters):
    if op == "fprop" and xp is not None: ops.conv_fprop_pp(d, xp, wp, y, stats)
    elif op == "fprop": ops.conv_fprop(d, x, w, y, None, False, stats)
    elif op == "dgrad": ops.conv_dgrad(d, gy, w, dx)
    else: ops.conv_wgrad(d, x, gy, dw)
torch.cuda.synchronize()
print("done", 2.0 * G * N * d.ho * d.wo * cout * k * k * cin / 1e9, "GFLOP per launch")
