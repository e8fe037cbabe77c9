#!/usr/bin/env python3
"""Randomised sweep of the split-operand kernels (mvg_conv_fprop_split / dgrad_split / wgrad_split) against torch float64:
random shapes (ragged maps, strides, 64-column tiles, one-row tiles), random MAGNITUDES of the three operands (activations
1e-2 .. 1e2 unscaled, weights 1e-4 .. 1e1 through the prep's scale, gradients 1e-9 .. 1 through a power-of-two scale),
relative L2 against fp64 <= 2e-6 like tests/test_split_gpu.py.  split_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
import rot_mvgaze_amd
from rot_mvgaze_amd import ops
from rot_mvgaze_amd._lib import ConvDesc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
TOL = 2e-6


def rel(got, ref):
    return float((got.double() - ref).norm() / (ref.norm() + 1e-300))


bad = 0
for it in range(cases):
    k = int(rng.choice([1, 1, 3, 3]))
    st = int(rng.choice([1, 1, 2]))
    pad = k // 2 if rng.random() < 0.8 else 0
    if k == 3:
        cin, cout = int(rng.choice([32, 64, 128, 256])), int(rng.choice([32, 64, 128, 256]))
    else:
        cin, cout = int(rng.integers(1, 9)) * 32, int(rng.integers(1, 9)) * 32
    G, N = int(rng.integers(1, 4)), int(rng.integers(1, 12))
    H, W = int(rng.integers(k, 37)), int(rng.integers(k, 37))
    d = ConvDesc.make(G, N, H, W, cin, cout, k, st, pad)
    if d.ho < 1 or d.wo < 1:
        continue
    ax, aw, ag = 10.0 ** rng.uniform(-2, 2), 10.0 ** rng.uniform(-4, 1), 10.0 ** rng.uniform(-9, 0)
    x = torch.relu(torch.randn(G, N, H, W, cin, device=dev)) * ax
    w = torch.randn(cout, k, k, cin, device=dev) * (aw / (k * k * cin) ** 0.5)
    gy = torch.randn(G, N, d.ho, d.wo, cout, device=dev) * ag
    add = torch.randn_like(x) * (ag * aw)
    gscale = 2.0 ** np.floor(np.log2(2.0 ** 14 / float(gy.abs().max())))
    xs, gys = ops.split_f32(x), ops.split_f32(gy, gscale)
    wk, wt = ops.split_weights(d, w, True)
    xr = x.double().view(G * N, H, W, cin).permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, st, pad)
    yr.backward(gy.double().view(G * N, d.ho, d.wo, cout).permute(0, 3, 1, 2))
    y = torch.empty(G, N, d.ho, d.wo, cout, device=dev)
    ops.conv_fprop_split(d, xs, wk, y, None)
    dx = torch.full_like(x, float("nan"))
    ops.conv_dgrad_split(d, gys, wt, dx, add)
    dw = torch.full_like(w, float("nan"))
    ops.conv_wgrad_split(d, xs, gys, dw)
    e = [rel(y, yr.detach().permute(0, 2, 3, 1).reshape(y.shape)),
         rel(dx, xr.grad.permute(0, 2, 3, 1).reshape(x.shape) + add.double()),
         rel(dw, wr.grad.permute(0, 2, 3, 1))]
    ok = all(v == v and v <= TOL for v in e)
    bad += not ok
    print(("ok  " if ok else "FAIL"), f"G{G} N{N} {H}x{W} cin{cin} cout{cout} k{k} s{st} p{pad} |x|{ax:.0e} |w|{aw:.0e} |g|{ag:.0e}:",
          " ".join(f"{v:.1e}" for v in e), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
