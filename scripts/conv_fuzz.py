#!/usr/bin/env python3
"""Randomised shape sweep of mvg_conv_fprop / dgrad / wgrad against torch float64 on the host
(beyond the fixed cases of tests/test_kernels_gpu.py): conv_fuzz.py [cases] [seed] [big]
("big": batch / image sizes that reach the stream-K and multi-round launches)"""
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
RTOL = 2e-5

def rel(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()

bad = 0
for it in range(cases):
    k = int(rng.choice([1, 1, 3, 3, 3, 5, 7]))
    st = int(rng.choice([1, 1, 2]))
    pad = int(rng.integers(0, k // 2 + 1))
    cin = int(rng.choice([4, 16, 32, 64, 128, 256]))
    cout = int(rng.choice([32, 64, 128, 256]))
    G, N = int(rng.integers(1, 4)), int(rng.integers(1, 10))
    H, W = int(rng.integers(k, 41)), int(rng.integers(k, 41))
    if len(sys.argv) > 3 and sys.argv[3] == "big":
        cin = int(rng.choice([16, 32, 64, 128, 256]))
        cout = int(rng.choice([64, 128, 256, 512]))
        k = int(rng.choice([1, 3, 3, 5]))
        pad = k // 2
        G, N = int(rng.integers(1, 3)), int(rng.integers(8, 40))
        H, W = int(rng.integers(12, 58)), int(rng.integers(12, 58))
    d = ConvDesc.make(G, N, H, W, cin, cout, k, st, pad)
    if d.ho < 1 or d.wo < 1:
        continue
    x = torch.from_numpy(rng.standard_normal((G, N, cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    xr = x.reshape(G * N, cin, H, W).double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, st, pad)
    gy = torch.from_numpy(rng.standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(gy.double())
    xd = x.permute(0, 1, 3, 4, 2).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    y = torch.empty(G, N, d.ho, d.wo, cout, device=dev)
    P, rpp = ops.conv_stats_partials(d)
    stats = torch.full((G, P, 2, cout), float("nan"), device=dev)
    ops.conv_fprop(d, xd, wd, y, None, False, stats)
    y_ref = yr.detach().float().reshape(G, N, cout, d.ho, d.wo).permute(0, 1, 3, 4, 2)
    e = [rel(y, y_ref)]
    e.append(rel(stats[:, :, 0].sum(1), y_ref.reshape(G, -1, cout).sum(1)) if True else 0.0)
    gyd = gy.reshape(G, N, cout, d.ho, d.wo).permute(0, 1, 3, 4, 2).contiguous().to(dev)
    dx = torch.full((G, N, H, W, cin), float("nan"), device=dev)
    ops.conv_dgrad(d, gyd, wd, dx)
    e.append(rel(dx, xr.grad.float().reshape(G, N, cin, H, W).permute(0, 1, 3, 4, 2)))
    dw = torch.full((cout, k, k, cin), float("nan"), device=dev)
    ops.conv_wgrad(d, xd, gyd, dw, False)
    e.append(rel(dw, wr.grad.float().permute(0, 2, 3, 1)))
    ok = all(v == v and v <= (5e-4 if i == 1 else RTOL) for i, v in enumerate(e))
    bad += not ok
    print(("ok  " if ok else "FAIL"), f"G{G} N{N} {H}x{W} cin{cin} cout{cout} k{k} s{st} p{pad}:",
          " ".join(f"{v:.1e}" for v in e), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
