#!/usr/bin/env python3
"""Randomised sweep of mvg_linear_fprop / mvg_linear_dgrad / mvg_linear_wgrad (weight + bias gradient) against float64:
linear_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops
from rot_mvgaze_amd._lib import ConvDesc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")

def rel(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()

bad = 0
for it in range(cases):
    rows = int(rng.choice([1, 2, 3, 7, 64, 127, 128, 129, 384, 640, 1536, 3584, int(rng.integers(1, 1500)), int(rng.integers(1500, 6000))]))
    fin = 4 * int(rng.integers(1, 520))
    fout = 4 * int(rng.integers(1, 520))
    x = torch.from_numpy(rng.standard_normal((rows, fin)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((fout, fin)) / np.sqrt(fin)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(fout).astype(np.float32))
    gy = torch.from_numpy(rng.standard_normal((rows, fout)).astype(np.float32))
    mask = torch.from_numpy(rng.standard_normal((rows, fin)).astype(np.float32))
    add = torch.from_numpy(rng.standard_normal((rows, fin)).astype(np.float32))
    xd, wd, bd, gyd = x.to(dev), w.to(dev), b.to(dev), gy.to(dev)
    y = torch.full((rows, fout), float("nan"), device=dev)
    ops.linear_fprop(xd, wd, bd, True, y, rows, fin, fout)
    e = [rel(y, torch.relu(x.double() @ w.double().T + b.double()))]
    dx = add.clone().to(dev)
    ops.linear_dgrad(gyd, wd, mask.to(dev), dx, dx, rows, fin, fout)
    e.append(rel(dx, (gy.double() @ w.double()) * (mask > 0) + add.double()))
    dw = torch.full((fout, fin), float("nan"), device=dev)
    db = torch.full((fout,), float("nan"), device=dev)
    ops.linear_wgrad(xd, gyd, dw, db, rows, fin, fout, False)       # weight and bias gradient in one launch
    e.append(rel(dw, gy.double().T @ x.double()))
    e.append(rel(db, gy.double().sum(0)))
    ok = all(v == v and v <= 3e-5 for v in e)
    bad += not ok
    print(("ok  " if ok else "FAIL"), f"rows {rows} fin {fin} fout {fout}:", " ".join(f"{v:.1e}" for v in e), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
