#!/usr/bin/env python3
"""Random-size runs of mvg_preprocess_u8hwc_resize against the oracle's restatement of
ToTensor -> Resize(antialias=True) -> Normalize: resize_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops
from oracle import restatement as R
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
bad = 0
for it in range(cases):
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(2, 500)), int(rng.integers(2, 500))
    size = int(rng.choice([7, 24, 64, 224, 225]))
    swap = bool(rng.integers(0, 2))
    u8 = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    dst = torch.empty(n, size, size, 4, device=dev)
    ops.preprocess_u8hwc_resize(torch.from_numpy(u8).to(dev), dst, n, h, w, size, size, MEAN, STD, swap)
    got = dst.cpu().numpy()[..., :3].transpose(0, 3, 1, 2)
    ref = R.preprocess_u8(u8, size, MEAN, STD, swap)
    err = float(np.abs(got - ref).max())
    ok = err <= 5e-6
    bad += not ok
    print(("ok  " if ok else "FAIL"), f"n{n} {h}x{w} -> {size} swap {swap}: max abs err {err:.2e}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
