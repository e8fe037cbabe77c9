#!/usr/bin/env python3
"""mvg_multi_erase_nchw against F.interpolate (the reference's own mask upsampling, utils/augment.py:21)
for random image sizes and grid sizes: erase_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
import rot_mvgaze_amd
from rot_mvgaze_amd.augment import RandomMultiErasing
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
aug = RandomMultiErasing(p=0.5, proportion=[0.5, 0.6], dot_size=[0.05, 0.3])
bad = 0
for it in range(cases):
    B, C = int(rng.integers(1, 6)), int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    img = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32))
    draws = []
    for b in range(B):
        g = int(rng.choice([0, 1, 2, 3, 5, 7, 13, 20, int(rng.integers(1, 40))]))
        draws.append((g, (torch.from_numpy(rng.random((g, g)).astype(np.float32)) > 0.5).float() if g else torch.zeros(0, 0)))
    ref = img.clone()
    for b, (g, m) in enumerate(draws):
        if g:
            ref[b] *= F.interpolate(m[None, None], (H, W)).squeeze(0).squeeze(0)
    got = aug.apply(img.clone().to(dev), draws).cpu()
    ok = torch.equal(got, ref)
    bad += not ok
    print(("ok  " if ok else "FAIL"), f"B{B} C{C} {H}x{W} grids {[g for g, _ in draws]}", "" if ok else f"mismatches {(got != ref).sum().item()}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
