#!/usr/bin/env python3
"""rot_mvgaze_amd.optim.Adam (one fused launch over the arenas) against torch.optim.Adam fed with the same
gradients, random hyper-parameters, several steps incl. lr changes (CyclicLR-like): adam_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_model_gpu as T
from rot_mvgaze_amd.optim import Adam
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    lr = float(10 ** rng.uniform(-6, -2)); wd = float(rng.choice([0.0, 1e-6, 1e-3, 0.1]))
    betas = (float(rng.choice([0.0, 0.5, 0.9])), float(rng.choice([0.9, 0.99, 0.999]))); eps = float(rng.choice([1e-8, 1e-5]))
    m = T.build(18)
    opt = Adam(m.parameters(), lr=lr, betas=betas, eps=eps, weight_decay=wd)
    names = [k for k, p in m.named_parameters()]
    cpu = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in m.named_parameters()}
    ref = torch.optim.Adam([cpu[k] for k in names if "fc." not in k], lr=lr, betas=betas, eps=eps, weight_decay=wd)
    worst, wkey = 0.0, ""
    for step in range(4):
        d = m(T.inputs(3, 64, seed=int(rng.integers(0, 1000))))
        T.metrics()(d).backward()
        for k, p in m.named_parameters():
            cpu[k].grad = None if p.grad is None else (p.grad.detach().contiguous() if p.grad.dim() == 4 else p.grad.detach()).cpu().clone()
        new_lr = lr * float(rng.uniform(0.1, 3.0))                 # scheduler step between optimizer steps
        for g in opt.param_groups: g["lr"] = new_lr
        for g in ref.param_groups: g["lr"] = new_lr
        opt.step(); ref.step()
        opt.zero_grad(); 
        for k, p in m.named_parameters():
            a, b = p.detach().cpu().double(), cpu[k].detach().double()
            e = ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
            if e > worst:
                worst, wkey = e, f"{k} step {step}"
    # elements without a gradient under strong weight decay take steps 10x their own size (Adam normalises
    # wd * p): their fp32 rounding differences grow ~10x per step on both sides - measured 4e-5 of the tensor's
    # max after 4 steps at lr 1e-2, wd 0.1; every other case stays below 3e-7
    ok = worst <= (5e-6 if lr * wd < 1e-5 else 2e-4)
    bad += not ok
    print(("ok  " if ok else "FAIL"), f"lr {lr:.1e} wd {wd} betas {betas} eps {eps}: worst rel param diff {worst:.1e} ({wkey})", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
