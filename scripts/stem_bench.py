#!/usr/bin/env python3
"""Stem tail at the C2 size (V=2, B=64, 112x112x64): fused BN+ReLU+maxpool kernels vs the separate ones."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops

dev = torch.device("cuda:0")
G, N, H, W, C = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 112, 112, 64
ho, wo = 56, 56
rows = N * H * W
y = torch.randn(G, N, H, W, C, device=dev)
mean = torch.zeros(G, C, device=dev); invstd = torch.ones(G, C, device=dev)
scale = torch.ones(G, C, device=dev); shift = torch.zeros(G, C, device=dev)
gamma = torch.ones(C, device=dev)
pooled = torch.empty(G, N, ho, wo, C, device=dev); am = torch.empty(G, N, ho, wo, C, dtype=torch.uint8, device=dev)
a0 = torch.empty_like(y); ga = torch.empty_like(y); dy = torch.empty_like(y)
gp = torch.randn_like(pooled)
s12 = torch.empty(2, G, C, device=dev); dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

def fwd_sep():
    ops.bn_apply(y, scale, shift, None, True, a0, G, rows, C)
    ops.maxpool_fwd(a0, pooled, am, G * N, H, W, C, ho, wo)
def fwd_fused():
    ops.bn_relu_maxpool_fwd(y, scale, shift, pooled, am, G, N, H, W, C, ho, wo)
def bwd_sep():
    ops.maxpool_bwd(gp, am, ga, G * N, H, W, C, ho, wo)
    ops.bn_bwd_reduce(ga, a0, y, mean, invstd, G, rows, C, s12[0], s12[1], dg, db, False)
    ops.bn_bwd_apply(ga, a0, y, mean, invstd, gamma, s12[0], s12[1], G, rows, C, ga, None)
def bwd_red():
    ops.bn_relu_maxpool_bwd_reduce(gp, am, y, mean, invstd, scale, shift, G, N, H, W, C, ho, wo, s12[0], s12[1], dg, db, False)
def bwd_app():
    ops.bn_relu_maxpool_bwd_apply(gp, am, y, mean, invstd, gamma, scale, shift, s12[0], s12[1], G, N, H, W, C, ho, wo, dy)
fwd_sep()
print(f"forward : separate {timeit(fwd_sep):.3f} ms   fused {timeit(fwd_fused):.3f} ms")
print(f"backward: separate {timeit(bwd_sep):.3f} ms   fused reduce {timeit(bwd_red):.3f} + apply {timeit(bwd_app):.3f} ms")
