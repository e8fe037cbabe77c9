#!/usr/bin/env python3
"""BatchNorm kernel timings at the ResNet-18 C2 activation sizes (V=2, B=64)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops
from rot_mvgaze_amd._lib import ConvDesc

dev = torch.device("cuda:0")
G, N = 2, 64

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

print(f"{'hw':>4} {'c':>4} | {'finalize':>9} {'apply':>9} {'apply+res':>9} {'bwd_red':>9} {'bwd_red(y)':>10} {'bwd_app':>9} {'bwd_app(y)':>10}   (us; GB/s in brackets)")
for hw, c in ((56, 64), (28, 128), (14, 256), (7, 512)):
    rows = N * hw * hw
    d = ConvDesc.make(G, N, hw, hw, c, c, 3, 1, 1)
    P, rpp = ops.conv_stats_partials(d)
    stats = torch.rand(G, P, 2, c, device=dev)
    y = torch.randn(G, rows, c, device=dev); out = torch.empty_like(y); res = torch.randn_like(y); g = torch.randn_like(y)
    dy = torch.empty_like(y)
    gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)
    rm = torch.zeros(c, device=dev); rv = torch.ones(c, device=dev)
    aff = torch.empty(4, G, c, device=dev)
    s12 = torch.empty(2, G, c, device=dev); dg = torch.empty(c, device=dev); db = torch.empty(c, device=dev)
    t_fin = timeit(lambda: ops.bn_finalize(stats, G, P, rpp, rows, c, gamma, beta, rm, rv, 0.1, 1e-5, aff[0], aff[1], aff[2], aff[3]))
    aff[0].zero_(); aff[1].fill_(1.0); aff[2].fill_(1.0); aff[3].zero_()
    t_app = timeit(lambda: ops.bn_apply(y, aff[2], aff[3], None, True, out, G, rows, c))
    t_appr = timeit(lambda: ops.bn_apply(y, aff[2], aff[3], res, True, out, G, rows, c))
    t_red = timeit(lambda: ops.bn_bwd_reduce(g, out, y, aff[0], aff[1], G, rows, c, s12[0], s12[1], dg, db, False))
    t_redy = timeit(lambda: ops.bn_bwd_reduce(g, None, y, aff[0], aff[1], G, rows, c, s12[0], s12[1], dg, db, False, (aff[2], aff[3])))
    t_ba = timeit(lambda: ops.bn_bwd_apply(g, out, y, aff[0], aff[1], gamma, s12[0], s12[1], G, rows, c, dy, None))
    t_bay = timeit(lambda: ops.bn_bwd_apply(g, None, y, aff[0], aff[1], gamma, s12[0], s12[1], G, rows, c, dy, None, (aff[2], aff[3])))
    nb = y.numel() * 4
    gb = lambda k, t: f"{t:7.1f}[{k * nb / t / 1e3:4.0f}]"
    print(f"{hw:4d} {c:4d} | {t_fin:9.1f} {gb(2, t_app)} {gb(3, t_appr)} {gb(3, t_red)} {gb(2, t_redy)} {gb(4, t_ba)} {gb(3, t_bay)}", flush=True)
