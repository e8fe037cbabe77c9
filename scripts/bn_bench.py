#!/usr/bin/env python3
"""Achieved HBM rate of the BatchNorm streaming passes on the ResNet-50 tensor shapes of one workload
(synthetic data).  Usage: bn_bench.py [N per view] [views] [f32|split]   (split: the passes that write sp)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
split = len(sys.argv) > 3 and sys.argv[3] == "split"
dev = torch.device("cuda:0")

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

print(f"N={N} G={G} {'split (sp outputs)' if split else 'fp32'}: pass, ms, GB/s (algorithmic bytes)")
tot = {}
for hw, C, cnt_plain, cnt_res in [(56, 64, 7, 0), (56, 256, 0, 3), (28, 128, 8, 0), (28, 512, 0, 4), (14, 256, 12, 0), (14, 1024, 0, 6),
                                  (7, 512, 6, 0), (7, 2048, 0, 3)]:
    rows = N * hw * hw
    n = G * rows * C
    y = torch.randn(G, rows, C, device=dev)
    g = torch.randn(G, rows, C, device=dev)
    scale, shift = torch.rand(G, C, device=dev) + 0.5, torch.randn(G, C, device=dev) * 0.3
    mean, invstd = torch.randn(G, C, device=dev) * 0.1, torch.rand(G, C, device=dev) + 0.5
    gamma = torch.rand(C, device=dev) + 0.5
    s1, s2 = torch.randn(G, C, device=dev), torch.randn(G, C, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    res = cnt_res > 0
    cnt = cnt_plain + cnt_res
    if split:
        out = ops.sp_empty(G, rows, C, device=dev)
        r = ops.split_f32(torch.randn(G, rows, C, device=dev)) if res else None
        t_apply = timeit(lambda: ops.bn_apply_split(y, scale, shift, r, True, out, G, rows, C, None, want_bits=res))
        b_apply = n * (4 + 4 + (4.25 if res else 0))
        dy = ops.sp_empty(G, rows, C, device=dev)
        mxc = g.abs().amax(dim=1).contiguous()
        t_bapply = timeit(lambda: ops.bn_bwd_apply_split(g, y, mean, invstd, gamma, s1, s2, G, rows, C, dy, None if res else (scale, shift), mxc))
        b_bapply = n * 12
    else:
        out = torch.empty_like(y)
        r = torch.randn(G, rows, C, device=dev) if res else None
        t_apply = timeit(lambda: (ops.bn_apply_bits(y, scale, shift, r, out, G, rows, C) if res else ops.bn_apply(y, scale, shift, None, True, out, G, rows, C)))
        b_apply = n * (8 + (4.25 if res else 0))
        dy = torch.empty_like(g)
        t_bapply = timeit(lambda: ops.bn_bwd_apply(g, None, y, mean, invstd, gamma, s1, s2, G, rows, C, dy, None, None if res else (scale, shift)))
        b_bapply = n * 12
    if res:
        bits = torch.randint(0, 255, (n // 4,), dtype=torch.uint8, device=dev)
        t_red = timeit(lambda: ops.bn_bwd_reduce_bits(g, bits, y, mean, invstd, G, rows, C, s1, s2, dg, db, False, dz_out=g))
        b_red = n * 12.25
    else:
        t_red = timeit(lambda: ops.bn_bwd_reduce(g, None, y, mean, invstd, G, rows, C, s1, s2, dg, db, False, (scale, shift)))
        b_red = n * 8
    print(f"hw{hw:3d} C{C:5d} x{cnt:2d} {'res' if res else '   '} | apply {t_apply*1e3:6.3f} ms {b_apply/t_apply/1e9:6.0f} | bwd_reduce {t_red*1e3:6.3f} ms {b_red/t_red/1e9:6.0f} | bwd_apply {t_bapply*1e3:6.3f} ms {b_bapply/t_bapply/1e9:6.0f}")
    for k, t, b in (("apply", t_apply, b_apply), ("bwd_reduce", t_red, b_red), ("bwd_apply", t_bapply, b_bapply)):
        a = tot.setdefault(k, [0.0, 0.0]); a[0] += t * cnt; a[1] += b * cnt
for k, (t, b) in tot.items():
    print(f"{k}: {t*1e3:.2f} ms per step, {b/t/1e9:.0f} GB/s")
print(f"all: {sum(v[0] for v in tot.values())*1e3:.2f} ms")
