#!/usr/bin/env python3
"""Per-shape throughput of the implicit-GEMM conv kernels over the ResNet-18/50 layer shapes
(synthetic data, N images per group x G groups).  Usage: conv_bench.py [depth] [N] [G] [iters] [f32|bf16|split]   (split: the fp32-accurate split-operand kernels; the stem stays fp32)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops
from rot_mvgaze_amd._lib import ConvDesc
from rot_mvgaze_amd.arch import backbone_spec

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 18
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
bf16 = len(sys.argv) > 5 and sys.argv[5] == "bf16"
split = len(sys.argv) > 5 and sys.argv[5] == "split"
adt = torch.bfloat16 if bf16 else torch.float32
dev = torch.device("cuda:0")
spec = backbone_spec(depth)
# walk the net to get input sizes
shapes = {}
H = 224
c = spec.stem
shapes[(8 if bf16 else 4, c.cout, c.k, c.stride, c.pad, H)] = 1
H = (H + 2 * c.pad - c.k) // c.stride + 1
H = (H + 2 - 3) // 2 + 1
for blk in spec.blocks:
    h = H
    for cv in blk.convs:
        key = (cv.cin, cv.cout, cv.k, cv.stride, cv.pad, h)
        shapes[key] = shapes.get(key, 0) + 1
        h = (h + 2 * cv.pad - cv.k) // cv.stride + 1
    if blk.downsample is not None:
        cv = blk.downsample
        key = (cv.cin, cv.cout, cv.k, cv.stride, cv.pad, H)
        shapes[key] = shapes.get(key, 0) + 1
    H = h

def timeit(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

tot = {"fprop": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
print(f"ResNet-{depth}  N={N} G={G}  {'bf16' if bf16 else ('fp32 values, split-operand kernels' if split else 'fp32')}")
print(f"{'cin':>5} {'cout':>5} k s {'hw':>4} cnt | {'fprop ms':>9} {'TF':>6} | {'dgrad ms':>9} {'TF':>6} | {'wgrad ms':>9} {'TF':>6} | dgrad + fused BN reduce (mask bits) ms, GB/s of (dy, dx, y, bits)")
for (cin, cout, k, st, pad, h), cnt in shapes.items():
    d = ConvDesc.make(G, N, h, h, cin, cout, k, st, pad)
    x = torch.randn(G, N, h, h, cin, device=dev).to(adt)
    w32 = torch.randn(cout, k, k, cin, device=dev) * 0.05
    w, wt = ops.cast_weights_bf16(d, w32, cin, True) if bf16 else (w32, w32)
    y = torch.empty(G, N, d.ho, d.wo, cout, device=dev, dtype=adt)
    P, rpp = ops.conv_stats_partials(d, bf16)
    stats = torch.empty(G, P, 2, cout, device=dev)
    gy = torch.randn(y.shape, device=dev).to(adt)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w32)
    flops = 2.0 * G * N * d.ho * d.wo * cout * k * k * cin
    tdf = float("nan")
    rows = N * h * h
    mean, invstd = torch.randn(G, cin, device=dev) * 0.1, torch.rand(G, cin, device=dev) + 0.5
    s12, dgb = torch.empty(2, G, cin, device=dev), torch.zeros(2, cin, device=dev)
    if split and cin == 4:
        # the stem in row-window form (mvg_stem_fprop_split / _wgrad_split); flops counted for the 7x7x3 filter
        xw = ops.stem_rowwindow_split(x)
        w8 = torch.zeros(cout, 7, 8, 4, device=dev)
        w8[:, :, 1:, :3] = w32[..., :3]
        wk8, _ = ops.split_weights(ConvDesc(1, 1, 7, 1, 32, cout, 7, 1, 1, 0, 1, 1), w8.view(cout, 7, 1, 32), False)
        Ps, _ = ops.conv_stats_partials_split(ConvDesc.make(G, N, d.ho, d.wo, 32, cout, 1, 1, 0))
        stats_s = torch.empty(G, Ps, 2, cout, device=dev)
        gys = ops.split_f32(gy)
        dw8 = torch.empty(cout, 7, 8, 4, device=dev)
        flops = flops * 3 / 4
        tf = timeit(lambda: ops.stem_fprop_split(d, xw, wk8, y, stats_s))
        td = float("nan")
        tw = timeit(lambda: ops.stem_wgrad_split(d, xw, gys, dw8))
    elif split and cin > 8:
        xs, gys = ops.split_f32(x), ops.split_f32(gy)
        wk, wts = ops.split_weights(d, w32, True)
        Ps, _ = ops.conv_stats_partials_split(d)
        stats_s = torch.empty(G, Ps, 2, cout, device=dev)
        tf = timeit(lambda: ops.conv_fprop_split(d, xs, wk, y, stats_s))
        td = timeit(lambda: ops.conv_dgrad_split(d, gys, wts, dx))
        tw = timeit(lambda: ops.conv_wgrad_split(d, xs, gys, dw))
        bits = torch.randint(0, 16, (G * rows * cin // 4,), dtype=torch.uint8, device=dev)
        mx = torch.empty(G, cin, device=dev)
        tdf = timeit(lambda: ops.conv_dgrad_split_bnreduce(d, gys, wts, dx, None, x, bits, mean, invstd, None, s12[0], s12[1], dgb[0], dgb[1], False, mx))
        fbytes = 4.0 * (gy.numel() + 2 * x.numel()) + bits.numel()
    else:
        tf = timeit(lambda: ops.conv_fprop(d, x, w, y, None, False, stats))
        td = timeit(lambda: ops.conv_dgrad(d, gy, wt, dx)) if cin > 8 else float("nan")
        tw = timeit(lambda: ops.conv_wgrad(d, x, gy, dw))
        if bf16 and cin % 64 == 0 and cout % 64 == 0:
            bits = torch.randint(0, 256, (G * rows * cin // 8,), dtype=torch.uint8, device=dev)
            tdf = timeit(lambda: ops.conv_dgrad_bf16_bnreduce(d, gy, wt, dx, None, x, bits, mean, invstd, None, s12[0], s12[1], dgb[0], dgb[1], False))
            fbytes = 2.0 * (gy.numel() + 2 * x.numel()) + bits.numel()
    fused = f" | {tdf*1e3:9.3f} {fbytes/tdf/1e9:7.0f}" if tdf == tdf else ""
    print(f"{cin:5d} {cout:5d} {k} {st} {h:4d} {cnt:3d} | {tf*1e3:9.3f} {flops/tf/1e12:6.1f} | {td*1e3:9.3f} {flops/td/1e12:6.1f} | {tw*1e3:9.3f} {flops/tw/1e12:6.1f}{fused}")
    if tdf == tdf:
        tot.setdefault("dgrad+bn", [0, 0]); tot["dgrad+bn"][0] += tdf * cnt; tot["dgrad+bn"][1] += flops * cnt
    for name, t in (("fprop", tf), ("dgrad", td), ("wgrad", tw)):
        if t == t:
            tot[name][0] += t * cnt; tot[name][1] += flops * cnt
for name, (t, f) in tot.items():
    print(f"{name}: {t*1e3:.2f} ms total, {f/t/1e12:.1f} TF/s")
tt = sum(v[0] for k, v in tot.items() if k != "dgrad+bn"); ff = sum(v[1] for k, v in tot.items() if k != "dgrad+bn")
print(f"all: {tt*1e3:.2f} ms, {ff/tt/1e12:.1f} TF/s")
