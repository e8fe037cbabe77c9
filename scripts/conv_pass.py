#!/usr/bin/env python3
"""Run every convolution op of ONE training step of a workload exactly once (fprop with BN partial
statistics, dgrad, wgrad per layer, in network order) - the launch set bench.py's `roofline`
averages over - so that `rocprofv3 --pmc` totals divide into per-op HBM traffic.
Usage: conv_pass.py [depth] [N per view] [views] [f32|split]"""
import os, sys
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
split = len(sys.argv) > 4 and sys.argv[4] == "split"
dev = torch.device("cuda:0")
spec = backbone_spec(depth)
layers = []
H = 224
c = spec.stem
layers.append((4, c.cout, c.k, c.stride, c.pad, H, False))
H = (H + 2 * c.pad - c.k) // c.stride + 1
H = (H + 2 - 3) // 2 + 1
for blk in spec.blocks:
    h = H
    for cv in blk.convs:
        layers.append((cv.cin, cv.cout, cv.k, cv.stride, cv.pad, h, True))
        h = (h + 2 * cv.pad - cv.k) // cv.stride + 1
    if blk.downsample is not None:
        cv = blk.downsample
        layers.append((cv.cin, cv.cout, cv.k, cv.stride, cv.pad, H, True))
    H = h
n_ops = 0
flops = 0.0
op_log = []          # (op, shape, algorithmic read bytes, algorithmic write bytes) in launch order: scripts/pmc_per_op.py joins it
                     # with the per-dispatch counters of a rocprofv3 --pmc pass over this script
for (cin, cout, k, st, pad, h, need_dx) in layers:
    d = ConvDesc.make(G, N, h, h, cin, cout, k, st, pad)
    x = torch.randn(G, N, h, h, cin, device=dev)
    w = torch.randn(cout, k, k, cin, device=dev) * 0.05
    y = torch.empty(G, N, d.ho, d.wo, cout, device=dev)
    P, rpp = ops.conv_stats_partials(d)
    stats = torch.empty(G, P, 2, cout, device=dev)
    gy = torch.randn_like(y)
    dw = torch.empty_like(w)
    if split and cin == 4:
        # the stem in row-window form (the window operand is built by the input transform, not a conv op)
        xw = ops.stem_rowwindow_split(x)
        w8 = torch.zeros(cout, 7, 8, 4, device=dev)
        w8[:, :, 1:, :3] = w[..., :3]
        wk, _ = ops.split_weights(ConvDesc(1, 1, 7, 1, 32, cout, 7, 1, 1, 0, 1, 1), w8.view(cout, 7, 1, 32), False)
        Ps, _ = ops.conv_stats_partials_split(ConvDesc.make(G, N, d.ho, d.wo, 32, cout, 1, 1, 0))
        stats_s = torch.empty(G, Ps, 2, cout, device=dev)
        gys = ops.split_f32(gy)
        torch.cuda.synchronize()
        ops.stem_fprop_split(d, xw, wk, y, stats_s)
        ops.stem_wgrad_split(d, xw, gys, torch.empty(cout, 7, 8, 4, device=dev))
        n_ops += 2
    elif split:
        xs, gys = ops.split_f32(x), ops.split_f32(gy)
        wk, wts = ops.split_weights(d, w, True)
        Ps, _ = ops.conv_stats_partials_split(d)
        stats_s = torch.empty(G, Ps, 2, cout, device=dev)
        ops.conv_fprop_split(d, xs, wk, y, stats_s)
        ops.conv_wgrad_split(d, xs, gys, dw)
        n_ops += 2
        shp = f"{cin}->{cout} k{k} s{st} hw{h}"
        op_log.append(("fprop", shp, 4.0 * (x.numel() + w.numel()), 4.0 * y.numel() + 4.0 * stats_s.numel()))
        op_log.append(("wgrad", shp, 4.0 * (x.numel() + gy.numel()), 4.0 * w.numel()))
        if need_dx:
            # as in the step: backward-data carries the BatchNorm-backward reduce of the unit it feeds (mask bits, in-place addend)
            dx = torch.randn_like(x)
            rows = N * h * h
            bits = torch.randint(0, 16, (G * rows * cin // 4,), dtype=torch.uint8, device=dev)
            mean, invstd = torch.zeros(G, cin, device=dev), torch.ones(G, cin, device=dev)
            s12, dgb = torch.empty(3, G, cin, device=dev), torch.zeros(2, cin, device=dev)
            ops.conv_dgrad_split_bnreduce(d, gys, wts, dx, dx, x, bits, mean, invstd, None, s12[0], s12[1], dgb[0], dgb[1], False, s12[2])
            n_ops += 1
            op_log.append(("dgrad+bn", shp, 4.0 * (gy.numel() + w.numel() + 2 * x.numel()) + bits.numel(), 4.0 * x.numel()))
    else:
        ops.conv_fprop(d, x, w, y, None, False, stats)
        ops.conv_wgrad(d, x, gy, dw)
        n_ops += 2
        if need_dx:
            dx = torch.randn_like(x)
            ops.conv_dgrad(d, gy, w, dx, None, dx)       # in-place addend, like the residual branches
            n_ops += 1
    flops += 2.0 * G * N * d.ho * d.wo * cout * k * k * cin * (3 if need_dx else 2)
    torch.cuda.synchronize()
    del x, y, gy, stats
print(f"conv ops: {n_ops}  flops: {flops:.4e}")
if os.environ.get("CONV_PASS_LOG"):
    import json
    json.dump(op_log, open(os.environ["CONV_PASS_LOG"], "w"))
