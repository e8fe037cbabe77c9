#!/usr/bin/env python3
"""How accurate is an fp32 GEMM computed as THREE fp16 MFMA products of two-piece operands (conv_split.hip's arithmetic:
a = a1 + a2, a1 = fp16(a), a2 = fp16(a - a1); a b ~ a1 b1 + a1 b2 + a2 b1)?  The split kernels themselves (a Linear =
a 1x1 conv on a 1x1 map) against float64, next to the fp32-MFMA kernel, torch's fp32 matmul, and - emulated in fp64 -
what each ingredient contributes: the operand rounding (at most the last bit), the dropped a2 b2 term, and the fp32
accumulation.  Output committed as profiles/r03_split_accuracy.txt."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rot_mvgaze_amd  # noqa
from rot_mvgaze_amd import ops
from rot_mvgaze_amd._lib import ConvDesc

dev = torch.device("cuda:0")
torch.manual_seed(0)


def pieces(t):
    a1 = t.to(torch.float16).double()
    a2 = (t.double() - a1).float().to(torch.float16).double()
    return a1, a2


def err(y, r):
    return ((y.double() - r).norm() / r.norm()).item(), ((y.double() - r).abs().max() / r.abs().max()).item()


for rows, fin, fout in [(2048, 256, 256), (2048, 2304, 256), (2048, 4608, 512), (4096, 1024, 256)]:
    x = torch.relu(torch.randn(rows, fin, device=dev))
    w = torch.randn(fout, fin, device=dev) / fin ** 0.5
    ref = x.double() @ w.double().T
    y32 = torch.empty(rows, fout, device=dev)
    ops.linear_fprop(x, w, None, False, y32, rows, fin, fout)
    d = ConvDesc.make(1, rows, 1, 1, fin, fout, 1, 1, 0)
    wk, _ = ops.split_weights(d, w.view(fout, 1, 1, fin).contiguous(), False)
    ysp = torch.empty(1, rows, 1, 1, fout, device=dev)
    ops.conv_fprop_split(d, ops.split_f32(x.view(1, rows, 1, 1, fin)), wk, ysp, None)
    ysp = ysp.view(rows, fout)
    # fp64 emulation of the ingredients (the weights through the same power-of-two scale as the kernel's copy)
    sw = 1.0 / float(wk.sinv)
    (a1, a2), (b1, b2) = pieces(x), pieces(w * sw)
    exact2 = ((a1 + a2) @ (b1 + b2).T) / sw          # operands rounded to two pieces, everything else exact
    three = (a1 @ b1.T + a1 @ b2.T + a2 @ b1.T) / sw    # ... and the a2 b2 term dropped
    print(f"rows {rows} fin {fin} fout {fout}")
    print("   split kernel (3 fp16 MFMAs)   rel L2 %.3e  max %.3e" % err(ysp, ref))
    print("   fp32-MFMA kernel              rel L2 %.3e  max %.3e" % err(y32, ref))
    print("   torch fp32 mm                 rel L2 %.3e  max %.3e" % err(x @ w.T, ref))
    print("   fp64: two-piece operands      rel L2 %.3e  max %.3e   (operand rounding alone)" % err(exact2, ref))
    print("   fp64: ... minus a2 b2         rel L2 %.3e  max %.3e   (operand rounding + dropped term, exact accumulation)" % err(three, ref))
