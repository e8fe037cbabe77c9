#!/usr/bin/env python3
"""How accurate is an fp32 GEMM computed as six bf16 MFMA products of three-piece operands (conv_split.hip's
arithmetic)?  Emulated with the bf16-product Linear kernel (fp32 in / fp32 out) on explicitly split operands and
compared, against float64, with the fp32-MFMA kernel, with three products only, and with torch's fp32 matmul.
Output committed as profiles/r02_split_accuracy.txt."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rot_mvgaze_amd  # noqa
from rot_mvgaze_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)

def split3(t):
    a1 = t.to(torch.bfloat16).float()
    r = t - a1
    a2 = r.to(torch.bfloat16).float()
    r = r - a2
    a3 = r.to(torch.bfloat16).float()
    assert torch.equal(a1 + a2 + a3, t) or (a1 + a2 + a3 - t).abs().max() < 1e-30
    return a1, a2, a3

for rows, fin, fout in [(2048, 256, 256), (2048, 2304, 256), (2048, 4608, 512), (4096, 1024, 256)]:
    x = torch.relu(torch.randn(rows, fin, device=dev))
    w = torch.randn(fout, fin, device=dev) / fin ** 0.5
    ref = x.double() @ w.double().T
    y32 = torch.empty(rows, fout, device=dev)
    ops.linear_fprop(x, w, None, False, y32, rows, fin, fout)
    xs, ws = split3(x), [p.to(torch.bfloat16).contiguous() for p in split3(w)]
    def mm(i, j):
        y = torch.empty(rows, fout, device=dev)
        ops.linear_fprop_mixed(xs[i].contiguous(), ws[j], None, False, y, rows, fin, fout)
        return y
    y11 = mm(0, 0)
    ref11 = xs[0].double() @ ws[0].double().T
    small = mm(0, 2) + mm(2, 0) + mm(1, 1)
    mid = mm(0, 1) + mm(1, 0)
    y3 = y11 + mid
    y6 = y11 + (mid + small)
    def err(y, r):
        return ((y.double() - r).norm() / r.norm()).item(), ((y.double() - r).abs().max() / r.abs().max()).item()
    print(f"rows {rows} fin {fin} fout {fout}")
    print("   fp32 MFMA        rel L2 %.3e  max %.3e" % err(y32, ref))
    print("   a1*b1 vs exact   rel L2 %.3e  max %.3e   (accumulation error of the bf16 MFMA)" % err(y11, ref11))
    print("   3 products       rel L2 %.3e  max %.3e" % err(y3, ref))
    print("   6 products       rel L2 %.3e  max %.3e" % err(y6, ref))
    print("   torch fp32 mm    rel L2 %.3e  max %.3e" % err(x @ w.T, ref))
