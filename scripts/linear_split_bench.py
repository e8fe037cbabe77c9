#!/usr/bin/env python3
"""Time the fusion block's Linears on the split-operand kernels (mvg_linear_fprop_split / _dgrad_split / _wgrad_split), one
shape per line: linear_split_bench.py rows [in,out ...].  Default: the three Linears of a C3 fusion iteration
(rows = 128 samples x 12 directions; fuser 3584 -> 3584 -> 1536, head 3584 -> 512)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import ops
from rot_mvgaze_amd._lib import ConvDesc

dev = torch.device("cuda:0")
iters = int(os.environ.get("ITERS", "30"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
shapes = [tuple(map(int, a.split(","))) for a in sys.argv[2:]] or [(3584, 3584), (3584, 1536), (3584, 512)]


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


st = torch.zeros(64, device=dev)
slot = [st[i:i + 1] for i in range(64)]
tot = [0.0, 0.0, 0.0, 0.0]
for fin, fout in shapes:
    torch.manual_seed(0)
    x = torch.randn(rows, fin, device=dev)
    w = torch.randn(fout, fin, device=dev) / fin ** 0.5
    b = torch.randn(fout, device=dev)
    g = torch.randn(rows, fout, device=dev)
    ops.absmax_multi([x, b, g], [slot[0], slot[1], slot[2]])
    x_sp = ops.sp_empty(rows, fin, device=dev); x_sp.sinv = slot[3]
    g_sp = ops.sp_empty(rows, fout, device=dev); g_sp.sinv = slot[4]
    h_sp = ops.sp_empty(rows, fout, device=dev); h_sp.sinv = slot[5]
    ops.split_colsum(x, rows, fin, slot[0], x_sp)
    ops.split_colsum(g, rows, fout, slot[2], g_sp)
    wk, wt = ops.split_weights(ConvDesc.make(1, 1, 1, 1, fin, fout, 1, 1, 0), w, True)
    y = torch.empty(rows, fout, device=dev)
    dx = torch.empty(rows, fin, device=dev)
    dw = torch.empty(fout, fin, device=dev)
    ref = x.double() @ w.double().t() + b.double()
    ops.linear_fprop_split(x_sp, wk, b, False, y, rows, fin, fout, out_absmax=slot[6])
    err_f = ((y.double() - ref).norm() / ref.norm()).item()
    ops.linear_dgrad_split(g_sp, wt, dx, rows, fin, fout, out_absmax=slot[7])
    refd = g.double() @ w.double()
    err_d = ((dx.double() - refd).norm() / refd.norm()).item()
    fl = 2.0 * rows * fin * fout
    tf = timeit(lambda: ops.linear_fprop_split(x_sp, wk, b, False, y, rows, fin, fout, out_absmax=slot[6]))
    th = timeit(lambda: ops.linear_fprop_split(x_sp, wk, b, True, h_sp, rows, fin, fout, bias_absmax=slot[1]))
    td = timeit(lambda: ops.linear_dgrad_split(g_sp, wt, dx, rows, fin, fout, out_absmax=slot[7]))
    tw = timeit(lambda: ops.linear_wgrad_split(x_sp, g_sp, dw, rows, fin, fout))
    for i, t in enumerate((tf, th, td, tw)):
        tot[i] += t
    y.fill_(float("nan")); dx.fill_(float("nan"))
    ops.linear_fprop_split(x_sp, wk, b, False, y, rows, fin, fout, out_absmax=slot[6])
    ops.linear_dgrad_split(g_sp, wt, dx, rows, fin, fout, out_absmax=slot[7])
    err_f = max(err_f, ((y.double() - ref).norm() / ref.norm()).item())
    err_d = max(err_d, ((dx.double() - refd).norm() / refd.norm()).item())
    print(f"{rows}x{fin}->{fout}: fprop(f32) {tf*1e6:6.1f} us {fl/tf/1e12:5.1f} TF | fprop(sp relu) {th*1e6:6.1f} us {fl/th/1e12:5.1f} TF | "
          f"dgrad {td*1e6:6.1f} us {fl/td/1e12:5.1f} TF | wgrad {tw*1e6:6.1f} us {fl/tw/1e12:5.1f} TF | err fprop {err_f:.1e} dgrad {err_d:.1e}",
          flush=True)
print("total us: fprop(f32) %.1f fprop(sp) %.1f dgrad %.1f wgrad %.1f" % tuple(t * 1e6 for t in tot))
