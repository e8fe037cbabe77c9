"""Per-kernel parity: every C-ABI entry point against the same op of the CPU oracle vocabulary
(torch-CPU functional ops, fp32/fp64) on seeded inputs.  Runs on a real MI355X only."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import rot_mvgaze_amd  # noqa: F401
from rot_mvgaze_amd import synth

pytestmark = pytest.mark.gpu

RTOL = 2e-5      # fp32 MFMA (exact products, fp32 accumulate) vs oneDNN fp32: reduction-order noise


def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, tag="t", scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((synth.normal(n, seed, tag) * scale).astype(np.float32).reshape(shape))


def close(got, ref, rtol=RTOL, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-30
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e})"


def to_nhwc(x):     # [G,N,C,H,W] -> [G,N,H,W,C]
    return x.permute(0, 1, 3, 4, 2).contiguous()


CONV_CASES = [
    # G, N, H, W, Cin, Cout, k, stride, pad
    (2, 3, 14, 14, 64, 128, 3, 1, 1),
    (2, 3, 15, 13, 64, 128, 3, 2, 1),
    (1, 5, 14, 14, 128, 64, 1, 1, 0),
    (2, 2, 14, 14, 64, 256, 1, 2, 0),
    (2, 2, 36, 36, 4, 64, 7, 2, 3),       # stem (3 channels padded to 4)
    (2, 8, 56, 56, 64, 256, 1, 1, 0),     # large M -> 128x128 tile
    (1, 16, 28, 28, 128, 128, 3, 1, 1),   # -> 128x64 / 128x128
    (2, 2, 7, 7, 512, 512, 3, 1, 1),
    (1, 3, 9, 9, 32, 32, 3, 1, 1),        # 32-wide -> 128x32 tile
    (2, 4, 16, 16, 64, 128, 3, 2, 1),     # stride 2, even size (parity-class dgrad)
    (1, 2, 28, 28, 128, 256, 1, 2, 0),    # 1x1 stride 2: three of four parity classes have no tap
    (1, 2, 17, 18, 64, 64, 7, 2, 3),      # 7x7 stride 2 dgrad (4/3-tap lattices per axis)
    (2, 8, 56, 56, 256, 512, 1, 2, 0),
    (2, 64, 56, 56, 64, 128, 1, 2, 0),    # many more wgrad splits than pixels/272: empty trailing splits
    (2, 16, 28, 28, 128, 128, 3, 1, 1),   # 196 tiles of 128x128 on 512 slots -> stream-K (fprop and dgrad)
    (2, 16, 56, 56, 64, 64, 3, 1, 1),     # 784 tiles of 128x64 -> stream-K with several tiles per workgroup
    (2, 16, 56, 56, 64, 128, 3, 2, 1),    # stride 2: stream-K fprop, parity-class dgrad
    (1, 30, 14, 14, 256, 256, 3, 1, 1),   # ragged last M tile (5880 rows) under stream-K
    (1, 24, 57, 57, 64, 64, 1, 1, 0),     # 1219 BN partials (ragged last one): two-level bn_finalize
    (1, 4, 20, 20, 16, 64, 3, 1, 1),      # 16 channels per tap: uniform-tap loader with tap-major K order
    (2, 6, 30, 30, 16, 128, 3, 2, 1),     # same, stride 2 (dgrad classes over 128 channels)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fprop_dgrad_wgrad(case):
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    G, N, H, W, Cin, Cout, k, st, pad = case
    x = rnd((G, N, Cin, H, W), 1, "x")
    if Cin == 4:
        x[:, :, 3] = 0
    w = rnd((Cout, Cin, k, k), 2, "w", 1.0 / np.sqrt(Cin * k * k))
    d = ConvDesc.make(G, N, H, W, Cin, Cout, k, st, pad)
    xr = x.reshape(G * N, Cin, H, W).double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, st, pad)
    gy = rnd(tuple(yr.shape), 3, "gy")
    yr.backward(gy.double())

    xd = to_nhwc(x).to(dev())
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev())        # KRSC
    y = torch.empty(G, N, d.ho, d.wo, Cout, device=dev())
    P, rpp = ops.conv_stats_partials(d)
    stats = torch.full((G, P, 2, Cout), float("nan"), device=dev())
    ops.conv_fprop(d, xd, wd, y, None, False, stats)
    y_ref = yr.detach().float().reshape(G, N, Cout, d.ho, d.wo).permute(0, 1, 3, 4, 2)
    close(y, y_ref, what="fprop")

    # statistics -> finalize
    rows = N * d.ho * d.wo
    gamma, beta = rnd((Cout,), 4, "g") * 0.1 + 1, rnd((Cout,), 5, "b") * 0.1
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    mean, invstd, scale, shift = (torch.empty(G, Cout, device=dev()) for _ in range(4))
    rmd, rvd = rm.to(dev()), rv.to(dev())
    ops.bn_finalize(stats, G, P, rpp, rows, Cout, gamma.to(dev()), beta.to(dev()), rmd, rvd, 0.1, 1e-5, mean, invstd,
                    scale, shift)
    yg = y_ref.double().reshape(G, rows, Cout)
    m_ref, v_ref = yg.mean(1), yg.var(1, unbiased=False)
    close(mean, m_ref, 1e-5, "bn mean")
    close(invstd, 1.0 / torch.sqrt(v_ref + 1e-5), 1e-5, "bn invstd")
    for g in range(G):      # running stats: group order
        rm = 0.9 * rm + 0.1 * m_ref[g].float()
        rv = 0.9 * rv + 0.1 * (yg[g].var(0, unbiased=True)).float()
    close(rmd, rm, 1e-5, "running_mean")
    close(rvd, rv, 1e-5, "running_var")

    gyd = to_nhwc(gy.reshape(G, N, Cout, d.ho, d.wo)).to(dev())
    dx = torch.empty(G, N, H, W, Cin, device=dev())
    ops.conv_dgrad(d, gyd, wd, dx)
    dx_ref = xr.grad.float().reshape(G, N, Cin, H, W).permute(0, 1, 3, 4, 2)
    close(dx, dx_ref, what="dgrad")
    # dgrad epilogue: mask + addend (aliasing dx)
    mask = rnd((G, N, H, W, Cin), 6, "m").to(dev())
    add = rnd((G, N, H, W, Cin), 7, "a").to(dev())
    dx2 = add.clone()
    ops.conv_dgrad(d, gyd, wd, dx2, mask, dx2)
    close(dx2, dx_ref * (mask.cpu() > 0) + add.cpu(), what="dgrad epilogue")

    dw = torch.empty(Cout, k, k, Cin, device=dev())
    ops.conv_wgrad(d, xd, gyd, dw, False)
    dw_ref = wr.grad.float().permute(0, 2, 3, 1)
    close(dw, dw_ref, what="wgrad")
    ops.conv_wgrad(d, xd, gyd, dw, True)
    close(dw, 2 * dw_ref, what="wgrad accumulate")


@pytest.mark.parametrize("rows,fin,fout,relu", [(128, 2048, 2048, True), (6, 512, 1536, False), (70, 3584, 512, True),
                                                (33, 1536, 1536, False), (1536, 3584, 3584, True), (3000, 2048, 1536, False)])
def test_linear_via_conv(rows, fin, fout, relu):
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    x, w, b = rnd((rows, fin), 1), rnd((fout, fin), 2, "w", fin ** -0.5), rnd((fout,), 3, "b")
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    pre = F.linear(xr, wr, br)
    gy = rnd((rows, fout), 4)
    d = ConvDesc.linear(rows, fin, fout)
    xd, wd, bd = x.to(dev()), w.to(dev()), b.to(dev())
    y = torch.empty(rows, fout, device=dev())
    ops.conv_fprop(d, xd, wd, y, bd, relu, None)
    yr = F.relu(pre) if relu else pre
    close(y, yr, what="linear fwd")
    # backward reference with the kernel's own ReLU pattern: among millions of outputs a few sit within fp32
    # rounding of 0 and fall on the other side of the ReLU than in fp64 (each flips a whole row of dx by ~1e-2)
    (pre * (y > 0).cpu().double() if relu else pre).backward(gy.double())
    g = gy.to(dev())
    if relu:
        g = g * (y > 0)
    dx = torch.empty(rows, fin, device=dev())
    ops.conv_dgrad(d, g, wd, dx)
    close(dx, xr.grad, what="linear dgrad")
    dw = torch.empty(fout, fin, device=dev())
    ops.conv_wgrad(d, xd, g, dw)
    close(dw, wr.grad, what="linear wgrad")
    # weight and bias gradient in one launch (the bias gradient rides on the kernel that streams g)
    dw2, db = torch.full((fout, fin), float("nan"), device=dev()), torch.full((fout,), float("nan"), device=dev())
    ops.linear_wgrad(xd, g, dw2, db, rows, fin, fout, False)
    close(dw2, wr.grad, what="linear_wgrad dw")
    close(db, br.grad, what="linear_wgrad db")
    ops.linear_wgrad(xd, g, dw2, db, rows, fin, fout, True)
    close(dw2, 2 * wr.grad, what="linear_wgrad dw accumulate")
    close(db, 2 * br.grad, what="linear_wgrad db accumulate")
    # split-K entry points (what the fusion block calls), incl. the fused mask / addend epilogue
    y2 = torch.empty(rows, fout, device=dev())
    ops.linear_fprop(xd, wd, bd, relu, y2, rows, fin, fout)
    close(y2, yr, what="linear_fprop (split-K)")
    mask, add = rnd((rows, fin), 8).to(dev()), rnd((rows, fin), 9).to(dev())
    dx2 = add.clone()
    ops.linear_dgrad(g, wd, mask, dx2, dx2, rows, fin, fout)
    close(dx2, xr.grad * (mask.cpu() > 0) + add.cpu(), what="linear_dgrad (split-K, mask+addend)")


@pytest.mark.parametrize("V,B,cf,fout,use_rel", [(2, 64, 512, 2048, True), (4, 32, 2048, 3584, True), (4, 128, 2048, 512, False),
                                                 (8, 5, 512, 2048, True), (2, 3, 2048, 3584, True)])
def test_fuser_gemm_generates_rotate_concat_in_the_loader(V, B, cf, fout, use_rel):
    """mvg_fuser_fprop / mvg_fuser_wgrad (X = [img_feat | R @ F] built inside the GEMM's operand loader, never
    written) == rotcat_fwd + linear_fprop / linear_wgrad on the materialised X; rows up to C3's 1536."""
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd.heads import directed_pairs
    nvec = 512
    vi, vj = directed_pairs(V)
    D = len(vi)
    rows, kin = D * B, cf + 3 * nvec
    img = rnd((V, B, cf), 1, "i").to(dev())
    feat = rnd((V, B, 3, nvec), 2, "f").to(dev())
    rel = rnd((D, B, 3, 3), 3, "r").to(dev()) if use_rel else None
    w = rnd((fout, kin), 4, "w", kin ** -0.5).to(dev())
    bias = rnd((fout,), 5, "b").to(dev())
    vi_t, vj_t = torch.tensor(vi, dtype=torch.int32, device=dev()), torch.tensor(vj, dtype=torch.int32, device=dev())
    X = torch.empty(rows, kin, device=dev())
    ops.rotcat_fwd(img, feat, rel, vi_t, vj_t, X, B, D, cf, nvec)
    y0 = torch.empty(rows, fout, device=dev())
    ops.linear_fprop(X, w, bias, True, y0, rows, kin, fout)
    b_idx = torch.arange(B, dtype=torch.int32)
    row_img = (torch.tensor(vi, dtype=torch.int32)[:, None] * B + b_idx[None]).reshape(-1).to(dev())
    row_src = (torch.tensor(vj, dtype=torch.int32)[:, None] * B + b_idx[None]).reshape(-1).to(dev())
    y1 = torch.full((rows, fout), float("nan"), device=dev())
    ops.fuser_fprop(img.reshape(V * B, cf), feat.reshape(V * B, 3 * nvec), rel, row_img, row_src, w, bias, True, y1, rows, cf, nvec,
                    fout)
    close(y1, y0, 1e-5, "fuser_fprop vs rotcat + linear_fprop")
    g = rnd((rows, fout), 6, "g").to(dev())
    dw0, db0 = torch.empty(fout, kin, device=dev()), torch.empty(fout, device=dev())
    dw1, db1 = torch.full((fout, kin), float("nan"), device=dev()), torch.full((fout,), float("nan"), device=dev())
    ops.linear_wgrad(X, g, dw0, db0, rows, kin, fout, False)
    ops.fuser_wgrad(img.reshape(V * B, cf), feat.reshape(V * B, 3 * nvec), rel, row_img, row_src, g, dw1, db1, rows, cf, nvec, fout,
                    False)
    close(dw1, dw0, 1e-5, "fuser_wgrad dw")
    close(db1, db0, 1e-6, "fuser_wgrad db")


def test_linear_skinny():
    from rot_mvgaze_amd import ops
    rows, k, nout = 37, 512, 2
    x, w, b = F.relu(rnd((rows, k), 1)), rnd((nout, k), 2, "w", k ** -0.5), rnd((nout,), 3, "b")
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    gy = rnd((rows, nout), 4)
    yr.backward(gy.double())
    xd, wd, bd, g = x.to(dev()), w.to(dev()), b.to(dev()), gy.to(dev())
    y = torch.empty(rows, nout, device=dev())
    ops.linear_skinny_fwd(xd, wd, bd, y, rows, k, nout)
    close(y, yr, what="skinny fwd")
    dx, dw, db = torch.empty(rows, k, device=dev()), torch.empty(nout, k, device=dev()), torch.empty(nout, device=dev())
    ops.linear_skinny_bwd(g, xd, wd, xd, dx, dw, db, rows, k, nout)
    close(dx, xr.grad * (x > 0), what="skinny dx")
    close(dw, wr.grad, what="skinny dw")
    close(db, br.grad, what="skinny db")


@pytest.mark.parametrize("G,N,H,W,C,relu,res", [(2, 3, 9, 9, 64, True, True), (2, 2, 7, 7, 2048, True, False),
                                                (1, 4, 12, 12, 256, False, False), (3, 2, 28, 28, 128, True, True),
                                                (2, 3, 16, 16, 64, True, True), (2, 2, 7, 7, 512, True, True),
                                                (2, 2, 56, 56, 64, True, False)])
def test_bn_apply_and_backward(G, N, H, W, C, relu, res):
    from rot_mvgaze_amd import ops
    rows = N * H * W
    y = rnd((G, rows, C), 1) * 2 + 0.5
    r = rnd((G, rows, C), 2, "r") if res else None
    gamma, beta = rnd((C,), 3) * 0.2 + 1, rnd((C,), 4) * 0.2
    yr = y.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    pre = []
    for g in range(G):          # per-group statistics (one backbone call per view in the reference)
        o = F.batch_norm(yr[g].reshape(N, H, W, C).permute(0, 3, 1, 2), None, None, gr, br, True, 0.1, 1e-5)
        o = o.permute(0, 2, 3, 1).reshape(rows, C)
        if res:
            o = o + rr[g]
        pre.append(o)
    pre = torch.stack(pre)

    yd = y.to(dev())
    yg = y.double()
    mean = yg.mean(1).float().to(dev())
    invstd = (1.0 / torch.sqrt(yg.var(1, unbiased=False) + 1e-5)).float().to(dev())
    gd, bd = gamma.to(dev()), beta.to(dev())
    scale = gd[None] * invstd
    shift = bd[None] - mean * scale
    out = torch.empty(G, rows, C, device=dev())
    ops.bn_apply(yd, scale, shift, r.to(dev()) if res else None, relu, out, G, rows, C)
    close(out, F.relu(pre) if relu else pre, what="bn_apply")
    # backward reference with the kernel's own ReLU pattern (an element within fp32 rounding of 0
    # may legitimately fall on the other side of the ReLU than in the fp64 reference)
    out_ref = pre * (out > 0).cpu().double() if relu else pre
    go = rnd((G, rows, C), 5, "go")
    out_ref.backward(go.double())

    s1, s2 = torch.empty(G, C, device=dev()), torch.empty(G, C, device=dev())
    dgamma, dbeta = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    god = go.to(dev())
    act = out if relu else None
    ops.bn_bwd_reduce(god, act, yd, mean, invstd, G, rows, C, s1, s2, dgamma, dbeta, False)
    close(dgamma, gr.grad, 1e-4, "dgamma")
    close(dbeta, br.grad, 1e-4, "dbeta")
    dy = torch.empty(G, rows, C, device=dev())
    dz = torch.empty(G, rows, C, device=dev()) if res else None
    ops.bn_bwd_apply(god, act, yd, mean, invstd, gd, s1, s2, G, rows, C, dy, dz)
    close(dy, yr.grad, 1e-4, "bn dy")
    if res:
        close(dz, rr.grad, what="residual grad")
    if relu and not res:
        # ReLU mask rebuilt from y (relu_affine): bit-identical to the mask taken from `out`
        s1b, s2b, dgb, dbb = (torch.empty_like(t) for t in (s1, s2, dgamma, dbeta))
        ops.bn_bwd_reduce(god, None, yd, mean, invstd, G, rows, C, s1b, s2b, dgb, dbb, False, (scale, shift))
        assert torch.equal(s1b, s1) and torch.equal(s2b, s2) and torch.equal(dgb, dgamma) and torch.equal(dbb, dbeta)
        dyb = torch.empty_like(dy)
        ops.bn_bwd_apply(god, None, yd, mean, invstd, gd, s1, s2, G, rows, C, dyb, None, (scale, shift))
        assert torch.equal(dyb, dy)
    # the aliased forms the backbone uses: dy in place of g; dz in place of g
    g2 = god.clone()
    ops.bn_bwd_apply(g2, act, yd, mean, invstd, gd, s1, s2, G, rows, C, g2, None)
    close(g2, yr.grad, 1e-4, "bn dy (in place)")
    if res:
        g3, dy3 = god.clone(), torch.empty(G, rows, C, device=dev())
        ops.bn_bwd_apply(g3, act, yd, mean, invstd, gd, s1, s2, G, rows, C, dy3, g3)
        close(dy3, yr.grad, 1e-4, "bn dy (dz in place)")
        close(g3, rr.grad, what="residual grad (in place)")
    if res and relu:
        # what the backbone runs for a block's last unit: the reduce pass writes the masked gradient over g,
        # the apply pass reads (dz, y) only; same sums and gradients bit for bit
        g4 = god.clone()
        s1c, s2c, dgc, dbc = (torch.empty_like(t) for t in (s1, s2, dgamma, dbeta))
        ops.bn_bwd_reduce(g4, act, yd, mean, invstd, G, rows, C, s1c, s2c, dgc, dbc, False, None, dz_out=g4)
        assert torch.equal(s1c, s1) and torch.equal(s2c, s2) and torch.equal(dgc, dgamma) and torch.equal(dbc, dbeta)
        assert torch.equal(g4, dz)
        dy4 = torch.empty_like(dy)
        ops.bn_bwd_apply(g4, None, yd, mean, invstd, gd, s1c, s2c, G, rows, C, dy4, None, None)
        assert torch.equal(dy4, dy)
    if res and relu:
        # ... and with the mask carried as bits (one byte per 16-byte access, written by the forward apply pass)
        out5 = torch.empty_like(out)
        bits = ops.bn_apply_bits(yd, scale, shift, r.to(dev()), out5, G, rows, C)
        assert torch.equal(out5, out) and bits.numel() == G * rows * C // 4
        want_bits = ((out > 0).reshape(-1, 4).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=dev())).sum(1)
        assert torch.equal(bits, want_bits.to(torch.uint8))
        g5 = god.clone()
        s1d, s2d, dgd, dbd = (torch.empty_like(t) for t in (s1, s2, dgamma, dbeta))
        ops.bn_bwd_reduce_bits(g5, bits, yd, mean, invstd, G, rows, C, s1d, s2d, dgd, dbd, False, dz_out=g5)
        assert torch.equal(s1d, s1) and torch.equal(s2d, s2) and torch.equal(dgd, dgamma) and torch.equal(dbd, dbeta)
        assert torch.equal(g5, dz)
    if res:
        # residual given as the RAW output of the downsample conv + its BatchNorm's (scale, shift)
        rs = (rnd((G, C), 11, "rs") * 0.2 + 1).to(dev())
        rh = (rnd((G, C), 12, "rh") * 0.2).to(dev())
        raw = r.to(dev())
        normalised = torch.empty_like(raw)
        ops.bn_apply(raw, rs, rh, None, False, normalised, G, rows, C)
        want, got = torch.empty_like(out), torch.empty_like(out)
        ops.bn_apply(yd, scale, shift, normalised, relu, want, G, rows, C)
        ops.bn_apply(yd, scale, shift, raw, relu, got, G, rows, C, (rs, rh))
        assert torch.equal(got, want), "downsample BatchNorm folded into the consumer's bn_apply"


@pytest.mark.parametrize("G,N,H,W", [(2, 3, 14, 18), (1, 2, 15, 13)])
def test_fused_stem_bn_relu_maxpool(G, N, H, W):
    """mvg_bn_relu_maxpool_{fwd,bwd_reduce,bwd_apply} against torch fp64 batch_norm -> relu ->
    max_pool2d (one BatchNorm call per group, like the reference's per-view backbone calls)."""
    from rot_mvgaze_amd import ops
    C = 64
    y = rnd((G, N, C, H, W), 11) * 1.5 + 0.2
    gamma, beta = rnd((C,), 12) * 0.3 + 1.0, rnd((C,), 13) * 0.3
    gamma[::7] *= -1                                           # negative scales flip the window order
    yr = y.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    outs = [F.max_pool2d(F.relu(F.batch_norm(yr[g], None, None, gr, br, True, 0.1, 1e-5)), 3, 2, 1) for g in range(G)]
    pr = torch.stack(outs)
    gp = rnd(tuple(pr.shape), 14)
    pr.backward(gp.double())
    ho, wo = pr.shape[3], pr.shape[4]
    rows = N * H * W
    yd = y.permute(0, 1, 3, 4, 2).contiguous().to(dev())
    yg = y.double().permute(0, 1, 3, 4, 2).reshape(G, rows, C)
    mean = yg.mean(1).float().to(dev())
    invstd = (1.0 / torch.sqrt(yg.var(1, unbiased=False) + 1e-5)).float().to(dev())
    scale = (gamma.to(dev())[None] * invstd).contiguous()
    shift = (beta.to(dev())[None] - mean * scale).contiguous()
    pooled = torch.empty(G, N, ho, wo, C, device=dev())
    am = torch.empty(G, N, ho, wo, C, dtype=torch.uint8, device=dev())
    ops.bn_relu_maxpool_fwd(yd, scale, shift, pooled, am, G, N, H, W, C, ho, wo)
    close(pooled, pr.detach().permute(0, 1, 3, 4, 2), 1e-5, "fused stem forward")
    # the unfused kernels give the same bits (same fma, same scan order)
    a0 = torch.empty_like(yd)
    ops.bn_apply(yd, scale, shift, None, True, a0, G, rows, C)
    p2, am2 = torch.empty_like(pooled), torch.empty_like(am)
    ops.maxpool_fwd(a0, p2, am2, G * N, H, W, C, ho, wo)
    assert torch.equal(p2, pooled)
    gpd = gp.permute(0, 1, 3, 4, 2).contiguous().to(dev())
    s12 = torch.empty(2, G, C, device=dev())
    dg, db = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    ops.bn_relu_maxpool_bwd_reduce(gpd, am, yd, mean, invstd, scale, shift, G, N, H, W, C, ho, wo, s12[0], s12[1], dg, db,
                                   False)
    dy = torch.empty_like(yd)
    ops.bn_relu_maxpool_bwd_apply(gpd, am, yd, mean, invstd, gamma.to(dev()), scale, shift, s12[0], s12[1], G, N, H, W, C,
                                  ho, wo, dy)
    close(dg, gr.grad, 1e-4, "fused stem dgamma")
    close(db, br.grad, 1e-4, "fused stem dbeta")
    close(dy, yr.grad.permute(0, 1, 3, 4, 2), 1e-4, "fused stem dy")


def test_pools_and_layout():
    from rot_mvgaze_amd import ops
    n, h, w, c = 3, 15, 17, 64
    x = rnd((n, c, h, w), 1)
    xr = x.double().requires_grad_(True)
    yr = F.max_pool2d(F.relu(xr), 3, 2, 1)
    gy = rnd(tuple(yr.shape), 2)
    yr.backward(gy.double())
    ho, wo = yr.shape[2], yr.shape[3]
    xd = F.relu(x).permute(0, 2, 3, 1).contiguous().to(dev())
    y = torch.empty(n, ho, wo, c, device=dev())
    am = torch.empty(n, ho, wo, c, dtype=torch.uint8, device=dev())
    ops.maxpool_fwd(xd, y, am, n, h, w, c, ho, wo)
    assert torch.equal(y.cpu(), yr.detach().float().permute(0, 2, 3, 1))
    dx = torch.empty(n, h, w, c, device=dev())
    ops.maxpool_bwd(gy.permute(0, 2, 3, 1).contiguous().to(dev()), am, dx, n, h, w, c, ho, wo)
    # the reference then applies ReLU backward (mask x>0): ties at 0 may route differently but are masked away
    mask = (xd > 0).cpu()
    close(dx.cpu() * mask, xr.grad.float().permute(0, 2, 3, 1) * mask, what="maxpool bwd")

    z = rnd((5, 49, 512), 3)
    zd = z.to(dev())
    p = torch.empty(5, 512, device=dev())
    ops.avgpool_fwd(zd, p, 5, 49, 512)
    close(p, z.double().mean(1), what="avgpool")
    gp = rnd((5, 512), 4).to(dev())
    dz = torch.empty(5, 49, 512, device=dev())
    ops.avgpool_bwd(gp, dz, 5, 49, 512)
    close(dz, (gp.cpu() / 49)[:, None, :].expand(5, 49, 512), what="avgpool bwd")

    img = rnd((4, 3, 20, 24), 5)
    d4 = torch.empty(4, 20, 24, 4, device=dev())
    ops.nchw_to_nhwc4(img.to(dev()), d4, 4, 3, 20, 24)
    assert torch.equal(d4[..., :3].cpu(), img.permute(0, 2, 3, 1)) and float(d4[..., 3].abs().max()) == 0.0
    back = torch.empty(4, 3, 20, 24, device=dev())
    ops.nhwc4_to_nchw(d4, back, 4, 3, 20, 24)
    assert torch.equal(back.cpu(), img)


def test_geometry_and_loss_golden(golden_dir):
    import os
    from rot_mvgaze_amd import ops
    g = np.load(os.path.join(golden_dir, "geometry_loss.npz"))
    hp = torch.from_numpy(g["hp"]).to(dev())
    R = torch.empty(hp.shape[0], 3, 3, device=dev())
    ops.rotation_matrix_2d(hp, R, False)
    np.testing.assert_allclose(R.cpu().numpy(), g["R"], atol=2e-7)
    ops.rotation_matrix_2d(hp, R, True)
    np.testing.assert_allclose(R.cpu().numpy(), g["R_inv"], atol=2e-7)

    pred, gt = torch.from_numpy(g["loss_pred"]).to(dev()), torch.from_numpy(g["loss_gt"]).to(dev())
    n = pred.shape[0]
    loss, dpred = torch.zeros(1, device=dev()), torch.empty(n, 2, device=dev())
    ops.gaze_angular_loss(pred, gt, n, 1.0 / n, loss, False, dpred, None)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    ref = g["loss_dpred"]
    np.testing.assert_allclose(dpred.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    p2 = torch.tensor([[0.1, 0.2], [0.0, 0.0]], device=dev())
    g2 = torch.tensor([[0.1, 0.25], [0.3, -0.2]], device=dev())
    d2 = torch.empty(2, 2, device=dev())
    ops.gaze_angular_loss(p2, g2, 2, 0.5, loss, False, d2, None)
    np.testing.assert_allclose(loss.item(), 11.706101, rtol=1e-5)
    np.testing.assert_allclose(d2.cpu().numpy(), g["ka_dpred"], rtol=1e-4)


@pytest.mark.parametrize("rows,fin,fout", [(584, 1984, 1728), (1239, 2072, 1520), (1536, 2048, 2048), (128, 2048, 2048),
                                            (3584, 2048, 1536)])
def test_linear_split_k_covers_every_slab(rows, fin, fout):
    """Linear GEMMs at the row counts of configs C3-C5 (M = pairs x batch = 1536, 3584) and at ragged sizes:
    the split-K plan must count K-steps in the units of the kernel that runs (the 128x128 fprop tile steps
    by 32; counting in 16s left the trailing slabs unwritten).  The workspace is poisoned with NaN so an
    unwritten slab cannot hide behind stale memory."""
    import ctypes as C
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import lib
    rng = np.random.default_rng(rows)
    x = torch.from_numpy(rng.standard_normal((rows, fin)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((fout, fin)) / np.sqrt(fin)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(fout).astype(np.float32))
    gy = torch.from_numpy(rng.standard_normal((rows, fout)).astype(np.float32))
    xd, wd, bd, gyd = x.to(dev()), w.to(dev()), b.to(dev()), gy.to(dev())
    n = lib().mvg_linear_workspace_floats(rows, fin, fout)
    ws = torch.full((n,), float("nan"), device=dev())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    y = torch.full((rows, fout), float("nan"), device=dev())
    assert lib().mvg_linear_fprop(p(xd), p(wd), p(bd), 1, p(y), rows, fin, fout, p(ws), n, st) == 0
    close(y, torch.relu(x.double() @ w.double().T + b.double()), what="linear fprop")
    ws.fill_(float("nan"))
    dx = torch.full((rows, fin), float("nan"), device=dev())
    assert lib().mvg_linear_dgrad(p(gyd), p(wd), None, None, p(dx), rows, fin, fout, p(ws), n, st) == 0
    close(dx, gy.double() @ w.double(), what="linear dgrad")


def test_gaze_loss_l1_l2_match_reference(golden_dir):
    """GazeLoss(loss_type='l1' | 'l2') (losses/gaze_loss.py:21-29,56-64) through the host class: value and
    the gradient autograd hands back, against the reference's own (tests/golden/lp_loss.npz); 1e-6 relative."""
    from rot_mvgaze_amd.losses import GazeLoss
    g = np.load(os.path.join(golden_dir, "lp_loss.npz"))
    label = torch.from_numpy(g["label"]).to(dev())
    for lt in ("l1", "l2"):
        pred = torch.from_numpy(g["pred"]).to(dev()).requires_grad_(True)
        loss = GazeLoss(gaze_weight=1.0, loss_type=lt)(pred, label)
        (3.0 * loss).backward()
        np.testing.assert_allclose(loss.item(), g[f"{lt}_loss"], rtol=1e-6)
        np.testing.assert_allclose(pred.grad.cpu().numpy(), 3.0 * g[f"{lt}_dpred"], rtol=1e-6, atol=1e-9)
        assert pred.grad[3].abs().max().item() == 0.0           # pred == label: torch.abs has zero gradient there
    with pytest.raises(AssertionError):
        GazeLoss(1.0, "huber")
    with pytest.raises(AssertionError):
        GazeLoss(1.0, "l1")(torch.zeros(4, 3, device=dev()), torch.zeros(4, 3, device=dev()))


def test_rotcat_and_relative_rotation():
    from rot_mvgaze_amd import ops
    B, V, cf, nvec = 5, 3, 512, 512
    pairs = [(i, j) for i in range(V) for j in range(i + 1, V)]
    vi = [x for (i, j) in pairs for x in (i, j)]
    vj = [x for (i, j) in pairs for x in (j, i)]
    D = len(vi)
    hp = torch.from_numpy((synth.uniform01(B * V * 2, 3, "hp") - 0.5).astype(np.float32)).reshape(B * V, 2)
    from oracle import restatement as R
    rot = R.rotation_matrix_2d(hp).reshape(B, V, 3, 3)
    rel_ref = torch.stack([rot[:, vi[d]] @ rot[:, vj[d]].transpose(-1, -2) for d in range(D)])
    vid, vjd = torch.tensor(vi, dtype=torch.int32, device=dev()), torch.tensor(vj, dtype=torch.int32, device=dev())
    rel = torch.empty(D, B, 3, 3, device=dev())
    ops.relative_rotation(rot.to(dev()), vid, vjd, rel, B, V, D)
    close(rel, rel_ref, 1e-6, "relative rotation")

    img_feat, feat = rnd((V, B, cf), 1), rnd((D, B, 3, nvec), 2)
    src = torch.tensor([d ^ 1 for d in range(D)], dtype=torch.int32, device=dev())
    x = torch.empty(D, B, cf + 3 * nvec, device=dev())
    ops.rotcat_fwd(img_feat.to(dev()), feat.to(dev()), rel, vid, src, x, B, D, cf, nvec)
    x_ref = torch.stack([torch.cat([img_feat[vi[d]], (rel_ref[d] @ feat[d ^ 1]).flatten(-2, -1)], -1) for d in range(D)])
    close(x, x_ref, 1e-6, "rotcat fwd")
    gx = rnd((D, B, cf + 3 * nvec), 3)
    dfeat = torch.empty(D, B, 3, nvec, device=dev())
    ops.rotcat_bwd(gx.to(dev()), rel, src, dfeat, B, D, cf, nvec)
    df_ref = torch.empty(D, B, 3, nvec)
    for d in range(D):
        df_ref[d ^ 1] = rel_ref[d].transpose(-1, -2) @ gx[d, :, cf:].reshape(B, 3, nvec)
    close(dfeat, df_ref, 1e-6, "rotcat bwd")
    dimg = torch.ones(V, B, cf, device=dev())
    ops.segment_sum(gx.to(dev()), cf + 3 * nvec, cf, vid, dimg, B, D, V, True)
    di_ref = torch.ones(V, B, cf)
    for d in range(D):
        di_ref[vi[d]] += gx[d, :, :cf]
    close(dimg, di_ref, 1e-6, "segment sum")
    a, b = rnd((1000,), 5).to(dev()), rnd((1000,), 6).to(dev())
    b0 = b.clone()
    ops.axpby(a, b, 2.0, 0.5)
    close(b, 2 * a + 0.5 * b0, 1e-6, "axpby")


def test_multi_erase_kernel_matches_reference_fixture(golden_dir):
    """mvg_multi_erase_nchw against images the reference's RandomMultiErasing produced (bit-exact:
    the kernel only multiplies by 0/1 with the reference's nearest-neighbour index rule)."""
    import random
    from rot_mvgaze_amd.augment import RandomMultiErasing
    from rot_mvgaze_amd import synth
    g = np.load(os.path.join(golden_dir, "multi_erase.npz"))
    random.seed(7)
    np.random.seed(7)
    torch.manual_seed(7)
    aug = RandomMultiErasing(p=0.5, proportion=[0.5, 0.6], dot_size=[0.05, 0.3])
    imgs = torch.from_numpy(synth.normal(10 * 3 * 40 * 56, 77, "erase").reshape(10, 3, 40, 56).astype(np.float32))
    out = aug(imgs.to(dev()))
    assert np.array_equal(out.cpu().numpy(), g["out"])


RESIZE_CASES = [(2, 37, 41, 24), (1, 100, 90, 64), (1, 50, 60, 96), (1, 96, 96, 48), (1, 64, 64, 64)]   # n, h, w, size
IMAGE_MEAN, IMAGE_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _resize_gpu(u8, size, swap=False):
    from rot_mvgaze_amd import ops
    n, h, w, _ = u8.shape
    dst = torch.empty(n, size, size, 4, device=dev())
    ops.preprocess_u8hwc_resize(torch.from_numpy(u8).to(dev()), dst, n, h, w, size, size, IMAGE_MEAN, IMAGE_STD, swap)
    out = dst.cpu().numpy()
    assert np.all(out[..., 3] == 0)
    return np.ascontiguousarray(out[..., :3].transpose(0, 3, 1, 2))


def test_preprocess_resize_matches_aten_fixture_and_oracle(golden_dir):
    """mvg_preprocess_u8hwc_resize (ToTensor -> Resize((S, S), antialias=True) -> Normalize of
    main.py:50-55, NHWC4 out) against the ATen-generated fixture (tests/golden/resize_aa.npz) and, at the
    benchmark's 224 x 224 output from 256 x 240 and 180 x 200 patches, against the oracle restatement.
    Tolerance 3e-6 absolute on the normalised values (float32 summation order; values are O(1))."""
    from oracle import restatement as R
    g = np.load(os.path.join(golden_dir, "resize_aa.npz"))
    for idx, (n, h, w, size) in enumerate(RESIZE_CASES):
        got = _resize_gpu(g[f"u8_{idx}"], size)
        np.testing.assert_allclose(got, g[f"y_{idx}"], rtol=0, atol=3e-6)
    rng = np.random.default_rng(5)
    for (h, w) in ((256, 240), (180, 200), (224, 224)):
        u8 = rng.integers(0, 256, size=(2, h, w, 3), dtype=np.uint8)
        for swap in (False, True):
            np.testing.assert_allclose(_resize_gpu(u8, 224, swap), R.preprocess_u8(u8, 224, IMAGE_MEAN, IMAGE_STD, swap),
                                       rtol=0, atol=3e-6)


def test_bad_arguments_fail_with_a_message():
    """Error convention of the C ABI: non-zero return code + mvg_last_error() text, surfaced
    by the host layer as RuntimeError - never a silent wrong answer or a device fault."""
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    x = torch.zeros(1, 1, 8, 8, 6, device=dev())
    with pytest.raises(RuntimeError, match="c %% 4|c % 4"):
        ops.bn_apply(x.view(1, 64, 6), torch.ones(1, 6, device=dev()), torch.zeros(1, 6, device=dev()), None, True,
                     torch.empty(1, 64, 6, device=dev()), 1, 64, 6)
    d = ConvDesc.make(1, 1, 8, 8, 8, 6, 3, 1, 1)                       # cout = 6: not a multiple of 4
    with pytest.raises(RuntimeError, match="power-of-two|cout"):
        ops.conv_dgrad(d, torch.zeros(1, 1, 8, 8, 6, device=dev()), torch.zeros(6, 3, 3, 8, device=dev()),
                       torch.empty(1, 1, 8, 8, 8, device=dev()), None, None)
    with pytest.raises(RuntimeError, match="out_features"):
        ops.linear_skinny_fwd(torch.zeros(4, 8, device=dev()), torch.zeros(5, 8, device=dev()), torch.zeros(5, device=dev()),
                              torch.empty(4, 5, device=dev()), 4, 8, 5)
    bad = ConvDesc.make(1, 1, 8, 8, 8, 8, 3, 1, 1)
    bad.ho = 3                                                          # inconsistent output size
    with pytest.raises(RuntimeError):
        ops.conv_fprop(bad, torch.zeros(1, 1, 8, 8, 8, device=dev()), torch.zeros(8, 3, 3, 8, device=dev()),
                       torch.empty(1, 1, 3, 8, 8, device=dev()), None, False, None)


@pytest.mark.parametrize("G,P,C,rows", [(4, 98, 512, 98 * 64), (8, 1219, 64, 1219 * 64 - 17), (1, 40, 256, 40 * 64), (3, 2500, 128, 2500 * 64 - 3)])
def test_bn_finalize_with_and_without_registered_scratch_agree_bit_for_bit(G, P, C, rows):
    """mvg_bn_finalize has two forms: with a registered workspace, one workgroup per (8 channels, group) + a
    running-statistics kernel (and slices above 1024 partials); without one, a single kernel that walks the groups.
    Same arithmetic in the same order: identical mean / invstd / scale / shift and running statistics."""
    import ctypes as C_
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import lib
    torch.manual_seed(G * P + C)
    stats = torch.randn(G, P, 2, C, device=dev())
    stats[:, :, 1].abs_()                                   # centred sums of squares are non-negative
    gamma, beta = torch.rand(C, device=dev()) + 0.5, torch.randn(C, device=dev())
    outs = []
    for with_scratch in (True, False):
        rm, rv = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
        o = [torch.full((G, C), float("nan"), device=dev()) for _ in range(4)]
        if with_scratch:
            ops.bn_finalize(stats, G, P, 64, rows, C, gamma, beta, rm, rv, 0.1, 1e-5, *o)
        else:
            side = torch.cuda.Stream()                       # a stream nobody registered a workspace for
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                rc = lib().mvg_bn_finalize(C_.c_void_p(stats.data_ptr()), G, P, 64, rows, C, C_.c_void_p(gamma.data_ptr()),
                                           C_.c_void_p(beta.data_ptr()), C_.c_void_p(rm.data_ptr()), C_.c_void_p(rv.data_ptr()),
                                           C_.c_float(0.1), C_.c_float(1e-5), *[C_.c_void_p(t.data_ptr()) for t in o],
                                           C_.c_void_p(side.cuda_stream))
            assert rc == 0
            side.synchronize()
        torch.cuda.synchronize()
        outs.append(o + [rm, rv])
    for a, b, name in zip(outs[0], outs[1], ("mean", "invstd", "scale", "shift", "running_mean", "running_var")):
        assert torch.equal(a, b), name
