"""Split-operand kernels (csrc/conv_split.hip): fp32 values held as two fp16 pieces and a per-tensor power-of-two scale
("sp": |v - pieces| <= 2^-23 |v|: at most the last significand bit is lost), three fp16 MFMAs per product.  These kernels serve the fp32 model
(the 1e-4 parity path), so the bar is the fp32-MFMA kernels' own: errors against an fp64 torch reference of F.conv2d
and its autograd backward (resnet.py:31-47) at fp32-rounding level (2e-6 relative L2 and never more than 3x the
fp32-MFMA kernel's own error on the same inputs), and the BatchNorm passes that write sp equal to the fp32 passes they
mirror to that last bit."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SPLIT_VS_F64 = 2e-6        # relative L2 against fp64; measured 1e-7 .. 7e-7 (the fp32-MFMA kernels: the same range)
SPLIT_VS_FP32_KERNEL = 3.0  # ... and never more than 3x the fp32-MFMA kernel's own error on the same inputs (+ 1e-7)


def dev():
    return torch.device("cuda:0")


def rel_l2(a, ref):
    return ((a.double() - ref).norm() / ref.norm()).item()


SP_ULP = 2.0 ** -23            # |v - (h1 + h2) 2^-k| <= 2^-23 |v| while the second piece is a normal fp16 number


def sp_close(got, want, what=""):
    """got == want to the sp format's last bit (relative 2^-23), with an absolute floor of 2^-25 for values whose second
    piece is an fp16 subnormal (|v| < 2^-3 of an unscaled tensor)."""
    err = (got.double() - want.double()).abs()
    bound = want.double().abs() * SP_ULP * 1.0001 + 2.0 ** -25
    assert bool((err <= bound).all()), f"{what}: max excess {float((err - bound).max()):.3e}"


def test_sp_round_trip_loses_at_most_the_last_bit():
    from rot_mvgaze_amd import ops
    torch.manual_seed(1)
    x = torch.randn(3, 5, 7, 64, device=dev()) * torch.logspace(-2, 3, 64, device=dev())
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 60000.0, 2.0 ** -10, 1e-3, -7.0], device=dev())
    s = ops.split_f32(x)
    assert s.shape == (3, 5, 7, 8, 2, 8) and s.dtype == torch.float16 and getattr(s, "sinv", None) is None
    back = ops.merge_sp(s)
    big = x.abs() >= 2.0 ** -2                     # (below: the second piece turns subnormal, absolute error <= 2^-25)
    assert bool(((back - x).abs()[big] <= x.abs()[big] * SP_ULP).all())
    assert float((back - x).abs().max()) <= max(float(x.abs().max()) * SP_ULP, 2.0 ** -25)
    assert float((back == x)[big].float().mean()) > 0.7     # three values in four come back exactly
    # the pieces are what the definition says: fp16(a), fp16(a - a1)
    a1 = x.to(torch.float16)
    a2 = (x - a1.float()).to(torch.float16)
    want = torch.stack([a1.view(3, 5, 7, 8, 8), a2.view(3, 5, 7, 8, 8)], dim=-2)
    assert torch.equal(s, want)
    # a tensor outside fp16's range goes through its power-of-two scale: gradients of 1e-9, weights of 1e6
    for mag, scale in ((1e-9, 2.0 ** 40), (1e6, 2.0 ** -8)):
        t = torch.randn(4, 9, 128, device=dev()) * mag
        st = ops.split_f32(t, scale)
        assert st.sinv is not None and float(st.sinv) == 1.0 / scale
        bt = ops.merge_sp(st)
        sel = t.abs() * scale >= 0.25
        assert bool(((bt - t).abs()[sel] <= t.abs()[sel] * SP_ULP).all())


def test_split_weights_carry_their_scale():
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    torch.manual_seed(2)
    for mag in (1.0, 1e-3, 300.0):
        d = ConvDesc.make(1, 1, 8, 8, 64, 128, 3, 1, 1)
        w = torch.randn(128, 3, 3, 64, device=dev()) * mag
        wk, wt = ops.split_weights(d, w, True)
        assert wk.sinv is wt.sinv or torch.equal(wk.sinv, wt.sinv)
        sinv = float(wk.sinv)
        k = np.log2(sinv)
        assert k == round(k) and 2.0 ** 14 <= float(w.abs().max()) / sinv < 2.0 ** 15        # max |w| 2^k just below 2^15
        for got, want, what in ((ops.merge_sp(wk).view(128, 3, 3, 64), w, "KRSC copy"),
                                (ops.merge_sp(wt).view(64, 3, 3, 128), w.permute(3, 1, 2, 0).contiguous(), "CRSK copy")):
            err = (got.double() - want.double()).abs()
            assert bool((err <= want.double().abs() * SP_ULP + 2.0 ** -25 * sinv).all()), what


CONV_CASES = [
    # G, N, h, cin, cout, k, stride, pad
    (2, 3, 14, 256, 256, 3, 1, 1), (1, 5, 28, 128, 128, 3, 2, 1), (3, 2, 14, 1024, 256, 1, 1, 0), (2, 4, 28, 256, 512, 1, 2, 0),
    (2, 3, 56, 64, 64, 3, 1, 1), (1, 2, 56, 64, 256, 1, 1, 0), (2, 7, 7, 512, 2048, 1, 1, 0), (1, 1, 9, 32, 64, 3, 1, 1),
    (2, 1, 5, 64, 32, 3, 2, 1),        # odd map, stride 2: ragged parity classes, a class with a single tap
    (1, 3, 1, 64, 128, 1, 1, 0),       # 1x1 map: 3 rows in a 128-row tile
    (4, 9, 15, 96, 160, 1, 1, 0),      # channel counts that are multiples of 32 but not powers of two (1x1 only)
    # few tiles, long K (less than one resident round of workgroups)
    (2, 16, 14, 256, 256, 3, 1, 1), (2, 16, 14, 1024, 256, 1, 1, 0), (2, 16, 28, 256, 256, 3, 2, 1), (3, 20, 7, 512, 64, 3, 1, 1),
    # >= 65536 rows per group with 64 GEMM columns and several taps: the 256 x 64 tile (four wave rows); ragged last tile, stride 2
    (2, 24, 56, 64, 64, 3, 1, 1), (1, 21, 57, 64, 64, 3, 1, 1), (1, 22, 112, 64, 64, 3, 2, 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=["g%d_n%d_h%d_%dto%d_k%d_s%d" % c[:7] for c in CONV_CASES])
def test_split_conv_fprop_dgrad_wgrad(case):
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    G, N, h, cin, cout, k, st, pad = case
    torch.manual_seed(sum(case))
    d = ConvDesc.make(G, N, h, h, cin, cout, k, st, pad)
    x = torch.relu(torch.randn(G, N, h, h, cin, device=dev()))
    w = torch.randn(cout, k, k, cin, device=dev()) * (1.0 / (k * k * cin) ** 0.5)
    gy = torch.randn(G, N, d.ho, d.wo, cout, device=dev())
    add = torch.randn_like(x)
    xs, gys = ops.split_f32(x), ops.split_f32(gy * 2.0 ** -20, 2.0 ** 28)      # a gradient-sized dy with its scale
    gy = gy * 2.0 ** -20
    wk, wt = ops.split_weights(d, w, True)
    # fp64 reference
    xr = x.double().view(G * N, h, h, cin).permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, st, pad)
    yr.backward(gy.double().view(G * N, d.ho, d.wo, cout).permute(0, 3, 1, 2))
    y_ref = yr.detach().permute(0, 2, 3, 1).reshape(G, N, d.ho, d.wo, cout)
    dx_ref = xr.grad.permute(0, 2, 3, 1).reshape(x.shape) + add.double()
    dw_ref = wr.grad.permute(0, 2, 3, 1)

    def both(name, split_fn, fp32_fn, ref, shape):
        a, b = torch.empty(shape, device=dev()), torch.empty(shape, device=dev())
        split_fn(a)
        fp32_fn(b)
        ea, eb = rel_l2(a, ref), rel_l2(b, ref)
        assert ea <= SPLIT_VS_F64 and ea <= SPLIT_VS_FP32_KERNEL * eb + 1e-7, f"{name}: split {ea:.2e}, fp32-MFMA kernel {eb:.2e}"
        return a

    P, rpp = ops.conv_stats_partials_split(d)
    stats = torch.full((G, P, 2, cout), float("nan"), device=dev())
    y = both("fprop", lambda o: ops.conv_fprop_split(d, xs, wk, o, stats), lambda o: ops.conv_fprop(d, x, w, o, None, False, None),
             y_ref, y_ref.shape)
    # the statistics partials describe the rows they cover: sums, and squares centred on the partial's own mean
    rows = N * d.ho * d.wo
    yg = y.view(G, rows, cout).double()
    for p in range(P):
        r0, r1 = p * rpp, min((p + 1) * rpp, rows)
        if r0 >= rows:
            continue
        blk = yg[:, r0:r1]
        np.testing.assert_allclose(stats[:, p, 0].cpu().double().numpy(), blk.sum(1).cpu().numpy(), rtol=1e-4,
                                   atol=1e-4 * float(blk.abs().sum(1).max()))
        q = ((blk - blk.mean(1, keepdim=True)) ** 2).sum(1)
        np.testing.assert_allclose(stats[:, p, 1].cpu().double().numpy(), q.cpu().numpy(), rtol=1e-3, atol=1e-5 * float(q.max()) + 1e-12)
    both("dgrad", lambda o: ops.conv_dgrad_split(d, gys, wt, o, add), lambda o: ops.conv_dgrad(d, gy, w, o, None, add), dx_ref, x.shape)
    both("wgrad", lambda o: ops.conv_wgrad_split(d, xs, gys, o), lambda o: ops.conv_wgrad(d, x, gy, o), dw_ref, w.shape)
    # accumulate into an existing gradient; dgrad accumulating in place (addend aliases dx: the downsample branch)
    dw = torch.ones_like(w)
    ops.conv_wgrad_split(d, xs, gys, dw, True)
    assert rel_l2(dw, dw_ref + 1.0) <= SPLIT_VS_F64
    dx = add.clone()
    ops.conv_dgrad_split(d, gys, wt, dx, dx)
    assert rel_l2(dx, dx_ref) <= SPLIT_VS_F64


@pytest.mark.parametrize("G,N,H,C,res", [(2, 3, 9, 64, "s3"), (2, 2, 7, 2048, None), (3, 2, 28, 128, "raw"), (1, 4, 12, 256, "s3")])
def test_batchnorm_passes_writing_sp_match_the_fp32_passes(G, N, H, C, res):
    """bn_apply_split / bn_bwd_apply_split / the stem's pooled map / avgpool over sp: the same numbers as the fp32
    kernels to the format's last bit; the ReLU mask bits identical; dy comes with a power-of-two scale that keeps its
    pieces finite for O(1) and for 1e-7-sized gradients alike."""
    from rot_mvgaze_amd import ops
    torch.manual_seed(G * 100 + C)
    rows = N * H * H
    y = torch.randn(G, rows, C, device=dev()) * 2 + 0.5
    scale, shift = torch.rand(G, C, device=dev()) + 0.5, torch.randn(G, C, device=dev()) * 0.3
    r = torch.randn(G, rows, C, device=dev()) if res else None
    raff = (torch.rand(G, C, device=dev()) + 0.5, torch.randn(G, C, device=dev()) * 0.2) if res == "raw" else None
    want = torch.empty_like(y)
    if res:
        want_bits = ops.bn_apply_bits(y, scale, shift, r, want, G, rows, C, raff)
    else:
        ops.bn_apply(y, scale, shift, None, True, want, G, rows, C)
    out = ops.sp_empty(G, rows, C, device=dev())
    rs = ops.split_f32(r) if res == "s3" else r
    if res == "s3":                                 # the fp32 pass sees the same residual values the sp pass does
        r = ops.merge_sp(rs)
        want_bits = ops.bn_apply_bits(y, scale, shift, r, want, G, rows, C, raff)
    bits = ops.bn_apply_split(y, scale, shift, rs, True, out, G, rows, C, raff, want_bits=bool(res))
    sp_close(ops.merge_sp(out), want, "bn_apply_split")
    if res:
        assert torch.equal(bits, want_bits)
    # backward apply (mask from the forward's affine, or an already masked gradient)
    g = torch.randn(G, rows, C, device=dev())
    mean, invstd = torch.randn(G, C, device=dev()) * 0.1 + 0.5, torch.rand(G, C, device=dev()) + 0.3
    gamma = torch.rand(C, device=dev()) + 0.5
    s1, s2 = torch.randn(G, C, device=dev()), torch.randn(G, C, device=dev())
    for gmag in (1.0, 1e-7):                     # O(1) test gradients and realistic tiny ones: the scale follows
        for ra in ((scale, shift), None):
            gg = g * gmag
            s1m, s2m = s1 * gmag, s2 * gmag
            dy_want = torch.empty_like(gg)
            ops.bn_bwd_apply(gg, None, y, mean, invstd, gamma, s1m, s2m, G, rows, C, dy_want, None, ra)
            # max |masked gradient| per (group, channel), as the reduce pass would leave it (here: of the unmasked g, an upper bound)
            mx = gg.abs().amax(dim=1).contiguous()
            dy = ops.sp_empty(G, rows, C, device=dev())
            ops.bn_bwd_apply_split(gg, y, mean, invstd, gamma, s1m, s2m, G, rows, C, dy, ra, mx)
            k = np.log2(float(dy.sinv))
            assert k == round(k)
            assert float(dy.float().abs().max()) < 65504.0                          # the bound kept every piece finite
            # same expression, separately compiled (fma contraction may differ): equal to fp32 rounding
            got = ops.merge_sp(dy)
            assert (got - dy_want).abs().max().item() <= 1e-6 * dy_want.abs().max().item()
    # average pool over an sp map
    feat, feat_want = torch.empty(G * N, C, device=dev()), torch.empty(G * N, C, device=dev())
    ops.avgpool_fwd(ops.merge_sp(out), feat_want, G * N, H * H, C)
    ops.avgpool_fwd_split(out, feat, G * N, H * H, C)
    assert (feat - feat_want).abs().max().item() <= 1e-6 * feat_want.abs().max().item()


def test_stem_tail_writing_sp_matches_the_fp32_kernel():
    from rot_mvgaze_amd import ops
    G, N, H, C = 2, 3, 30, 64
    torch.manual_seed(3)
    y = torch.randn(G, N, H, H, C, device=dev())
    scale, shift = torch.rand(G, C, device=dev()) + 0.5, torch.randn(G, C, device=dev()) * 0.3
    hp = (H + 2 - 3) // 2 + 1
    want, am_want = torch.empty(G, N, hp, hp, C, device=dev()), torch.empty(G, N, hp, hp, C, dtype=torch.uint8, device=dev())
    ops.bn_relu_maxpool_fwd(y, scale, shift, want, am_want, G, N, H, H, C, hp, hp)
    out, am = ops.sp_empty(G, N, hp, hp, C, device=dev()), torch.empty_like(am_want)
    ops.bn_relu_maxpool_fwd_split(y, scale, shift, out, am, G, N, H, H, C, hp, hp)
    sp_close(ops.merge_sp(out), want, "pooled map")
    assert torch.equal(am, am_want)


@pytest.mark.parametrize("case", [(2, 3, 14, 256, 256, 3, 1, 1), (3, 2, 14, 1024, 256, 1, 1, 0), (2, 3, 56, 64, 64, 3, 1, 1),
                                  (1, 2, 56, 64, 256, 1, 1, 0), (2, 7, 7, 512, 512, 3, 1, 1), (2, 5, 9, 128, 160, 1, 1, 0),
                                  # stride 2: four parity classes in one launch, each with its own partials; a 1x1 filter
                                  # leaves three of them without taps (epilogue-only tiles); odd maps: ragged classes
                                  (1, 5, 28, 128, 128, 3, 2, 1), (2, 3, 15, 64, 128, 3, 2, 1), (2, 3, 56, 256, 512, 1, 2, 0),
                                  (1, 4, 9, 128, 256, 1, 2, 0),
                                  # the 256 x 64 tile (>= 65536 rows, 64 columns, 3x3): one partial per 256-row tile; stride 2
                                  (2, 24, 56, 64, 64, 3, 1, 1), (1, 21, 57, 64, 64, 3, 1, 1), (1, 22, 112, 64, 128, 3, 2, 1)],
                         ids=lambda c: "g%d_n%d_h%d_%dto%d_k%d_s%d" % c[:7])
@pytest.mark.parametrize("mask", ["bits", "affine", "none"])
def test_split_dgrad_fused_with_bn_backward_reduce(case, mask):
    """mvg_conv_dgrad_split_bnreduce == mvg_conv_dgrad_split followed by the reduce pass over its result: the same
    masked gradient bit for bit, the same sums to summation order."""
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    G, N, h, cin, cout, k, st, pad = case
    torch.manual_seed(sum(case) + len(mask))
    d = ConvDesc.make(G, N, h, h, cin, cout, k, st, pad)
    rows = N * h * h
    w = torch.randn(cout, k, k, cin, device=dev()) * (1.0 / (k * k * cout) ** 0.5)
    gy = torch.randn(G, N, d.ho, d.wo, cout, device=dev())
    add = torch.randn(G, N, h, h, cin, device=dev())
    _, wt = ops.split_weights(d, w, True)
    gys = ops.split_f32(gy)
    # the unit whose output gradient dx is: its conv output y, statistics, and ReLU mask
    y = torch.randn(G, rows, cin, device=dev()) * 1.5 + 0.3
    mean, invstd = torch.randn(G, cin, device=dev()) * 0.1 + 0.3, torch.rand(G, cin, device=dev()) + 0.4
    scale, shift = torch.rand(G, cin, device=dev()) + 0.5, torch.randn(G, cin, device=dev()) * 0.3
    bits = torch.randint(0, 16, (G * rows * cin // 4,), dtype=torch.uint8, device=dev()) if mask == "bits" else None
    ra = (scale, shift) if mask == "affine" else None
    # reference: two launches
    dx_ref = torch.empty(G, N, h, h, cin, device=dev())
    ops.conv_dgrad_split(d, gys, wt, dx_ref, add)
    s_ref = [torch.empty(G, cin, device=dev()) for _ in range(2)]
    dg_ref, db_ref = torch.full((cin,), 0.5, device=dev()), torch.full((cin,), -0.25, device=dev())
    g2 = dx_ref.view(G, rows, cin)
    am_ref = torch.full((G, cin), float("nan"), device=dev())
    ops.bn_bwd_reduce_split(g2, bits, y, mean, invstd, G, rows, cin, s_ref[0], s_ref[1], dg_ref, db_ref, True, am_ref, ra, dz_out=g2)
    # fused
    dx = torch.empty_like(dx_ref)
    s = [torch.empty(G, cin, device=dev()) for _ in range(2)]
    dg, db = torch.full((cin,), 0.5, device=dev()), torch.full((cin,), -0.25, device=dev())
    am = torch.full((G, cin), float("nan"), device=dev())
    ops.conv_dgrad_split_bnreduce(d, gys, wt, dx, add, y, bits, mean, invstd, ra, s[0], s[1], dg, db, True, am)
    assert torch.equal(dx, dx_ref), "masked gradient"
    assert torch.equal(am, am_ref) and torch.equal(am, dx.view(G, rows, cin).abs().amax(dim=1)), "max |masked gradient| per (group, channel)"
    for got, want, name in ((s[0], s_ref[0], "s1"), (s[1], s_ref[1], "s2"), (dg, dg_ref, "dgamma"), (db, db_ref, "dbeta")):
        err = (got - want).abs().max().item()
        assert err <= 2e-5 * max(want.abs().max().item(), 1.0) * (rows ** 0.5), f"{name}: {err:.3e}"
    # the addend aliasing the result (the downsample branch adds its gradient into the main branch's): same answer
    dx2 = add.clone()
    s2 = [torch.empty(G, cin, device=dev()) for _ in range(2)]
    ops.conv_dgrad_split_bnreduce(d, gys, wt, dx2, dx2, y, bits, mean, invstd, ra, s2[0], s2[1], None, None, False, None)
    assert torch.equal(dx2, dx_ref), "masked gradient, in-place addend"
    assert torch.equal(s2[0], s[0]) and torch.equal(s2[1], s[1])
    # ONE finalize launch (partials -> s1 / s2 / max, dgamma / dbeta and dy's scale, folded into the last-arriving workgroups)
    # against the three-launch form: the reduce entry given the unit's gamma leaves the 2^-k that mvg_bn_bwd_apply_split computes
    gamma = torch.rand(cin, device=dev()) + 0.5
    dx3 = torch.empty_like(dx_ref)
    s3 = [torch.empty(G, cin, device=dev()) for _ in range(2)]
    dg3, db3 = torch.full((cin,), 0.5, device=dev()), torch.full((cin,), -0.25, device=dev())
    am3 = torch.full((G, cin), float("nan"), device=dev())
    sinv = torch.full((1,), float("nan"), device=dev())
    ops.conv_dgrad_split_bnreduce(d, gys, wt, dx3, add, y, bits, mean, invstd, ra, s3[0], s3[1], dg3, db3, True, am3, gamma, sinv)
    assert torch.equal(dx3, dx) and torch.equal(s3[0], s[0]) and torch.equal(s3[1], s[1]) and torch.equal(am3, am)
    assert torch.equal(dg3, dg) and torch.equal(db3, db)
    dy_a, dy_b = ops.sp_empty(G, rows, cin, device=dev()), ops.sp_empty(G, rows, cin, device=dev())
    ops.bn_bwd_apply_split(dx.view(G, rows, cin), y, mean, invstd, gamma, s[0], s[1], G, rows, cin, dy_a, None, am)            # computes the scale itself
    ops.bn_bwd_apply_split(dx.view(G, rows, cin), y, mean, invstd, gamma, s[0], s[1], G, rows, cin, dy_b, None, am, sinv)      # takes the finalize launch's
    assert float(sinv) == float(dy_a.sinv) and torch.equal(dy_a, dy_b)


@pytest.mark.parametrize("case", [(2, 3, 14, 256, 256, 3, 1, 1), (1, 5, 28, 128, 128, 3, 2, 1), (3, 2, 14, 1024, 256, 1, 1, 0),
                                  (1, 2, 56, 64, 256, 1, 1, 0), (2, 7, 7, 512, 2048, 1, 1, 0), (2, 2, 7, 512, 512, 3, 1, 1)],
                         ids=lambda c: "g%d_n%d_h%d_%dto%d_k%d_s%d" % c[:7])
@pytest.mark.parametrize("res", [None, "s3", "fp32"])
@pytest.mark.parametrize("relu,out_sp", [(True, True), (False, False)])
def test_split_inference_forward_with_folded_batchnorm(case, res, relu, out_sp):
    """mvg_conv_fprop_split_affine (out = relu?(conv * scale + shift (+ residual)), result and residual in s3 or fp32)
    against float64 and against the fp32-MFMA inference kernel mvg_conv_fprop_affine."""
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    G, N, h, cin, cout, k, st, pad = case
    torch.manual_seed(sum(case))
    d = ConvDesc.make(G, N, h, h, cin, cout, k, st, pad)
    x = torch.relu(torch.randn(G, N, h, h, cin, device=dev()))
    w = torch.randn(cout, k, k, cin, device=dev()) * (1.0 / (k * k * cin) ** 0.5)
    scale, shift = torch.rand(cout, device=dev()) + 0.5, torch.randn(cout, device=dev()) * 0.3
    r = torch.randn(G, N, d.ho, d.wo, cout, device=dev()) if res else None
    ref = F.conv2d(x.double().view(G * N, h, h, cin).permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, st, pad).permute(0, 2, 3, 1)
    ref = ref.reshape(G, N, d.ho, d.wo, cout) * scale.double() + shift.double()
    if res:
        ref = ref + r.double()
    if relu:
        ref = torch.relu(ref)
    want = torch.empty(G, N, d.ho, d.wo, cout, device=dev())
    ops.conv_fprop_affine(d, x, w, want, scale, shift, r, relu)
    wk, _ = ops.split_weights(d, w, False)
    out = ops.sp_empty(G, N, d.ho, d.wo, cout, device=dev()) if out_sp else torch.empty_like(want)
    ops.conv_fprop_split_affine(d, ops.split_f32(x), wk, out, scale, shift, ops.split_f32(r) if res == "s3" else r, relu)
    got = ops.merge_sp(out) if out_sp else out
    e_split, e_fp32 = rel_l2(got, ref), rel_l2(want, ref)
    assert e_split <= SPLIT_VS_F64 and e_split <= SPLIT_VS_FP32_KERNEL * e_fp32 + 1e-7, f"split {e_split:.2e}, fp32-MFMA kernel {e_fp32:.2e}"


def test_split_kernels_reject_shapes_they_do_not_cover():
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    d = ConvDesc.make(1, 2, 8, 8, 48, 64, 1, 1, 0)            # cin not a multiple of 32
    with pytest.raises(RuntimeError, match="multiples of 32"):
        ops.split_weights(d, torch.randn(64, 1, 1, 48, device=dev()), True)
    with pytest.raises(AssertionError):
        ops.split_f32(torch.randn(4, 12, device=dev()))       # channels not a multiple of 8


@pytest.mark.parametrize("G,N,H,W,cout", [(2, 3, 32, 32, 64), (1, 2, 224, 224, 64), (2, 2, 23, 40, 64), (1, 5, 7, 6, 64), (3, 1, 96, 64, 64)],
                         ids=lambda v: str(v))
def test_stem_in_row_window_form_on_the_split_kernels(G, N, H, W, cout):
    """mvg_stem_rowwindow_split / _fprop_split / _wgrad_split: the 7x7 stride-2 stem as a 7 x 1 filter over windows of 8
    columns x 4 stored channels == conv2d in float64 on the 3-channel image (and never worse than 3x the fp32-MFMA kernel
    on 4-channel taps); odd heights, ragged row tiles, image borders inside the first / last window."""
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    torch.manual_seed(G * 1000 + H + W)
    d = ConvDesc.make(G, N, H, W, 4, cout, 7, 2, 3)
    x = torch.randn(G, N, H, W, 4, device=dev())
    x[..., 3] = 0
    w = torch.randn(cout, 7, 7, 3, device=dev()) * 0.08
    gy = torch.randn(G, N, d.ho, d.wo, cout, device=dev()) * 3e-4
    xr = x[..., :3].double().view(G * N, H, W, 3).permute(0, 3, 1, 2)
    wr = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 2, 3)
    yr.backward(gy.double().view(G * N, d.ho, d.wo, cout).permute(0, 3, 1, 2))
    y_ref, dw_ref = yr.detach().permute(0, 2, 3, 1).reshape(G, N, d.ho, d.wo, cout), wr.grad.permute(0, 2, 3, 1)
    # the window operand holds the image exactly (to the sp format's last bit), zero outside it
    xw = ops.stem_rowwindow_split(x)
    win = ops.merge_sp(xw).view(G, N, H, W // 2, 8, 4)
    want = torch.zeros_like(win)
    for j in range(8):
        cols = torch.arange(W // 2, device=dev()) * 2 - 4 + j
        ok = (cols >= 0) & (cols < W)
        want[:, :, :, ok, j] = x[:, :, :, cols[ok]]
    sp_close(win, want, "row windows")
    xw2 = ops.sp_empty(G, N, H, W // 2, 32, device=dev())
    for g in range(G):                                  # ... and straight from the module's NCHW input, view by view
        ops.stem_rowwindow_split_nchw(x[g, ..., :3].permute(0, 3, 1, 2).contiguous(), xw2[g])
    assert torch.equal(xw2, xw), "windows from NCHW == windows from NHWC4"

    # weights in the window's tap layout
    w8 = torch.zeros(cout, 7, 8, 4, device=dev())
    w8[:, :, 1:, :3] = w
    dk = ConvDesc(1, 1, 7, 1, 32, cout, 7, 1, 1, 0, 1, 1)
    wk, _ = ops.split_weights(dk, w8.view(cout, 7, 1, 32).contiguous(), False)
    y = torch.full((G, N, d.ho, d.wo, cout), float("nan"), device=dev())
    P, rpp = ops.conv_stats_partials_split(ConvDesc.make(G, N, d.ho, d.wo, 32, cout, 1, 1, 0))
    stats = torch.full((G, P, 2, cout), float("nan"), device=dev())
    ops.stem_fprop_split(d, xw, wk, y, stats)
    w4 = torch.zeros(cout, 7, 7, 4, device=dev())
    w4[..., :3] = w
    y32 = torch.empty_like(y)
    ops.conv_fprop(d, x, w4, y32, None, False, None)
    e, e32 = rel_l2(y, y_ref), rel_l2(y32, y_ref)
    assert e <= SPLIT_VS_F64 and e <= SPLIT_VS_FP32_KERNEL * e32 + 1e-7, f"fprop: split {e:.2e}, fp32-MFMA {e32:.2e}"
    rows = N * d.ho * d.wo
    assert rel_l2(stats[:, :, 0].sum(1), y_ref.reshape(G, rows, cout).sum(1)) <= 1e-5, "BatchNorm partial sums"
    # weight gradient, dy scaled by a power of two
    gscale = 2.0 ** float(torch.floor(torch.log2(2.0 ** 14 / gy.abs().max())))
    gys = ops.split_f32(gy, gscale)
    dw8 = torch.full((cout, 7, 8, 4), float("nan"), device=dev())
    ops.stem_wgrad_split(d, xw, gys, dw8, False)
    dw32 = torch.empty(cout, 7, 7, 4, device=dev())
    ops.conv_wgrad(d, x, gy, dw32, False)
    e, e32 = rel_l2(dw8[:, :, 1:, :3], dw_ref), rel_l2(dw32[..., :3], dw_ref)
    assert e <= SPLIT_VS_F64 and e <= SPLIT_VS_FP32_KERNEL * e32 + 1e-7, f"wgrad: split {e:.2e}, fp32-MFMA {e32:.2e}"
    # (the padding tap j = 0 sees real pixels: its "gradient" is not zero, the caller drops it)
    ops.stem_wgrad_split(d, xw, gys, dw8, True)
    assert rel_l2(dw8[:, :, 1:, :3], 2 * dw_ref) <= SPLIT_VS_F64, "wgrad accumulate"


def test_stem_tail_backward_writing_sp_matches_the_fp32_kernels():
    """mvg_bn_relu_maxpool_bwd_reduce_split / _apply_split == the fp32 pair, with dy in sp at the scale the reduce pass's
    bound allows (4 x the largest masked window gradient per channel bounds a pixel's gradient)."""
    from rot_mvgaze_amd import ops
    G, N, H, C = 2, 3, 30, 64
    torch.manual_seed(5)
    y = torch.randn(G, N, H, H, C, device=dev())
    scale, shift = torch.rand(G, C, device=dev()) + 0.5, torch.randn(G, C, device=dev()) * 0.3
    mean, invstd = torch.randn(G, C, device=dev()) * 0.1, torch.rand(G, C, device=dev()) + 0.5
    gamma = torch.rand(C, device=dev()) + 0.5
    hp = (H + 2 - 3) // 2 + 1
    pooled, am = torch.empty(G, N, hp, hp, C, device=dev()), torch.empty(G, N, hp, hp, C, dtype=torch.uint8, device=dev())
    ops.bn_relu_maxpool_fwd(y, scale, shift, pooled, am, G, N, H, H, C, hp, hp)
    for gmag in (1.0, 1e-6):
        g = torch.randn(G, N, hp, hp, C, device=dev()) * gmag
        s_ref, dg_ref, db_ref = torch.empty(2, G, C, device=dev()), torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
        ops.bn_relu_maxpool_bwd_reduce(g, am, y, mean, invstd, scale, shift, G, N, H, H, C, hp, hp, s_ref[0], s_ref[1], dg_ref, db_ref, False)
        dy_ref = torch.empty_like(y)
        ops.bn_relu_maxpool_bwd_apply(g, am, y, mean, invstd, gamma, scale, shift, s_ref[0], s_ref[1], G, N, H, H, C, hp, hp, dy_ref)
        s, dg, db = torch.empty(3, G, C, device=dev()), torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
        ops.bn_relu_maxpool_bwd_reduce_split(g, am, y, mean, invstd, scale, shift, G, N, H, H, C, hp, hp, s[0], s[1], dg, db, False, s[2])
        assert torch.equal(s[:2], s_ref) and torch.equal(dg, dg_ref) and torch.equal(db, db_ref)
        assert bool((s[2] <= 4 * g.abs().amax(dim=(1, 2, 3)) * 1.0001).all()) and float(s[2].max()) > 0
        dy = ops.sp_empty(G, N, H, H, C, device=dev())
        ops.bn_relu_maxpool_bwd_apply_split(g, am, y, mean, invstd, gamma, scale, shift, s[0], s[1], G, N, H, H, C, hp, hp, dy, s[2])
        k = float(dy.sinv)
        assert k > 0 and np.log2(k) == round(np.log2(k)), "dy scale: a power of two"
        got = ops.merge_sp(dy)
        assert float((got - dy_ref).abs().max()) <= 2.0 ** -22 * float(dy_ref.abs().max()), "dy"
        assert float(dy_ref.abs().max()) / k < 2.0 ** 15.01 and float(dy_ref.abs().max()) / k > 2.0 ** 4, "dy sits in fp16's range with headroom to spare, not far below it"
