"""The fusion block's Linears on the split-operand kernels: range and accuracy (VERDICT r03 item 4 / ADVICE medium).

The reference computes ImageFeatFuser / the gaze head in plain fp32 (/root/reference/models/rot_mv.py:35-50,234-254;
models/backbones/blocks.py:41-47): no range limit.  The sp format (two fp16 pieces) spans 6e-8 .. 65504, so every tensor
these kernels read carries a per-tensor power-of-two scale that its PRODUCER derives on the device (abs-max from the
producing launch's epilogue, or a bound for a hidden activation).  Tolerance: the split kernels' own bar, 2e-6 relative
L2 against fp64 (tests/test_split_gpu.py), at feature magnitudes 1e-6 .. 1e5 - where the unscaled round-3 path lost
everything below 6e-8 and turned everything above 65504 into inf / NaN."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SPLIT_VS_F64 = 2e-6
MAGS = [1e-6, 1e-3, 1.0, 1e3, 1e5]


def dev():
    return torch.device("cuda:0")


def rel_l2(a, ref):
    return ((a.double() - ref).norm() / ref.norm()).item()


def _slots(n=32):
    st = torch.zeros(n, dtype=torch.float32, device=dev())
    return st, [st[i:i + 1] for i in range(n)]


def _sp(rows, cols, slot):
    from rot_mvgaze_amd import ops
    t = ops.sp_empty(rows, cols, device=dev())
    t.sinv = slot
    return t


def _weights_sp(w):
    """sp copies (KRSC, CRSK) of a Linear weight [fout][fin] through the batched prep (scale from max |w|)."""
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    fout, fin = w.shape
    return ops.split_weights(ConvDesc.make(1, 1, 1, 1, fin, fout, 1, 1, 0), w.contiguous(), True)


@pytest.mark.parametrize("mag", MAGS)
@pytest.mark.parametrize("rows,fin,fout", [(1536, 512, 256), (1100, 3584, 512)])
def test_split_linear_fprop_dgrad_wgrad_over_magnitudes(mag, rows, fin, fout):
    """x at magnitude `mag` (and a gradient at 1 / mag): split -> fprop (fp32 result + abs-max, sp hidden result with its
    bound-derived scale), dgrad, wgrad, all against fp64."""
    from rot_mvgaze_amd import ops
    torch.manual_seed(3)
    x = torch.randn(rows, fin, device=dev()) * mag
    w = torch.randn(fout, fin, device=dev()) / fin ** 0.5
    b = torch.randn(fout, device=dev()) * mag
    st, s = _slots()
    ops.absmax_multi([x, b], [s[0], s[1]])
    assert float(s[0]) == float(x.abs().max())
    x_sp = _sp(rows, fin, s[2])
    ops.split_colsum(x, rows, fin, s[0], x_sp)                       # (any fp32 tensor with a known abs-max: the same kernel splits gradients)
    assert float(x.abs().max()) / float(s[2]) < 2.0 ** 15 and float(x.abs().max()) / float(s[2]) >= 2.0 ** 14
    wk, wt = _weights_sp(w)
    ref = x.double() @ w.double().t() + b.double()
    # fp32 result + its abs-max from the epilogue
    y = torch.empty(rows, fout, device=dev())
    ops.linear_fprop_split(x_sp, wk, b, False, y, rows, fin, fout, out_absmax=s[3])
    assert rel_l2(y, ref) < SPLIT_VS_F64
    assert float(s[3]) == float(y.abs().max())
    # sp result (a hidden activation): ReLU, stored scaled from the bound; reading it back through its sinv gives relu(ref)
    h = _sp(rows, fout, s[4])
    ops.linear_fprop_split(x_sp, wk, b, True, h, rows, fin, fout, bias_absmax=s[1])
    hs = float(s[4])
    assert np.log2(hs) == round(np.log2(hs))
    hb = ops.merge_sp(h)
    assert torch.isfinite(hb).all() and rel_l2(hb, ref.clamp_min(0)) < SPLIT_VS_F64
    assert float(hb.abs().max()) / hs < 2.0 ** 15                  # inside fp16's range by construction
    # backward: a gradient at the inverse magnitude, split with the scale its abs-max gives
    g = torch.randn(rows, fout, device=dev()) / mag
    ops.absmax_multi([g], [s[5]])
    g_sp = _sp(rows, fout, s[6])
    db = torch.empty(fout, device=dev())
    ops.split_colsum(g, rows, fout, s[5], g_sp, db)
    assert rel_l2(db, g.double().sum(0)) < 1e-5
    dx = torch.empty(rows, fin, device=dev())
    ops.linear_dgrad_split(g_sp, wt, dx, rows, fin, fout, out_absmax=s[7])
    assert rel_l2(dx, g.double() @ w.double()) < SPLIT_VS_F64
    assert float(s[7]) == float(dx.abs().max())
    # the ReLU mask from the SCALED sp hidden activation (sign only)
    dxm = torch.empty(rows, fout, device=dev())
    g2 = torch.randn(rows, fin, device=dev())
    ops.absmax_multi([g2], [s[8]])
    g2_sp = _sp(rows, fin, s[9])
    ops.split_colsum(g2, rows, fin, s[8], g2_sp)
    wk2, wt2 = _weights_sp(w.t().contiguous())                      # a second layer [fin <- fout]: dgrad gives [rows, fout]
    ops.linear_dgrad_split(g2_sp, wt2, dxm, rows, fout, fin, relu_mask_sp=h)
    want = (g2.double() @ w.double().t()) * (ref > 0)
    assert rel_l2(dxm, want) < SPLIT_VS_F64 * 2                  # (a mask flip on a hidden value within rounding of 0 is not an error of the product)
    # wgrad with both operands scaled
    dw = torch.empty(fout, fin, device=dev())
    ops.linear_wgrad_split(x_sp, g_sp, dw, rows, fin, fout)
    assert rel_l2(dw, g.double().t() @ x.double()) < SPLIT_VS_F64
    # ... and with the scaled hidden activation as the x operand
    dw2 = torch.empty(fin, fout, device=dev())
    ops.linear_wgrad_split(h, g2_sp, dw2, rows, fout, fin)
    assert rel_l2(dw2, g2.double().t() @ ref.clamp_min(0)) < SPLIT_VS_F64


@pytest.mark.parametrize("rows,fin,fout", [(6016, 256, 3584), (384, 32, 128), (384, 96, 256), (1536, 3584, 512), (200, 160, 1536)])
def test_split_linear_k_loop_forms(rows, fin, fout):
    """The Linears' two K-loop forms - the two-stage software pipeline (launches of <= four tiles per CU; one, two, three and
    112 K-steps, i.e. odd and even counts and the prologue-only cases) and the single-stage loop (6016 x 3584 outputs: 1316
    tiles) - forward, with ReLU into sp, and backward-data, against fp64."""
    from rot_mvgaze_amd import ops
    torch.manual_seed(11)
    x = torch.randn(rows, fin, device=dev())
    w = torch.randn(fout, fin, device=dev()) / fin ** 0.5
    b = torch.randn(fout, device=dev())
    g = torch.randn(rows, fout, device=dev())
    st, s = _slots()
    ops.absmax_multi([x, b, g], [s[0], s[1], s[2]])
    x_sp, g_sp, h = _sp(rows, fin, s[3]), _sp(rows, fout, s[4]), _sp(rows, fout, s[5])
    ops.split_colsum(x, rows, fin, s[0], x_sp)
    ops.split_colsum(g, rows, fout, s[2], g_sp)
    wk, wt = _weights_sp(w)
    ref = x.double() @ w.double().t() + b.double()
    y = torch.full((rows, fout), float("nan"), device=dev())
    ops.linear_fprop_split(x_sp, wk, b, False, y, rows, fin, fout, out_absmax=s[6])
    assert rel_l2(y, ref) < SPLIT_VS_F64 and float(s[6]) == float(y.abs().max())
    ops.linear_fprop_split(x_sp, wk, b, True, h, rows, fin, fout, bias_absmax=s[1])
    assert rel_l2(ops.merge_sp(h), ref.clamp_min(0)) < SPLIT_VS_F64
    dx = torch.full((rows, fin), float("nan"), device=dev())
    ops.linear_dgrad_split(g_sp, wt, dx, rows, fin, fout, out_absmax=s[7])
    assert rel_l2(dx, g.double() @ w.double()) < SPLIT_VS_F64 and float(s[7]) == float(dx.abs().max())


@pytest.mark.parametrize("mag", MAGS)
def test_builders_write_scaled_sp_operands_and_their_backward(mag):
    """mvg_fuse_build_split / mvg_fuse_unbuild against their definition (rot_mv.py:44-50,234-239,249-254)."""
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd.heads import directed_pairs
    torch.manual_seed(4)
    V, B, cf, nv = 3, 5, 64, 16
    vi, vj = directed_pairs(V)
    D = len(vi)
    rows = D * B
    img = torch.randn(V * B, cf, device=dev()).abs() * mag
    F_ = torch.randn(rows, 3 * nv, device=dev()) * (mag * 7)
    q, _ = torch.linalg.qr(torch.randn(rows, 3, 3, dtype=torch.float64))
    rel = q.float().to(dev()).contiguous()
    mk = lambda idx: (torch.tensor(idx, dtype=torch.int32)[:, None] * B + torch.arange(B, dtype=torch.int32)[None, :]).reshape(-1).contiguous().to(dev())
    row_img, partner, ident = mk(vi), mk([d ^ 1 for d in range(D)]), mk(list(range(D)))
    st, s = _slots()
    ops.absmax_multi([img, F_], [s[0], s[1]])
    xf, xh = _sp(rows, cf + 3 * nv, s[2]), _sp(rows, cf + 3 * nv, s[3])
    ops.fuse_build_split(img, F_, rel, row_img, partner, ident, xf, xh, s[0], s[1], rows, cf, nv)
    Fd = F_.double().view(rows, 3, nv)
    rot = torch.einsum("mij,mjk->mik", rel.double(), Fd[partner.long()]).reshape(rows, -1)
    want_f = torch.cat([img.double()[row_img.long()], rot], 1)
    want_h = torch.cat([img.double()[row_img.long()], F_.double()[ident.long()]], 1)
    for got, want, sv in ((xf, want_f, s[2]), (xh, want_h, s[3])):
        back = ops.merge_sp(got)
        assert torch.isfinite(back).all()
        assert float(want.abs().max()) / float(sv) < 2.0 ** 15
        # the sp format keeps 2^-23 relative for values within 2^-17 of the tensor's bound, 2^-40 of the bound below that
        err = (back.double() - want).abs()
        # (+ the fp32 rounding of the 3-term rotation itself)
        assert bool((err <= want.abs() * 2.0 ** -22 + float(sv) * 2.0 ** -24 + 6e-7 * float(F_.abs().max())).all())
    # backward: dF = dxh[cf:] + rel^T dxn[partner rows][cf:], da = segment sums of the image parts
    dxh = torch.randn(rows, cf + 3 * nv, device=dev()) / mag
    dxn = torch.randn(rows, cf + 3 * nv, device=dev()) / mag
    ixp = torch.tensor([d ^ 1 for d in range(D)], dtype=torch.int32, device=dev())
    ixv = torch.tensor(vi, dtype=torch.int32, device=dev())
    dF = torch.empty(rows, 3 * nv, device=dev())
    da = torch.empty(V, B, cf, device=dev())
    ops.fuse_unbuild(dxh, dxn, rel, ixp, ixv, dF, da, False, D, V, D, B, cf, nv, absmax=s[4])
    back = torch.einsum("mji,mjk->mik", rel.double(), dxn.double()[:, cf:].view(rows, 3, nv)).reshape(rows, -1)     # rel^T @ g, row m
    want_dF = dxh.double()[:, cf:].clone()
    want_dF[partner.long()] += back                                     # row m of dxn feeds F row partner(m)
    assert rel_l2(dF, want_dF) < 1e-6
    assert float(s[4]) == float(dF.abs().max())
    want_da = torch.zeros(V, B, cf, dtype=torch.float64, device=dev())
    for d in range(D):
        want_da[vi[d]] += (dxh.double() + dxn.double())[d * B:(d + 1) * B, :cf]
    assert rel_l2(da, want_da) < 1e-6
    # iteration 0: the lifted features' gradient = sums over the directions that read view v
    ixj = torch.tensor(vj, dtype=torch.int32, device=dev())
    dl = torch.empty(V * B, 3 * nv, device=dev())
    ops.fuse_unbuild(None, dxn, rel, ixj, ixv, dl, da, True, V, V, D, B, cf, nv)
    want_dl = torch.zeros(V, B, 3 * nv, dtype=torch.float64, device=dev())
    for d in range(D):
        want_dl[vj[d]] += back[d * B:(d + 1) * B]
    assert rel_l2(dl, want_dl.view(V * B, -1)) < 1e-6
    want_da2 = want_da.clone()
    for d in range(D):
        want_da2[vi[d]] += dxn.double()[d * B:(d + 1) * B, :cf]
    assert rel_l2(da, want_da2) < 1e-6


@pytest.mark.parametrize("depth,V,B,scale", [(18, 4, 96, 1.0), (18, 4, 96, 300.0), (18, 4, 96, 1.0 / 300.0),
                                             (50, 4, 128, 1.0)],      # the last one: the fusion block of benchmark configuration C3 at FULL size
                         ids=["r18_x1", "r18_x300", "r18_div300", "c3_full_size"])
def test_fusion_block_with_scaled_fuser_weights_against_the_fp64_oracle(depth, V, B, scale):
    """The fusion head alone on the split path (D * B >= 1024 rows) with every `_img_fusers.*` weight and bias scaled
    x300 / x(1/300): features reach ~1e13 resp. ~5e-5 - far outside what UNSCALED fp16 pieces hold (65504; 6e-8 .. with
    an absolute error floor of 2^-25) - and the features, predictions and every gradient (for a random upstream gradient
    on the predictions: no trigonometry of 1e13-radian angles involved) still follow the fp64 oracle of the same block
    (oracle/restatement.py: lift / fuse_pair = rot_mv.py:91-98,205-263)."""
    from oracle import restatement as R
    from rot_mvgaze_amd import synth
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.model import MultiViewGaze
    # D * B = 1152 rows (ResNet-18 widths) / 1536 rows of 3584 inputs (C3: ResNet-50, V = 4, B = 128 - the workload of the headline
    # number; round 3 checked its fusion-block gradients at 1e-2 against an oracle that ran on its OWN pooled features)
    torch.set_num_threads(min(16, __import__("os").cpu_count() or 1))
    sd = {k: np.array(v) for k, v in synth.make_state_dict(depth, 0, 3).items()}
    for k in sd:
        if k.startswith("_img_fusers."):
            sd[k] = (sd[k] * np.float32(scale)).astype(np.float32)
    m = MultiViewGaze(depth, 3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.to(dev()).train()
    m.ensure_layout()
    torch.manual_seed(5)
    cf = m._fc_dim
    img_feat = (torch.rand(V, B, cf, device=dev()) * 2.0).contiguous()
    rot = rotation_matrix_2d(torch.rand(B * V, 2, device=dev()) - 0.5).reshape(B, V, 3, 3)
    head = m._head
    head.split, head.mixed = True, False
    lifted, feats, preds, tape = head.forward(img_feat, rot, True, True)
    assert tape["mode"] == "split"
    assert torch.isfinite(feats).all() and torch.isfinite(preds).all()
    if scale > 1:
        assert float(feats.abs().max()) > 1e9        # (unscaled fp16 pieces: inf at 65504)
    if scale < 1:
        assert float(feats.abs().max()) < 1e-3
    gpred = torch.randn_like(preds)
    m._sink.active = False
    m._sink.begin()
    dimg = head.backward(tape, None, None, gpred, m._sink, None)
    m._sink.active = False
    torch.cuda.synchronize()

    # fp64 oracle on the same features
    sd64 = {k: torch.from_numpy(v).double() for k, v in sd.items() if v.dtype == np.float32 and not k.startswith("_feat_extractor")}
    for v in sd64.values():
        v.requires_grad_(True)
    f64 = [img_feat[v].double().cpu().requires_grad_(True) for v in range(V)]
    r64 = rot.double().cpu()
    # the ReLU patterns of the HIP forward are IMPOSED on the oracle (oracle/restatement.py:_relu): a hidden unit within
    # fp32 rounding of 0 may fall on either side, which moves gradients discontinuously and says nothing about the
    # kernels (first run of this test without it: every output at 2e-5, one weight gradient at 7e-4 from such flips)
    from rot_mvgaze_amd import ops
    hl_mask = (tape["hl"][0] > 0).cpu().view(V, B, -1)
    fuse_mask = [(ops.merge_sp(tape["saved"][it][1]) > 0).cpu().view(-1, B, tape["saved"][it][1].shape[-3] * 8) for it in range(3)]
    head_mask = [(tape["saved"][it][3] > 0).cpu().view(-1, B, tape["saved"][it][3].shape[-1]) for it in range(3)]
    lift64 = [R.lift(sd64, f64[v], hl_mask[v]) for v in range(V)]
    total = 0.0
    p = 0
    for (i, j) in R.view_pairs(V):
        masks = {}
        for it in range(3):
            masks[("fuse", it)] = [fuse_mask[it][2 * p], fuse_mask[it][2 * p + 1]]
            masks[("head", it)] = [head_mask[it][2 * p], head_mask[it][2 * p + 1]]
        o = R.fuse_pair(sd64, 3, f64[i], f64[j], lift64[i], lift64[j], r64[:, i], r64[:, j], masks)
        for it in range(3):
            for side in (0, 1):
                d = 2 * p + side
                assert rel_l2(feats[it, d].cpu().reshape(B, -1), o[f"iter_{it}"][f"feat_{side}"].detach().reshape(B, -1)) < 2e-5, (it, d)
                assert rel_l2(preds[it, d].cpu(), o[f"iter_{it}"][f"pred_gaze_{side}"].detach()) < 2e-5, (it, d)
                total = total + (o[f"iter_{it}"][f"pred_gaze_{side}"] * gpred[it, d].double().cpu()).sum()
        p += 1
    total.backward()
    for v in range(V):
        assert rel_l2(dimg[v].cpu(), f64[v].grad) < 2e-5, v
    params = dict(m.named_parameters())
    for k, ref in sd64.items():
        if ref.grad is None:
            continue
        got = m._grad_views[id(params[k])]
        assert rel_l2(got.cpu(), ref.grad) < 2e-5, k


@pytest.mark.parametrize("kw,V,B", [({"ignore_rotmat": True}, 3, 176), ({"share_weights": True}, 8, 20), ({}, 2, 512),
                                    ({"share_weights": True, "ignore_rotmat": True}, 5, 52)],
                         ids=["ignore_rotmat_v3", "share_weights_v8", "default_v2_b512", "shared_unrotated_v5"])
def test_split_fusion_path_covers_the_variants_and_view_counts(kw, V, B):
    """The split path also serves ignore_rotmat (rot_mv.py:226-232: the partner feature is NOT rotated) and share_weights
    (:148-156: one fuser / head for every iteration - their gradients accumulate over the iterations), at any V with
    V (V - 1) B >= 1024 rows.  Same statement as above: fp64 oracle with the HIP forward's ReLU pattern, 2e-5."""
    from oracle import restatement as R
    from rot_mvgaze_amd import ops, synth
    from rot_mvgaze_amd.arch import Variant
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.model import MultiViewGaze
    depth = 18
    v = Variant(**kw)
    torch.set_num_threads(min(16, __import__("os").cpu_count() or 1))
    sd = {k: np.array(x) for k, x in synth.make_state_dict(depth, 0, 3, variant=v).items()}
    m = MultiViewGaze(depth, 3, v)
    m.load_state_dict({k: torch.from_numpy(x) for k, x in sd.items()}, strict=True)
    m.to(dev()).train()
    m.ensure_layout()
    torch.manual_seed(6)
    cf = m._fc_dim
    img_feat = (torch.rand(V, B, cf, device=dev()) * 2.0).contiguous()
    rot = rotation_matrix_2d(torch.rand(B * V, 2, device=dev()) - 0.5).reshape(B, V, 3, 3)
    head = m._head
    head.split, head.mixed = True, False
    lifted, feats, preds, tape = head.forward(img_feat, rot, True, True)
    assert tape["mode"] == "split" and V * (V - 1) * B >= 1024
    gpred = torch.randn_like(preds)
    m._sink.active = False
    m._sink.begin()
    dimg = head.backward(tape, None, None, gpred, m._sink, None)
    m._sink.active = False
    torch.cuda.synchronize()
    sd64 = {k: torch.from_numpy(x).double() for k, x in sd.items() if x.dtype == np.float32 and not k.startswith("_feat_extractor")}
    for t in sd64.values():
        t.requires_grad_(True)
    f64 = [img_feat[i].double().cpu().requires_grad_(True) for i in range(V)]
    r64 = rot.double().cpu()
    hl_mask = (tape["hl"][0] > 0).cpu().view(V, B, -1)
    fuse_mask = [(ops.merge_sp(tape["saved"][it][1]) > 0).cpu().view(-1, B, tape["saved"][it][1].shape[-3] * 8) for it in range(3)]
    head_mask = [(tape["saved"][it][3] > 0).cpu().view(-1, B, tape["saved"][it][3].shape[-1]) for it in range(3)]
    lift64 = [R.lift(sd64, f64[i], hl_mask[i]) for i in range(V)]
    total, p = 0.0, 0
    for (i, j) in R.view_pairs(V):
        masks = {}
        for it in range(3):
            masks[("fuse", it)] = [fuse_mask[it][2 * p], fuse_mask[it][2 * p + 1]]
            masks[("head", it)] = [head_mask[it][2 * p], head_mask[it][2 * p + 1]]
        o = R.fuse_pair(sd64, 3, f64[i], f64[j], lift64[i], lift64[j], r64[:, i], r64[:, j], masks, v)
        for it in range(3):
            for side in (0, 1):
                d = 2 * p + side
                assert rel_l2(feats[it, d].cpu().reshape(B, -1), o[f"iter_{it}"][f"feat_{side}"].detach().reshape(B, -1)) < 2e-5, (it, d)
                assert rel_l2(preds[it, d].cpu(), o[f"iter_{it}"][f"pred_gaze_{side}"].detach()) < 2e-5, (it, d)
                total = total + (o[f"iter_{it}"][f"pred_gaze_{side}"] * gpred[it, d].double().cpu()).sum()
        p += 1
    total.backward()
    for i in range(V):
        assert rel_l2(dimg[i].cpu(), f64[i].grad) < 2e-5, i
    # share_weights: one device Parameter under several state_dict names - the oracle's gradient is the sum over the names
    groups = {}
    for k, prm in m.named_parameters(remove_duplicate=False):
        if not k.startswith("_feat_extractor"):
            groups.setdefault(id(prm), (prm, []))[1].append(k)
    for prm, names in groups.values():
        refs = [sd64[k].grad for k in names if sd64[k].grad is not None]
        if not refs:
            continue
        assert rel_l2(m._grad_views[id(prm)].cpu(), sum(refs)) < 2e-5, names
