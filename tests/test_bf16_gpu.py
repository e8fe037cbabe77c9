"""The bf16 storage path (BASELINE.json configs[4]: "bf16 MFMA path"; SURVEY.md 7 hard part 3).

Nothing in the reference runs in bf16, so this path has no reference fixture: its parity statement is
(a) kernel level - every bf16 kernel against float64 arithmetic on the SAME bf16-rounded operands
(products of bf16 values are exact in fp32 and the accumulation is fp32, so only the output rounding
to bf16, 2^-9 relative, separates the two), and (b) model level - the whole training step against the
fp32 CPU oracle at a tolerance DECLARED HERE, before measuring:

    BF16_PRED_TOL = 3e-2   max |pred_gaze - oracle| relative to max |oracle|  (gaze angles, radians)
    BF16_LOSS_TOL = 3e-2   relative error of the loss
    BF16_GRAD_L2  = 1e-1   relative L2 error of sampled weight gradients

(bf16 carries 8 significant bits: every one of the ~50 conv/BN layers re-rounds its activations to
2^-9 relative, BatchNorm renormalises the error instead of letting it grow, so the end-to-end error is
a small multiple of 2^-9 = 2e-3; gradients additionally see the rounded activations' ReLU flips.)
The fp32 path's bar (1e-4) is NOT claimed for this path.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import rot_mvgaze_amd  # noqa: F401
from rot_mvgaze_amd import synth

pytestmark = pytest.mark.gpu

BF16_PRED_TOL = 3e-2
BF16_LOSS_TOL = 3e-2
BF16_GRAD_L2 = 1e-1
OUT_RTOL = 6e-3      # a bf16-rounded output vs fp64: half an ulp (2^-9 = 2e-3) at the largest magnitude, with margin


def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, tag="t", scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((synth.normal(n, seed, tag) * scale).astype(np.float32).reshape(shape))


def bf(x):
    return x.to(torch.bfloat16)


def close(got, ref, rtol, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-30
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e})"


def to_nhwc(x):     # [G,N,C,H,W] -> [G,N,H,W,C]
    return x.permute(0, 1, 3, 4, 2).contiguous()


BF16_CONV_CASES = [
    # G, N, H, W, Cin, Cout, k, stride, pad
    (2, 3, 14, 14, 64, 128, 3, 1, 1),
    (2, 3, 15, 13, 64, 128, 3, 2, 1),     # odd sizes, stride 2: ragged parity classes in dgrad
    (1, 5, 14, 14, 128, 64, 1, 1, 0),     # 64 output columns -> the 128x64 tile
    (2, 2, 14, 14, 64, 256, 1, 2, 0),     # 1x1 stride 2: three of four dgrad classes have no tap
    (2, 2, 36, 36, 8, 64, 7, 2, 3),       # stem: 3 channels padded to 8, K = 392 (per-lane tap decode)
    (2, 8, 56, 56, 64, 256, 1, 1, 0),
    (1, 16, 28, 28, 128, 128, 3, 1, 1),
    (2, 2, 7, 7, 512, 512, 3, 1, 1),      # 7x7 maps: wgrad without incremental pixel stepping
    (2, 4, 16, 16, 64, 128, 3, 2, 1),
    (1, 2, 28, 28, 256, 512, 1, 2, 0),
    (1, 30, 14, 14, 256, 256, 3, 1, 1),   # ragged last M tile
    (1, 3, 9, 9, 32, 32, 3, 1, 1),        # 32 channels per tap (not a multiple of the 64-deep K-step)
    (2, 16, 56, 56, 64, 64, 3, 1, 1),     # large: many tiles, several wgrad splits
]


@pytest.mark.parametrize("case", BF16_CONV_CASES)
def test_bf16_conv_fprop_dgrad_wgrad(case):
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc
    G, N, H, W, Cin, Cout, k, st, pad = case
    x = bf(rnd((G, N, Cin, H, W), 1, "x")).float()
    stem = Cin == 8 and k == 7
    if stem:
        x[:, :, 3:] = 0
    cin_src = 3 if stem else Cin
    w = rnd((Cout, cin_src, k, k), 2, "w", 1.0 / np.sqrt(cin_src * k * k))
    wb = bf(w).float()                                       # what the cast kernel must produce
    d = ConvDesc.make(G, N, H, W, Cin, Cout, k, st, pad)
    xr = x.reshape(G * N, Cin, H, W)[:, :cin_src].double().requires_grad_(True)
    wr = wb.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, st, pad)
    gy = bf(rnd(tuple(yr.shape), 3, "gy")).float()
    yr.backward(gy.double())

    xd = bf(to_nhwc(x)).to(dev())
    w_krsc = w.permute(0, 2, 3, 1).contiguous().to(dev())    # fp32 master weights, KRSC (cin_src channels)
    wk, wt = ops.cast_weights_bf16(d, w_krsc, cin_src, True)
    assert torch.equal(wk[..., :cin_src].float().cpu(), wb.permute(0, 2, 3, 1)), "cast: KRSC copy"
    assert torch.equal(wt.float().cpu(), wk.float().cpu().permute(3, 1, 2, 0)), "cast: transposed copy"
    if cin_src < Cin:
        assert float(wk[..., cin_src:].float().abs().max()) == 0.0

    y = torch.empty(G, N, d.ho, d.wo, Cout, dtype=torch.bfloat16, device=dev())
    P, rpp = ops.conv_stats_partials(d, True)
    stats = torch.full((G, P, 2, Cout), float("nan"), device=dev())
    ops.conv_fprop(d, xd, wk, y, None, False, stats)
    y_ref = yr.detach().reshape(G, N, Cout, d.ho, d.wo).permute(0, 1, 3, 4, 2)
    close(y, y_ref, OUT_RTOL, "fprop")
    # BN partial statistics come from the fp32 accumulators (before the bf16 rounding of y)
    rows = N * d.ho * d.wo
    close(stats[:, :, 0].sum(1), y_ref.reshape(G, rows, Cout).sum(1), 2e-4, "fprop stats: column sums")
    mean, invstd, scale, shift = (torch.empty(G, Cout, device=dev()) for _ in range(4))
    ops.bn_finalize(stats, G, P, rpp, rows, Cout, torch.ones(Cout, device=dev()), torch.zeros(Cout, device=dev()),
                    torch.zeros(Cout, device=dev()), torch.ones(Cout, device=dev()), 0.1, 1e-5, mean, invstd, scale, shift)
    yg = y_ref.reshape(G, rows, Cout)
    close(mean, yg.mean(1), 1e-4, "bn mean")
    close(invstd, 1.0 / torch.sqrt(yg.var(1, unbiased=False) + 1e-5), 1e-4, "bn invstd")

    gyd = bf(to_nhwc(gy.reshape(G, N, Cout, d.ho, d.wo))).to(dev())
    if not stem:
        dx = torch.empty(G, N, H, W, Cin, dtype=torch.bfloat16, device=dev())
        ops.conv_dgrad(d, gyd, wt, dx)
        dx_ref = xr.grad.reshape(G, N, Cin, H, W).permute(0, 1, 3, 4, 2)
        close(dx, dx_ref, OUT_RTOL, "dgrad")
        add = bf(rnd((G, N, H, W, Cin), 7, "a")).to(dev())
        dx2 = add.clone()
        ops.conv_dgrad(d, gyd, wt, dx2, None, dx2)           # in-place addend (residual fan-in)
        close(dx2, dx_ref + add.float().cpu().double(), OUT_RTOL, "dgrad + addend")

    dw = torch.empty(Cout, k, k, Cin, device=dev())
    ops.conv_wgrad(d, xd, gyd, dw, False)
    dw_ref = wr.grad.permute(0, 2, 3, 1)
    close(dw[..., :cin_src], dw_ref, 2e-5, "wgrad")           # fp32 output: exact products, fp32 accumulation
    dw2 = dw.clone()
    ops.conv_wgrad(d, xd, gyd, dw2, True)
    close(dw2[..., :cin_src], 2 * dw_ref, 2e-5, "wgrad accumulate")


@pytest.mark.parametrize("G,rows,C,residual", [(2, 3000, 64, False), (3, 777, 256, True), (1, 50, 2048, True)])
def test_bf16_batchnorm_passes(G, rows, C, residual):
    """bn_apply / bn_bwd_reduce / bn_bwd_apply with bf16 activations == the fp32 kernels fed with the same
    (bf16-representable) values, up to the rounding of the bf16 outputs."""
    from rot_mvgaze_amd import ops
    y = bf(rnd((G, rows, C), 1, "y")).to(dev())
    g = bf(rnd((G, rows, C), 2, "g")).to(dev())
    res = bf(rnd((G, rows, C), 3, "r")).to(dev()) if residual else None
    scale = (rnd((G, C), 4, "s") * 0.1 + 1).to(dev())
    shift = (rnd((G, C), 5, "h") * 0.1).to(dev())
    mean, invstd = (rnd((G, C), 6, "m") * 0.1).to(dev()), (rnd((G, C), 7, "i") * 0.1 + 1).to(dev())
    gamma = (rnd((C,), 8, "ga") * 0.1 + 1).to(dev())
    out_b, out_f = torch.empty_like(y), torch.empty(G, rows, C, device=dev())
    ops.bn_apply(y, scale, shift, res, True, out_b, G, rows, C)
    ops.bn_apply(y.float(), scale, shift, res.float() if residual else None, True, out_f, G, rows, C)
    assert torch.equal(out_b, out_f.to(torch.bfloat16)), "bn_apply: bf16 output must be the rounded fp32 output"
    act_b = out_b if residual else None
    ra = None if residual else (scale, shift)
    outs = []
    for conv in (lambda t: t, lambda t: t.float()):
        s12 = torch.empty(2, G, C, device=dev())
        dg, db = torch.empty(C, device=dev()), torch.empty(C, device=dev())
        gg, yy, aa = conv(g), conv(y), (conv(act_b) if act_b is not None else None)
        ops.bn_bwd_reduce(gg, aa, yy, mean, invstd, G, rows, C, s12[0], s12[1], dg, db, False, ra)
        dy, dz = torch.empty_like(gg), torch.empty_like(gg)
        ops.bn_bwd_apply(gg, aa, yy, mean, invstd, gamma, s12[0], s12[1], G, rows, C, dy, dz, ra)
        outs.append((s12, dg, db, dy, dz))
    (s_b, dg_b, db_b, dy_b, dz_b), (s_f, dg_f, db_f, dy_f, dz_f) = outs
    close(s_b, s_f, 1e-6, "bn_bwd sums")
    close(dg_b, dg_f, 1e-6, "dgamma")
    close(db_b, db_f, 1e-6, "dbeta")
    assert torch.equal(dz_b, dz_f.to(torch.bfloat16))
    close(dy_b, dy_f, OUT_RTOL, "bn_bwd_apply dy")


def test_bf16_stem_tail_and_pools():
    from rot_mvgaze_amd import ops
    G, N, H, W, C = 2, 3, 18, 20, 64
    y = bf(rnd((G, N, H, W, C), 1, "y")).to(dev())
    scale = (rnd((G, C), 4, "s") * 0.1 + 1).to(dev())
    shift = (rnd((G, C), 5, "h") * 0.1).to(dev())
    hp, wp = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    outs = []
    for conv, dt in ((lambda t: t, torch.bfloat16), (lambda t: t.float(), torch.float32)):
        pooled = torch.empty(G, N, hp, wp, C, dtype=dt, device=dev())
        am = torch.empty(G, N, hp, wp, C, dtype=torch.uint8, device=dev())
        ops.bn_relu_maxpool_fwd(conv(y), scale, shift, pooled, am, G, N, H, W, C, hp, wp)
        outs.append((pooled, am))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0].to(torch.bfloat16))
    gp = bf(rnd((G, N, hp, wp, C), 2, "g")).to(dev())
    mean, invstd = (rnd((G, C), 6, "m") * 0.1).to(dev()), (rnd((G, C), 7, "i") * 0.1 + 1).to(dev())
    gamma = (rnd((C,), 8, "ga") * 0.1 + 1).to(dev())
    res = []
    for conv in (lambda t: t, lambda t: t.float()):
        s12 = torch.empty(2, G, C, device=dev())
        dg, db = torch.empty(C, device=dev()), torch.empty(C, device=dev())
        ops.bn_relu_maxpool_bwd_reduce(conv(gp), outs[0][1], conv(y), mean, invstd, scale, shift, G, N, H, W, C, hp, wp, s12[0],
                                       s12[1], dg, db, False)
        dy = torch.empty_like(conv(y))
        ops.bn_relu_maxpool_bwd_apply(conv(gp), outs[0][1], conv(y), mean, invstd, gamma, scale, shift, s12[0], s12[1], G, N,
                                      H, W, C, hp, wp, dy)
        res.append((s12, dy))
    close(res[0][0], res[1][0], 1e-6, "stem bwd sums")
    close(res[0][1], res[1][1], OUT_RTOL, "stem bwd dy")
    # average pool: bf16 map -> fp32 features; fp32 feature gradient -> bf16 map
    x = bf(rnd((6, 49, 256), 9, "x")).to(dev())
    f = torch.empty(6, 256, device=dev())
    ops.avgpool_fwd(x, f, 6, 49, 256)
    close(f, x.float().mean(1), 1e-6, "avgpool fwd")
    dx = torch.empty_like(x)
    ops.avgpool_bwd(f, dx, 6, 49, 256)
    close(dx, (f / 49)[:, None, :].expand(6, 49, 256), OUT_RTOL, "avgpool bwd")
    img = rnd((4, 3, 10, 12), 10, "img").to(dev())
    o = torch.empty(4, 10, 12, 8, dtype=torch.bfloat16, device=dev())
    ops.nchw_to_nhwc8_bf16(img, o, 4, 3, 10, 12)
    assert torch.equal(o[..., :3], img.permute(0, 2, 3, 1).to(torch.bfloat16)) and float(o[..., 3:].float().abs().max()) == 0


BF16_MODEL_CASES = [(18, 2, 8, 96), (50, 2, 4, 128), (50, 4, 4, 64), (50, 8, 2, 224)]


@pytest.mark.parametrize("depth,V,B,hw", BF16_MODEL_CASES, ids=[f"r{d}_V{v}_B{b}_hw{h}" for d, v, b, h in BF16_MODEL_CASES])
def test_bf16_training_step_against_fp32_oracle(depth, V, B, hw):
    """One training step with compute_dtype = bfloat16 (C5's shapes at reduced batch: ResNet-50, V = 8, 224 px)
    against the fp32 CPU oracle at the tolerances declared at the top of this file."""
    from oracle import restatement as R
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.losses import MultiViewIterationLoss
    from rot_mvgaze_amd.model import MultiViewGaze
    m = MultiViewGaze(depth, 3)
    sdn = synth.make_state_dict(depth, 0, 3, perturb_bn=True)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdn.items()})
    m.to(dev()).train()
    m.compute_dtype = torch.bfloat16
    inp = synth.make_inputs(B, V, 5, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    rot_d = rotation_matrix_2d(hp.reshape(-1, 2).to(dev())).reshape(B, V, 3, 3)
    out = m.forward_multiview([img[:, v].contiguous().to(dev()) for v in range(V)], rot_d)
    loss = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)(out, gt.to(dev()))
    loss.backward()
    assert out["img_feat"].dtype == torch.float32                       # the fusion block stays fp32
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and "running" not in k}
    rot = R.rotation_matrix_2d(hp.reshape(-1, 2)).reshape(B, V, 3, 3)
    oo = R.multiview_forward(sd, img, rot, depth, 3, True)
    ol = R.multiview_loss(oo, gt, iter_decay=0.5, rel_weight=0.01, reference_decay=1.0)
    ol.backward()
    assert abs(loss.item() - ol.item()) <= BF16_LOSS_TOL * abs(ol.item()), (loss.item(), ol.item())
    for pr in R.view_pairs(V):
        for it in range(3):
            for k in ("pred_gaze_0", "pred_gaze_1"):
                close(out["pairs"][pr][f"iter_{it}"][k], oo["pairs"][pr][f"iter_{it}"][k], BF16_PRED_TOL, f"pair {pr} iter {it} {k}")
    params = dict(m.named_parameters())
    worst = 0.0
    for k in ("_lifter._lifter.blocks.0.0.weight", "_img_fusers.0._fuser.blocks.0.0.weight", "_gaze_estimators.1.blocks.1.0.weight",
              "_feat_extractor.0.layer4.0.conv1.weight", "_feat_extractor.0.layer2.0.conv2.weight", "_feat_extractor.0.conv1.weight",
              "_feat_extractor.0.bn1.bias"):
        got, ref = params[k].grad.detach().cpu().double().numpy(), leaves[k].grad.double().numpy()
        err = np.linalg.norm((got - ref).ravel()) / (np.linalg.norm(ref.ravel()) + 1e-30)
        worst = max(worst, err)
        assert err <= BF16_GRAD_L2, f"grad {k}: relative L2 {err:.3e}"
    # BN running statistics (fp32, from the fp32 accumulators)
    close(m.state_dict()["_feat_extractor.0.bn1.running_mean"], sd["_feat_extractor.0.bn1.running_mean"], 1e-2, "running_mean")
    assert int(m.state_dict()["_feat_extractor.0.bn1.num_batches_tracked"]) == V


def test_bf16_path_leaves_fp32_path_untouched():
    """Switching compute_dtype back restores the fp32 results bit for bit (same model object)."""
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.model import MultiViewGaze
    m = MultiViewGaze(18, 3)
    sdn = synth.make_state_dict(18, 0, 3, perturb_bn=True)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdn.items()})
    m.to(dev()).eval()
    inp = synth.make_inputs(3, 2, 5, 64)
    img, hp = torch.from_numpy(inp["img"]).to(dev()), torch.from_numpy(inp["head_pose"])
    rot_d = rotation_matrix_2d(hp.reshape(-1, 2).to(dev())).reshape(3, 2, 3, 3)
    with torch.no_grad():
        a = m.forward_multiview(img, rot_d)["pred_gaze"].clone()
        m.compute_dtype = torch.bfloat16
        b = m.forward_multiview(img, rot_d)["pred_gaze"].clone()      # bf16 inference (BN from running statistics)
        m.compute_dtype = torch.float32
        c = m.forward_multiview(img, rot_d)["pred_gaze"].clone()
    assert torch.equal(a, c)
    close(b, a, BF16_PRED_TOL, "bf16 eval vs fp32 eval")
