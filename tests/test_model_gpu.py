"""End-to-end parity of the MI355X FeatRotationSymm (forward, loss, backward, BN running stats)
against (a) the golden fixtures produced by the reference's own Python and (b) the CPU oracle on
the same seeded inputs.  Tolerance: the north star's 1e-4 relative for gaze vectors and loss
(fp32); gradients 1e-3 relative to the tensor's max (they pass through ~50 layers of fp32
reductions in a different order)."""
import os

import numpy as np
import pytest
import torch

import rot_mvgaze_amd  # noqa: F401
from rot_mvgaze_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4       # gaze vectors and loss (the north star's bar)
FTOL = 3e-4      # intermediate feature tensors: the fixtures' 2-3 sample batches leave BatchNorm in
                 # layer4 with 8-12 samples per channel (2x2 maps at 64x64 input), which amplifies
                 # fp32 reduction-order noise ~10x; predictions and loss still meet 1e-4
GTOL = 2e-4      # gradients, same ReLU activation pattern on both sides (strict test below)
# Against the reference's fp32 fixtures the activation pattern itself differs on a handful of
# elements whose pre-activation is within fp32 rounding of 0 (forward values agree to ~1e-6
# relative, but a flipped ReLU decision moves downstream gradients discontinuously, and at the tiny
# fixture batches - 2..3 samples - one element is up to 1/sqrt(rows) of a weight gradient).  Those
# checks therefore bound the relative L2 error instead of the max error.
# Bound: 1e-2 (round 1 held 5e-2).  What sets it: the full-size runs measure 1e-6 (fusion block) .. 3e-3
# (layer4) .. 6e-3 (stem) at C2; the 2-3-sample reference fixtures (model_r50_b3_hw64, variant_*) are the
# worst cases because one flipped element is a larger share of a weight gradient there.
GTOL_L2_FLIPS = 1e-2
# ResNet-50 and tiny BatchNorm populations, measured on MI355X (MVG_TEST_L2_LOG): 1.2e-2 .. 2.1e-2 on the stem /
# layer3 conv weights (r50 V=4 B=3..4, V=8 B=2, and 2.0e-2 on the stem at C4's full per-GPU size: ~600 of
# 1.4 G activations flip and the stem sees all of them through 50 layers); ResNet-18 at 40x40 9.8e-3.  That
# these are flips and not a systematic error is pinned by the imposed-pattern tests (max-norm 2e-4..4e-4
# against fp64, incl. ResNet-50 x V=4): test_backward_strict_*.
GTOL_L2_FLIPS_DEEP = 3e-2


# ResNet-50 on ONE sample (layer4's BatchNorm sees 9 values per channel at 96 px): the realisation of the flips decides -
# over 8 input seeds the worst gradient's error is 1.6e-2 .. 6.3e-2 on the split kernels and 1.1e-2 .. 4.5e-2 on the
# fp32-MFMA kernels (scripts/flip_spread.py -> profiles/r03_flip_spread_r50_b1_hw96.txt; round 2's 3e-2 held on the one
# seed the test uses); loss and predictions stay at 1e-5.
GTOL_L2_FLIPS_SINGLE = 8e-2


def l2_bound(depth, batch, hw):
    """1e-2 where layer4's BatchNorm sees >= 64 values per channel on ResNet-18; 3e-2 for ResNet-50 and tiny populations;
    8e-2 for ResNet-50 with fewer than 16 values per channel there (a single small sample)."""
    pop = batch * max(hw // 32, 1) ** 2
    if depth == 50 and pop < 16:
        return GTOL_L2_FLIPS_SINGLE
    return GTOL_L2_FLIPS if (depth == 18 and pop >= 64) else GTOL_L2_FLIPS_DEEP
# the reference's own 2-3-sample fixtures: measured <= 1.2e-2 (model_r50_b3_hw64, stem conv: 12-sample
# BatchNorm in layer4, one flipped ReLU there reaches the stem through 50 layers); everything else <= 1e-5
GTOL_L2_FIXTURE = 2e-2          # ResNet-18 fixtures (measured <= 1e-5); the ResNet-50 ones use GTOL_L2_FLIPS_DEEP (2.1e-2 on b2_hw224)
_L2_LOG = os.environ.get("MVG_TEST_L2_LOG")      # optional: append every relative-L2 figure to this file


def dev():
    return torch.device("cuda:0")


def build(depth, seed=0, train=True, conditioned=False):
    from rot_mvgaze_amd.model import FeatRotationSymm
    m = FeatRotationSymm(backbone_depth=depth, num_iter=3)
    sd = synth.make_state_dict(depth, seed, 3, perturb_bn=True, conditioned=conditioned)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    m.to(dev())
    return m.train() if train else m.eval()


def inputs(batch, hw, seed=1234, views=2):
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    inp = synth.make_inputs(batch, views, seed, hw)
    img, hp, gt = (torch.from_numpy(inp[k]).to(dev()) for k in ("img", "head_pose", "gt_gaze"))
    return {"img_0": img[:, 0].contiguous(), "img_1": img[:, 1].contiguous(),
            "rot_0": rotation_matrix_2d(hp[:, 0].contiguous()), "rot_1": rotation_matrix_2d(hp[:, 1].contiguous()),
            "gt_gaze": gt[:, 0].contiguous(), "gt_gaze_1": gt[:, 1].contiguous()}


def metrics():
    from rot_mvgaze_amd.losses import IterationLoss, StereoL1Loss
    return IterationLoss(StereoL1Loss(rel_weight=0.01, reference_decay=1.0, distance_metric="angular_error",
                                      pred_gaze_key="pred_gaze"), iter_decay=0.5)


def rel_close(got, ref, tol, what):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref).max()
    scale = np.abs(ref).max() + 1e-30
    assert err <= tol * scale, f"{what}: max err {err:.3e}, scale {scale:.3e}, rel {err / scale:.3e}"


def l2_close(got, ref, tol, what):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.linalg.norm((got - ref).ravel()) / (np.linalg.norm(ref.ravel()) + 1e-30)
    if _L2_LOG:
        with open(_L2_LOG, "a") as f:
            f.write(f"{err:.3e} {tol:.1e} {os.environ.get('PYTEST_CURRENT_TEST', '?').split('::')[-1]} {what}\n")
    assert err <= tol, f"{what}: relative L2 error {err:.3e} > {tol}"


def check_outputs(data, g, prefix, tol):
    for k in ("img_feat_0", "img_feat_1", "initial_rot_feat_0", "initial_rot_feat_1", "pred_gaze"):
        rel_close(data[k], g[f"{prefix}.{k}"], tol if k == "pred_gaze" else FTOL, f"{prefix}.{k}")
    for i in range(3):
        for k in ("feat_0", "feat_1", "pred_gaze_0", "pred_gaze_1"):
            rel_close(data[f"iter_{i}"][k], g[f"{prefix}.iter_{i}.{k}"], tol if "pred" in k else FTOL,
                      f"{prefix}.iter_{i}.{k}")


# The fp32 model's training convs run on one of two kernel families, both held to the same bounds: "split" (default:
# fp32 values as two fp16 pieces ("sp") with a per-tensor power-of-two scale, three fp16 MFMAs per product, conv_split.hip) and
# "fp32mfma" (MVG_SPLIT=0: the v_mfma_f32_32x32x2_f32 kernels of conv_igemm.hip, which also serve the lifter, small-row Linears
# and the stem's backward-data).
KERNELS = ["split", "fp32mfma"]


def select_kernels(monkeypatch, kernels):
    monkeypatch.setenv("MVG_SPLIT", "1" if kernels == "split" else "0")


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("depth,batch,hw", [(18, 3, 64), (50, 3, 64), (18, 2, 224), (50, 2, 224)])
def test_against_reference_golden(golden_dir, monkeypatch, depth, batch, hw, kernels):
    select_kernels(monkeypatch, kernels)
    g = np.load(os.path.join(golden_dir, f"model_r{depth}_b{batch}_hw{hw}.npz"))
    # ---- eval
    m = build(depth, train=False)
    with torch.no_grad():
        data = m(inputs(batch, hw))
    check_outputs(data, g, "eval", TOL)
    # ---- train step
    m = build(depth, train=True)
    data = inputs(batch, hw)
    data["img_0"].requires_grad_(True)
    extra = {"idx_0": torch.arange(batch), "something": "else"}
    data.update(extra)
    ret = m(data)
    assert ret is data and data["something"] == "else" and data["num_iter"] == 3      # in-place dict protocol
    loss = metrics()(data)
    loss.backward()
    check_outputs(data, g, "train", TOL)
    rel_close(loss, g["train.loss"], TOL, "loss")
    params = dict(m.named_parameters())
    for key in [k[5:] for k in g.files if k.startswith("grad._")]:
        ref = g["grad." + key]
        p = params[key]
        gr = p.grad
        if gr.dim() == 4:
            gr = gr.contiguous()          # logical OIHW order, like the fixture
        l2_close(gr.reshape(-1)[: ref.size], ref, (GTOL_L2_FLIPS_DEEP if depth == 50 else GTOL_L2_FIXTURE), "grad " + key)
        rel_close(gr.double().norm().item(), g["gradnorm." + key], (GTOL_L2_FLIPS_DEEP if depth == 50 else GTOL_L2_FIXTURE), "gradnorm " + key)
    assert params["_feat_extractor.0.fc.weight"].grad is None
    l2_close(data["img_0"].grad[:, :, ::16, ::16], g["grad.img_0"], (GTOL_L2_FLIPS_DEEP if depth == 50 else GTOL_L2_FIXTURE), "grad img_0")
    sd = m.state_dict()
    for k in [k for k in g.files if k.startswith("stat.")]:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k[5:]]) == int(g[k]) == 2
        else:
            rel_close(sd[k[5:]], g[k], TOL, k)


VARIANTS = {
    "share_weights": dict(share_weights=True),
    "ignore_rotmat": dict(ignore_rotmat=True),
    "encode_rotmat": dict(encode_rotmat=True),
    "share_feature": dict(share_feature=True),
    "share_weights_encode_rotmat": dict(share_weights=True, encode_rotmat=True),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variants_against_reference_golden(golden_dir, name):
    """Constructor variants of the reference model (rot_mv.py:136-171) against
    fixtures the reference produced: eval outputs, train outputs, loss, gradients, IntensityBatchNorm buffers."""
    from rot_mvgaze_amd.arch import Variant
    from rot_mvgaze_amd.model import FeatRotationSymm
    kw = VARIANTS[name]
    g = np.load(os.path.join(golden_dir, f"variant_{name}_r18_b3_hw64.npz"))

    def make(train):
        m = FeatRotationSymm(backbone_depth=18, num_iter=3, **kw)
        sd = synth.make_state_dict(18, 0, 3, perturb_bn=True, variant=Variant(**kw))
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
        m.to(dev())
        return m.train() if train else m.eval()
    m = make(False)
    with torch.no_grad():
        data = m(inputs(3, 64))
    check_outputs(data, g, "eval", TOL)
    m = make(True)
    data = m(inputs(3, 64))
    loss = metrics()(data)
    loss.backward()
    check_outputs(data, g, "train", TOL)
    rel_close(loss, g["train.loss"], TOL, "loss")
    params = dict(m.named_parameters(remove_duplicate=False))
    for key in [k[5:] for k in g.files if k.startswith("grad._")]:
        ref = g["grad." + key]
        gr = params[key].grad
        if gr.dim() == 4:
            gr = gr.contiguous()
        l2_close(gr.reshape(-1)[: ref.size], ref, GTOL_L2_FIXTURE, "grad " + key)
        rel_close(gr.double().norm().item(), g["gradnorm." + key], GTOL_L2_FIXTURE, "gradnorm " + key)
    sd = m.state_dict()
    for k in [k for k in g.files if k.startswith("stat.")]:
        rel_close(sd[k[5:]], g[k], TOL, k)
    if kw.get("share_weights"):
        assert params["_img_fusers.2._fuser.blocks.0.0.weight"] is params["_img_fusers.0._fuser.blocks.0.0.weight"]
    torch.optim.Adam(m.parameters(), lr=1e-4).step()


def test_against_oracle_and_generic_loss_path():
    """Same step as the CPU oracle; the generic (per-call) loss path and the fused one agree."""
    from oracle import restatement as R
    from rot_mvgaze_amd.losses import IterationLoss, StereoL1Loss
    depth, batch, hw = 18, 5, 96
    m = build(depth)
    data = inputs(batch, hw, seed=77)
    data = m(data)
    loss = metrics()(data)
    loss.backward()
    g_fused = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    # generic path: hide the fast-path handle
    m.zero_grad(set_to_none=True)
    data2 = m(inputs(batch, hw, seed=77))
    data2.pop("_mvg_preds")
    loss2 = metrics()(data2)
    loss2.backward()
    rel_close(loss2, loss.item(), 1e-6, "generic vs fused loss")
    for k, p in m.named_parameters():
        if p.grad is not None:
            rel_close(p.grad, g_fused[k].cpu().numpy(), 1e-5, "generic vs fused grad " + k)
    # oracle, free-running (its own ReLU decisions): forward quantities
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_state_dict(depth, 0, 3, perturb_bn=True).items()}
    inp = synth.make_inputs(batch, 2, 77, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    od = {"img_0": img[:, 0], "img_1": img[:, 1], "rot_0": R.rotation_matrix_2d(hp[:, 0]),
          "rot_1": R.rotation_matrix_2d(hp[:, 1]), "gt_gaze": gt[:, 0], "gt_gaze_1": gt[:, 1]}
    with torch.no_grad():
        od = R.model_forward(sd, od, depth, 3, True)
        ol = R.iteration_loss(od)
    rel_close(loss, ol.item(), TOL, "loss vs oracle")
    for i in range(3):
        rel_close(data[f"iter_{i}"]["pred_gaze_1"], od[f"iter_{i}"]["pred_gaze_1"].numpy(), TOL, "pred vs oracle")


def _captured_masks(m, V=2):
    """ReLU patterns of the HIP forward, in the order the oracle applies its ReLUs."""
    bt, ht = m._last_backbone_tape, m._last_head_tape
    relu_units = [u for u in bt["units"] if u.relu]

    def unit_mask(u, v):
        # the mask the BACKWARD kernels use - not "stored activation > 0": the split path stores activations as two fp16
        # pieces, which flush a positive value below fp16's smallest subnormal (6e-8) to zero while its mask stays on
        if getattr(u, "relu_bits", None) is not None:       # residual units: one byte per 4 channels, bit k = channel k on
            G = u.y.shape[0]
            bits = u.relu_bits.view(G, -1)[v]
            on = ((bits[:, None] >> torch.arange(4, device=bits.device, dtype=torch.uint8)[None, :]) & 1).bool()
            return on.reshape(u.y.shape[1:])
        if getattr(u, "relu_affine", None) is not None and u.out is not None and u.out.dtype == torch.float16:
            scale, shift = u.relu_affine            # fma(y, scale, shift) > 0: the sign of the exact value (fp64 here)
            return (u.y[v].double() * scale[v].double() + shift[v].double()) > 0
        if u.out is not None:
            return u.out[v].contiguous() > 0
        # fused stem: the normalised map is not stored; the kernels use fma(y, scale, shift) > 0, whose
        # sign equals the sign of the exact value (evaluated here in fp64)
        scale, shift = u.pool[1], u.pool[2]
        return (u.y[v].double() * scale[v].double() + shift[v].double()) > 0
    masks = {"backbone": [iter([unit_mask(u, v).permute(0, 3, 1, 2).cpu() for u in relu_units]) for v in range(V)]}
    B = bt["B"]

    def f32(t):                                  # hidden activations of the split Linears are sp tensors
        from rot_mvgaze_amd import ops
        return ops.merge_sp(t) if t.dtype == torch.float16 else t
    hl = (f32(ht["hl"][0]) > 0).cpu()            # hidden activations are lists (one entry per hidden layer)
    masks["lift"] = [hl[v * B:(v + 1) * B] for v in range(V)]
    D = V * (V - 1)
    for it, rec in enumerate(ht["saved"]):
        if ht.get("mode") == "split":            # heads._forward_split: (xf, h, xh, hh, weight copies...) - h in (scaled) sp, hh fp32
            H1, Hh = [rec[1]], [rec[3]]
        else:
            (X, H1, Xh, Hh, _scales) = rec
        h1, hh = (f32(H1[0]) > 0).cpu(), (f32(Hh[0]) > 0).cpu()
        masks[("fuse", it)] = [h1[d * B:(d + 1) * B] for d in range(D)]
        masks[("head", it)] = [hh[d * B:(d + 1) * B] for d in range(D)]
    return masks


# ResNet-50 at 160 px: the stem gradients have gone through 50 fp32 layers; their max-norm error moves
# between 1.5e-4 and 2.1e-4 with the summation order (stream-K cuts, wgrad split count), hence 4e-4.
@pytest.mark.parametrize("depth,batch,hw,gtol", [(18, 4, 96, GTOL), (50, 2, 160, 2 * GTOL), (18, 2, 224, GTOL),
                                                 (50, 3, 64, 1e-3),    # 12-sample BatchNorm in layer4
                                                 (18, 64, 224, GTOL / 2)])   # the benchmark configuration C2 at full size
@pytest.mark.parametrize("kernels", KERNELS)
def test_backward_strict_with_imposed_relu_pattern(monkeypatch, depth, batch, hw, gtol, kernels):
    """Every parameter gradient (and d/d img) against the fp64 oracle evaluated with the SAME
    ReLU activation pattern as the HIP forward (oracle._relu): isolates the backward kernels from
    the handful of boundary ReLU decisions that fp32 reduction order flips."""
    _strict_backward_check(monkeypatch, depth, batch, hw, gtol, kernels, True)


@pytest.mark.parametrize("depth,batch,hw,gtol", [(18, 4, 96, GTOL), (50, 2, 160, 2 * GTOL), (18, 64, 224, GTOL / 2)])
def test_backward_strict_without_image_gradients_stem_on_the_split_kernels(monkeypatch, depth, batch, hw, gtol):
    """The same check as a training step runs it - the images need no gradient - which is when the 7x7 stem runs on the
    split kernels in its row-window form (forward, weight gradient, the stem tail's backward writing dy in sp)."""
    _strict_backward_check(monkeypatch, depth, batch, hw, gtol, "split", False)


def _strict_backward_check(monkeypatch, depth, batch, hw, gtol, kernels, img_grad):
    from oracle import restatement as R
    select_kernels(monkeypatch, kernels)
    m = build(depth)
    m._debug_keep_tapes = True
    data = inputs(batch, hw, seed=99)
    if img_grad:
        data["img_0"].requires_grad_(True)
        data["img_1"].requires_grad_(True)
    data = m(data)
    assert m._backbone._stem_rw == (kernels == "split" and not img_grad)
    masks = _captured_masks(m)            # before backward: it releases the saved activations
    loss = metrics()(data)
    loss.backward()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_state_dict(depth, 0, 3, perturb_bn=True).items()}
    sd = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in sd.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    inp = synth.make_inputs(batch, 2, 99, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    od = {"img_0": img[:, 0].double().requires_grad_(True), "img_1": img[:, 1].double().requires_grad_(True),
          "rot_0": R.rotation_matrix_2d(hp[:, 0]).double(), "rot_1": R.rotation_matrix_2d(hp[:, 1]).double(),
          "gt_gaze": gt[:, 0], "gt_gaze_1": gt[:, 1]}
    od = R.model_forward(sd, od, depth, 3, True, masks)
    ol = R.iteration_loss(od)
    ol.backward()
    rel_close(loss, ol.item(), TOL, "loss")
    n, errs = 0, []
    for k, p in m.named_parameters():
        if leaves[k].grad is None:
            assert p.grad is None, k
            continue
        g_dev, g_ref = p.grad.detach().cpu().double().numpy(), leaves[k].grad.numpy()
        errs.append((float(np.abs(g_dev - g_ref).max() / (np.abs(g_ref).max() + 1e-30)), k))
        n += 1
    # Full size: 25.7 M max-pool windows; the one or two whose two largest entries differ by less than fp32
    # rounding hand their gradient to the neighbouring pixel in the fp64 oracle (a "pool flip", the max-pool
    # twin of a ReLU flip - the imposed pattern covers ReLUs only).  That moves ~4e-5 of absolute weight
    # gradient in the stem conv (1e-3 of its max) and single pixels of d img; everything downstream of the
    # pool is unaffected (measured: every other gradient <= 1.6e-5).  mvg_conv_wgrad itself matches fp64 to
    # 1e-6 on the stem shape at this size.
    full = batch * hw * hw > 1_000_000
    stem = "_feat_extractor.0.conv1.weight"
    if full:
        stem_err = [e for e, k in errs if k == stem][0]
        assert stem_err <= 5e-3, f"{stem}: {stem_err:.2e}"
        errs = [(e, k) for e, k in errs if k != stem]
    errs.sort(reverse=True)                  # worst first (round 1 forgot the sort and checked one tensor only)
    assert errs[0][0] <= gtol, "worst gradients (max-norm relative error): " + ", ".join(f"{k} {e:.2e}" for e, k in errs[:8])
    assert n == len(leaves) - 2          # everything but the unused fc.weight / fc.bias
    if not img_grad:
        return
    if full:                             # pool flips move single pixels: relative L2 instead of the max norm
        l2_close(data["img_0"].grad, od["img_0"].grad.numpy(), 1e-3, "grad img_0")
        l2_close(data["img_1"].grad, od["img_1"].grad.numpy(), 1e-3, "grad img_1")
    else:
        rel_close(data["img_0"].grad, od["img_0"].grad.numpy(), gtol, "grad img_0")
        rel_close(data["img_1"].grad, od["img_1"].grad.numpy(), gtol, "grad img_1")


@pytest.mark.parametrize("depth,batch,hw", [(18, 1, 64), (50, 1, 96), (18, 5, 40)])
def test_small_and_odd_shapes_against_oracle(depth, batch, hw):
    """Edge shapes: a single sample (4-value BatchNorm in layer4), odd batch, maps whose sizes are not
    multiples of any tile (40 -> 20 -> 10 -> 5 -> 3 -> 2): forward, loss and gradient norms vs the oracle."""
    from oracle import restatement as R
    m = build(depth)
    data = m(inputs(batch, hw, seed=21))
    loss = metrics()(data)
    loss.backward()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_state_dict(depth, 0, 3, perturb_bn=True).items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    inp = synth.make_inputs(batch, 2, 21, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    od = {"img_0": img[:, 0].contiguous(), "img_1": img[:, 1].contiguous(),
          "rot_0": R.rotation_matrix_2d(hp[:, 0]), "rot_1": R.rotation_matrix_2d(hp[:, 1]),
          "gt_gaze": gt[:, 0], "gt_gaze_1": gt[:, 1]}
    od = R.model_forward(sd, od, depth, 3, True)
    ol = R.iteration_loss(od)
    ol.backward()
    # tiny BatchNorm populations amplify fp32 reduction-order noise: 5x the fixture tolerances
    rel_close(loss, ol.item(), 5 * TOL, "loss")
    for i in range(3):
        rel_close(data[f"iter_{i}"]["pred_gaze_0"], od[f"iter_{i}"]["pred_gaze_0"].detach().numpy(), 5 * TOL, "pred")
    for k, p in m.named_parameters():
        if leaves[k].grad is not None:
            l2_close(p.grad.contiguous() if p.grad.dim() == 4 else p.grad, leaves[k].grad.numpy(), l2_bound(depth, batch, hw),
                     "grad " + k)


def test_run_to_run_determinism():
    """Stream-K pieces, wgrad slabs and BN partials are all summed in a fixed order: two runs of the
    same step give bit-identical loss and gradients (batch large enough for stream-K to engage)."""
    m = build(18)
    runs = []
    for _ in range(2):
        m.zero_grad(set_to_none=False)
        data = m(inputs(16, 224, seed=5))
        loss = metrics()(data)
        loss.backward()
        runs.append((loss.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()
                                             if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    for k, g in runs[0][1].items():
        assert torch.equal(g, runs[1][1][k]), k


def test_dp_reducer_one_rank_over_rccl(monkeypatch):
    """The data-parallel path with one rank: RCCL all-reduce of the arena buckets on the reducer's side
    stream, fed by the compute stream AND the low-priority backward-weight stream.  With world size 1
    the averaged gradients must equal the plain ones bit for bit."""
    import socket
    import torch.distributed as dist
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd.dp import GradAllReducer
    monkeypatch.setenv("MVG_RESERVED_CUS", "0")       # same work split (= summation order) as the plain run
    m = build(18)
    data = m(inputs(4, 96, seed=11))
    metrics()(data).backward()
    plain = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=dev())
    try:
        m2 = build(18)
        red = GradAllReducer(m2, bucket_mb=8.0, force=True)
        # stream / CU settings are applied at construction, not from the first publish callback in the
        # middle of the first backward: the backward-weight stream is the same object before and after
        assert m2._backbone.wgrad_low_priority is False and len(red.buckets) > 3
        wg0 = m2._backbone._side(dev())
        data = m2(inputs(4, 96, seed=11))
        metrics()(data).backward()
        assert m2._backbone._wg_stream is wg0 and m2._grad_streams == [wg0]
        torch.cuda.synchronize()
        for k, p in m2.named_parameters():
            if p.grad is not None:
                assert torch.equal(p.grad, plain[k]), k
        # a second step under torch's sync debug mode: issuing the bucket all-reduces must not block the host
        d2 = inputs(4, 96, seed=11)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            m2.zero_grad(set_to_none=True)
            metrics()(m2(d2)).backward()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        for k, p in m2.named_parameters():
            if p.grad is not None:
                assert torch.equal(p.grad, plain[k]), k
    finally:
        dist.destroy_process_group()
        ops.set_reserved_cus(0)


def _dp_rank(rank, port, out_dir):
    import torch.distributed as dist
    from rot_mvgaze_amd.dp import GradAllReducer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MVG_RESERVED_CUS"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        m = build(18, seed=rank)          # replicas start DIFFERENT; the reducer broadcasts rank 0's weights
        GradAllReducer(m, bucket_mb=16.0)
        sd0 = synth.make_state_dict(18, 0, 3, perturb_bn=True)
        for k, v in m.state_dict().items():
            assert torch.equal(v.cpu(), torch.from_numpy(np.array(sd0[k]))), f"rank {rank}: {k} not rank 0's"
        data = m(inputs(3, 64, seed=300 + rank))
        metrics()(data).backward()
        torch.cuda.synchronize()
        torch.save({k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_equals_average_of_independent_steps(tmp_path):
    """N ranks with averaged gradients == the average of N independent steps (BatchNorm
    statistics rank-local).  Two processes share this GPU and reduce over gloo (RCCL needs one GPU per
    rank); everything else - arena buckets, grad-ready order, side streams - is the production path."""
    import socket
    import torch.multiprocessing as mp
    want = None
    for r in range(2):
        m = build(18)
        data = m(inputs(3, 64, seed=300 + r))
        metrics()(data).backward()
        g = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
        want = g if want is None else {k: want[k] + g[k] for k in g}
    want = {k: v * 0.5 for k, v in want.items()}
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_rank, args=(port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = torch.load(tmp_path / f"rank{r}.pt", weights_only=True)
        assert set(got) == set(want)
        for k in want:
            assert torch.equal(got[k], want[k]), (r, k)


def test_bench_self_launch_two_ranks_gloo_rehearsal():
    """`python bench.py --gpus 2 ...` with NO launcher around it and no RANK in the environment: bench.py starts its two
    rank processes itself (bench.launch_ranks) and prints exactly one JSON line.  On this one-GPU box the ranks share
    the device and the gradients travel over gloo (MVG_DIST_BACKEND=gloo): the line says REHEARSAL and is not a
    measurement; on a multi-GPU node the same command runs over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["MVG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "c2", "--batch", "4", "--no-roofline", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["rccl_ranks"] == 2 and j["config"]["global_batch"] == 8
    assert "REHEARSAL" in j["config"]["parallelism"] and j["config"]["dp"]["backend"] == "gloo"
    assert j["value"] > 0 and np.isfinite(j["config"]["loss"])


def test_second_backward_through_released_tape_raises():
    m = build(18)
    d = m(inputs(2, 64))
    loss = metrics()(d)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()


def test_fused_adam_state_dict_round_trip():
    """optimizer.state_dict() carries the flat moments and the step count; a restored optimizer continues
    exactly like the original one."""
    from rot_mvgaze_amd.optim import Adam

    def run(resume_after):
        m = build(18)
        opt = Adam(m.parameters(), lr=1e-3, weight_decay=1e-6)
        for it in range(3):
            if it == resume_after:
                blob = opt.state_dict()
                opt = Adam(m.parameters(), lr=1e-3, weight_decay=1e-6)
                opt.load_state_dict(blob)
            opt.zero_grad()
            metrics()(m(inputs(2, 64, seed=40 + it))).backward()
            opt.step()
        return {k: p.detach().clone() for k, p in m.named_parameters()}, opt.state_dict()
    a, sa = run(None)
    b, sb = run(2)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert sa["mvg_arena_state"][0]["step"] == sb["mvg_arena_state"][0]["step"] == 3
    assert torch.equal(sa["mvg_arena_state"][0]["exp_avg_sq"], sb["mvg_arena_state"][0]["exp_avg_sq"])


def test_gradient_arena_slices_are_16_byte_aligned():
    m = build(18)
    m.ensure_layout()
    arena, entries = m.grad_arena()
    for p, off, n in entries:
        assert off % 4 == 0 and p.data_ptr() % 16 == 0 and m._grad_views[id(p)].data_ptr() % 16 == 0


@pytest.mark.parametrize("depth,dtype", [(18, torch.float32), (50, torch.float32), (18, torch.bfloat16), (50, torch.bfloat16)])
def test_split_path_fused_bn_backward_reduce_switch_gives_the_same_step(depth, dtype):
    """Backbone.fuse_bn_split: the split (and bf16) backward-data launches deliver the BatchNorm-backward sums of the unit
    they feed.  Same forward, same masks, gradients to summation-order noise (bf16: the sums move dy by an ulp here and
    there: that comparison runs on the well-conditioned weight recipe with 16 x 128 x 128 inputs - with 36 values per
    channel in layer4 an ulp of bf16 grows to several per cent on the way down, tests/test_bf16_gpu.py)."""
    grads = []
    bf = dtype == torch.bfloat16
    for fuse in (False, True):
        m = build(depth, conditioned=bf)
        m.compute_dtype = dtype
        m.ensure_layout()
        assert m._backbone.split or bf
        m._backbone.fuse_bn_split = fuse
        d = m(inputs(16, 128, seed=3) if bf else inputs(4, 96, seed=3))
        loss = metrics()(d)
        loss.backward()
        grads.append((loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert grads[0][0] == grads[1][0]
    for k, g in grads[0][1].items():
        if bf:      # relative L2 per tensor: sums of bf16-rounded gradients over 10^5 .. 10^6 elements cancel to ~1e-3 of their terms
            l2_close(grads[1][1][k], g.cpu().numpy(), 5e-2, "fused vs separate reduce (bf16): " + k)
        else:
            rel_close(grads[1][1][k], g.cpu().numpy(), 2e-4, "fused vs separate reduce: " + k)


@pytest.mark.parametrize("depth,dtype", [(18, torch.float32), (50, torch.float32), (50, torch.bfloat16)])
def test_relu_mask_bits_give_the_same_step_as_reading_the_activation(depth, dtype):
    """Residual units hand their ReLU mask to the backward as bits (Backbone.relu_bits, default on): the same mask the
    activation gives, so every gradient is bit-identical to the run that reads the activation."""
    grads = []
    for bits in (False, True):
        m = build(depth)
        m.compute_dtype = dtype
        m.ensure_layout()
        m._backbone.split = False           # the split path always carries the mask as bits
        m._backbone.fuse_bn_split = False   # (bf16: a unit without bits keeps its separate reduce pass - another summation order)
        m._backbone.relu_bits = bits
        d = m(inputs(4, 96, seed=5))
        loss = metrics()(d)
        loss.backward()
        grads.append((loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert grads[0][0] == grads[1][0]
    for k, g in grads[0][1].items():
        assert torch.equal(grads[1][1][k], g), "mask bits vs activation: " + k


@pytest.mark.parametrize("depth,batch,hw", [(18, 4, 96), (50, 3, 128)])
def test_split_and_fp32_mfma_kernels_give_the_same_step(depth, batch, hw):
    """The two conv kernel families of the fp32 model are both fp32-accurate: same loss to 1e-5, gradients to the
    ReLU-flip bound (they round differently, so a boundary ReLU may fall on either side; the well-conditioned weight
    recipe keeps what one flip does to the 50-layer chain below the bound for BOTH noisy runs)."""
    runs = []
    for split in (True, False):
        m = build(depth, conditioned=True)
        m.ensure_layout()
        m._backbone.split = split
        d = m(inputs(batch, hw, seed=11))
        loss = metrics()(d)
        loss.backward()
        runs.append((loss.item(), d["iter_2"]["pred_gaze_1"].detach().clone(),
                     {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    rel_close(runs[0][0], runs[1][0], 1e-5, "loss, split vs fp32-MFMA kernels")
    rel_close(runs[0][1], runs[1][1].cpu().numpy(), TOL, "pred, split vs fp32-MFMA kernels")
    for k, g in runs[1][2].items():
        l2_close(runs[0][2][k], g.cpu().numpy(), GTOL_L2_FLIPS_DEEP, "split vs fp32-MFMA kernels: " + k)   # two flip-noisy runs


def test_split_kernels_step_aside_when_a_view_would_not_fit_32_bit_offsets(monkeypatch):
    """One view of the largest sp tensor (4 bytes per element) must stay below 2 GiB; beyond that the call runs on the
    fp32-MFMA kernels.  Exercised by scaling the guard's size estimate, not by allocating 2 GiB."""
    import rot_mvgaze_amd.backbone as B
    m = build(18)
    m.ensure_layout()
    bb = m._backbone
    assert bb.split
    d = m(inputs(2, 64, seed=1))
    assert bb._split_now
    monkeypatch.setattr(B.Backbone, "_guard_scale", 10 ** 6)
    d2 = m(inputs(2, 64, seed=1))
    assert not bb._split_now
    rel_close(d2["iter_2"]["pred_gaze_1"], d["iter_2"]["pred_gaze_1"].detach().cpu().numpy(), TOL, "pred with and without the split kernels")
    metrics()(d2).backward()                      # and the backward of that call runs on the fp32-MFMA kernels too


def test_gradient_accumulation_and_zero_grad():
    m = build(18)
    d = m(inputs(3, 64))
    metrics()(d).backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    d = m(inputs(3, 64))            # running stats moved, same batch statistics -> same gradients
    metrics()(d).backward()         # accumulates into the existing .grad (PyTorch semantics)
    for k, p in m.named_parameters():
        if p.grad is not None:
            rel_close(p.grad, 2 * g1[k].cpu().numpy(), 1e-5, "accumulated " + k)
    m.zero_grad(set_to_none=False)
    d = m(inputs(3, 64))
    metrics()(d).backward()
    for k, p in m.named_parameters():
        if p.grad is not None:
            rel_close(p.grad, g1[k].cpu().numpy(), 1e-5, "after zero_grad " + k)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-6)      # trainer.py:54 (fc has no grad: skipped)
    opt.step()


MV_CASES = [
    # depth, V, B, hw
    (18, 3, 4, 64), (18, 4, 3, 64), (18, 8, 2, 64),
    (18, 4, 64, 64),            # 12 x 64 = 768 fusion-block rows: the large-tile, split-K Linear launches
    (18, 4, 96, 64),            # 12 x 96 = 1152 rows: the fuser / head Linears on the split-operand kernels (>= 1024 rows)
    (50, 4, 3, 64), (50, 8, 2, 128),                   # ResNet-50 x V > 2 (C3 / C4 / C5 recurrences), small maps
    (50, 4, 4, 224), (50, 8, 2, 224),                  # ... at the benchmark's image size (K_in = 3584 rows of 3584)
]


def _multiview_case(depth, V, B, hw, seed=5, perturb_bn=True, grad_keys=None, gtol=None):
    """One training step of MultiViewGaze against the CPU oracle on the same inputs: loss, every pair's
    features and predictions (rot_mv.py:187-269 per pair) at 1e-4, sampled weight gradients in relative L2."""
    from oracle import restatement as R
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.losses import MultiViewIterationLoss
    from rot_mvgaze_amd.model import MultiViewGaze
    gtol = l2_bound(depth, B, hw) if gtol is None else gtol
    tol = TOL if B * max(hw // 32, 1) ** 2 >= 32 else 2 * TOL     # < 32 values per layer4 channel: see test_small_and_odd_*
    m = MultiViewGaze(depth, 3)
    sdn = synth.make_state_dict(depth, 0, 3, perturb_bn=perturb_bn)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdn.items()})
    m.to(dev()).train()
    inp = synth.make_inputs(B, V, seed, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    rot_d = rotation_matrix_2d(hp.reshape(-1, 2).to(dev())).reshape(B, V, 3, 3)
    out = m.forward_multiview([img[:, v].contiguous().to(dev()) for v in range(V)], rot_d)
    loss = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)(out, gt.to(dev()))
    loss.backward()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and "running" not in k}
    rot = R.rotation_matrix_2d(hp.reshape(-1, 2)).reshape(B, V, 3, 3)
    oo = R.multiview_forward(sd, img, rot, depth, 3, True)
    ol = R.multiview_loss(oo, gt, iter_decay=0.5, rel_weight=0.01, reference_decay=1.0)
    ol.backward()
    rel_close(loss, ol.item(), TOL, "mv loss")
    for pr in R.view_pairs(V):
        for it in range(3):
            for k in ("feat_0", "feat_1", "pred_gaze_0", "pred_gaze_1"):
                rel_close(out["pairs"][pr][f"iter_{it}"][k], oo["pairs"][pr][f"iter_{it}"][k].detach().numpy(), tol,
                          f"pair {pr} iter {it} {k}")
    params = dict(m.named_parameters())
    for k in grad_keys or ("_lifter._lifter.blocks.0.0.weight", "_img_fusers.0._fuser.blocks.0.0.weight",
                           "_gaze_estimators.1.blocks.1.0.weight", "_feat_extractor.0.layer3.0.conv1.weight",
                           "_feat_extractor.0.bn1.bias"):
        l2_close(params[k].grad, leaves[k].grad.numpy(), gtol, "mv grad " + k)
    assert int(m.state_dict()["_feat_extractor.0.bn1.num_batches_tracked"]) == V
    return m


@pytest.mark.parametrize("depth,V,B,hw", MV_CASES, ids=[f"r{d}_V{v}_B{b}_hw{h}" for d, v, b, h in MV_CASES])
def test_multiview_against_oracle(depth, V, B, hw):
    """A9: V > 2 views (configs C3/C4: V = 4, C5: V = 8) for BOTH backbones, at reduced batch, fp32 - shared
    backbone features, every pair equals the two-view oracle recurrence."""
    _multiview_case(depth, V, B, hw)


@pytest.mark.parametrize("depth,V,B,hw", [(50, 4, 2, 128), (18, 4, 96, 64)], ids=["r50_V4_B2_hw128", "r18_V4_B96_hw64_split_linears"])
def test_backward_strict_multiview_with_imposed_relu_pattern(depth, V, B, hw):
    """ResNet-50 x V = 4 (the C3 / C4 network), and ResNet-18 x V = 4 x B = 96 (1152 fusion-block rows: the fuser / head
    Linears run on the split-operand kernels, as at C3): EVERY parameter gradient against the fp64 oracle evaluated with
    the HIP forward's own ReLU pattern, max-norm 4e-4 - the check that the 1e-2..3e-2 relative-L2 figures of the
    free-running comparisons are ReLU flips and not a systematic error of the backward kernels."""
    from oracle import restatement as R
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.losses import MultiViewIterationLoss
    from rot_mvgaze_amd.model import MultiViewGaze
    m = MultiViewGaze(depth, 3)
    sdn = synth.make_state_dict(depth, 0, 3, perturb_bn=True)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdn.items()})
    m.to(dev()).train()
    m._debug_keep_tapes = True
    inp = synth.make_inputs(B, V, 7, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    rot_d = rotation_matrix_2d(hp.reshape(-1, 2).to(dev())).reshape(B, V, 3, 3)
    out = m.forward_multiview(img.to(dev()), rot_d)
    masks = _captured_masks(m, V)
    loss = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)(out, gt.to(dev()))
    loss.backward()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}
    sd = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in sd.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    rot = R.rotation_matrix_2d(hp.reshape(-1, 2)).reshape(B, V, 3, 3).double()
    oo = R.multiview_forward(sd, img.double(), rot, depth, 3, True, masks)
    ol = R.multiview_loss(oo, gt, iter_decay=0.5, rel_weight=0.01, reference_decay=1.0)
    ol.backward()
    rel_close(loss, ol.item(), TOL, "loss")
    errs = []
    for k, p in m.named_parameters():
        if leaves[k].grad is None:
            assert p.grad is None, k
            continue
        g_dev, g_ref = p.grad.detach().cpu().double().numpy(), leaves[k].grad.numpy()
        errs.append((float(np.abs(g_dev - g_ref).max() / (np.abs(g_ref).max() + 1e-30)), k))
    # the stem conv also sees max-pool near-ties resolved the other way in fp64 ("pool flips", see the strict
    # test above): measured 1.0e-3 there, <= 1.5e-4 on every other tensor
    stem = "_feat_extractor.0.conv1.weight"
    assert [e for e, k in errs if k == stem][0] <= 5e-3
    errs = sorted(((e, k) for e, k in errs if k != stem), reverse=True)
    assert errs[0][0] <= 2 * GTOL, "worst gradients (max-norm relative error): " + ", ".join(f"{k} {e:.2e}" for e, k in errs[:8])
    assert len(errs) == len(leaves) - 3


def test_benchmark_configuration_c4_share_full_size_against_oracle():
    """C4's per-GPU share at full size (ResNet-50, V = 4, B = 32, 224 x 224: 128 images, 384 fusion-block
    rows, bench.py's weights and inputs) against the CPU oracle; C3 is the same launch geometry with 4x the
    images per view.  Loss and every pair's features / predictions within 1e-4; one weight gradient per stage."""
    _multiview_case(50, 4, 32, 224, seed=1234, perturb_bn=False,
                    grad_keys=("_feat_extractor.0.conv1.weight", "_feat_extractor.0.layer1.0.conv3.weight",
                               "_feat_extractor.0.layer2.0.downsample.0.weight", "_feat_extractor.0.layer3.2.conv2.weight",
                               "_feat_extractor.0.layer4.2.conv1.weight", "_lifter._lifter.blocks.1.0.weight",
                               "_img_fusers.2._fuser.blocks.0.0.weight", "_gaze_estimators.0.blocks.0.0.weight"))


def test_benchmark_configuration_c3_full_size_against_oracle():
    """C3 - the workload the headline number is quoted on - at FULL size: ResNet-50, V = 4, B = 128, 224 x 224
    (512 images, 1536 fusion-block rows, bench.py's weights and inputs; sp activations of 411 MB per view against
    the 2 GiB guard of the split kernels, more than 2 GiB across views).  The HIP TRAINING forward against the CPU
    oracle's forward (batch statistics) on the same inputs: loss and EVERY pair's gaze predictions and fused
    features of every iteration within the north star's 1e-4 (rot_mv.py:187-269 per pair), and the split-operand
    kernels must be the ones that ran.  The backward runs at full size too: finite gradients everywhere, and the
    gradients that do not need a 512-image CPU autograd graph - d loss / d pooled feature of every view and the
    fusion block's weight gradients, from the oracle's fusion block + loss on the oracle's own pooled features -
    are compared (relative L2 1e-2, measured 2.1e-3 on d loss / d feature: no ReLU of the backbone is involved, but the two sides' pooled
    features differ by up to 1e-4 and flip a few of the fusion block's 1536 x 3584 hidden ReLUs per layer)."""
    from oracle import restatement as R
    from rot_mvgaze_amd.arch import backbone_spec
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.losses import MultiViewIterationLoss
    from rot_mvgaze_amd.model import MultiViewGaze
    depth, V, B, hw = 50, 4, 128, 224
    m = MultiViewGaze(depth, 3)
    sdn = synth.make_state_dict(depth, 0, 3)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}, strict=True)
    m.to(dev()).train()
    inp = synth.make_inputs(B, V, 1234, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    rot_d = rotation_matrix_2d(hp.reshape(-1, 2).to(dev())).reshape(B, V, 3, 3)
    out = m.forward_multiview([img[:, v].contiguous().to(dev()) for v in range(V)], rot_d)
    assert m._backbone._split_now is True
    out["img_feat"].retain_grad()
    loss = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)(out, gt.to(dev()))
    loss.backward()
    for k, p in m.named_parameters():
        assert p.grad is None or bool(torch.isfinite(p.grad).all()), k

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}
    spec = backbone_spec(depth)
    rot = R.rotation_matrix_2d(hp.reshape(-1, 2)).reshape(B, V, 3, 3)
    with torch.no_grad():
        feats = [R.backbone_forward(sd, img[:, v], spec, True) for v in range(V)]
    for v in range(V):
        rel_close(out["img_feat"][v], feats[v].numpy(), TOL, f"C3 pooled feature of view {v}")
    feats = [f.requires_grad_(True) for f in feats]
    head_keys = [k for k in sd if not k.startswith("_feat_extractor") and sd[k].dtype == torch.float32]
    for k in head_keys:
        sd[k].requires_grad_(True)
    lifted = [R.lift(sd, f) for f in feats]
    oo = {"num_iter": 3, "views": V, "pairs": {}}
    for (i, j) in R.view_pairs(V):
        oo["pairs"][(i, j)] = R.fuse_pair(sd, 3, feats[i], feats[j], lifted[i], lifted[j], rot[:, i], rot[:, j])
    ol = R.multiview_loss(oo, gt, iter_decay=0.5, rel_weight=0.01, reference_decay=1.0)
    ol.backward()
    rel_close(loss, ol.item(), TOL, "C3 loss")
    for pr in R.view_pairs(V):
        for it in range(3):
            for k in ("pred_gaze_0", "pred_gaze_1", "feat_0", "feat_1"):
                rel_close(out["pairs"][pr][f"iter_{it}"][k], oo["pairs"][pr][f"iter_{it}"][k].detach().numpy(), TOL,
                          f"C3 pair {pr} iter {it} {k}")
    for v in range(V):
        l2_close(out["img_feat"].grad[v], feats[v].grad.numpy(), 1e-2, f"C3 d loss / d pooled feature of view {v}")
    params = dict(m.named_parameters())
    for k in head_keys:
        l2_close(params[k].grad, sd[k].grad.numpy(), 1e-2, "C3 grad " + k)
    assert int(m.state_dict()["_feat_extractor.0.bn1.num_batches_tracked"]) == V


def test_benchmark_configuration_c3_full_size_backbone_gradients_on_both_kernel_families():
    """C3 at FULL size, the part the oracle test above cannot afford (a 512-image CPU autograd graph): every BACKBONE
    gradient of the split-operand step against the same step on the fp32-MFMA kernels (v_mfma_f32_32x32x2f32: exact fp32
    products; that family is pinned to the oracle by the golden and strict tests at the sizes the CPU finishes).  Same
    weights and inputs as bench.py.  Both runs choose their own ReLU patterns on bench.py's random-init weights (no
    conditioning), so what is bounded is the flip noise of two fp32 implementations of a 50-layer ReLU network, not the
    kernels' 2e-6 (same-pattern agreement: test_backward_strict_*): relative L2 per parameter tensor measured 2.2e-2 at the
    median and 2.9e-2 at worst (layer2.0.bn1.bias) - the level test_against_reference_golden sees against the reference's
    own fixtures (2.1e-2) and C4's share against the oracle (2.0e-2 on the stem) - asserted at 5e-2; loss to 1e-5."""
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.losses import MultiViewIterationLoss
    from rot_mvgaze_amd.model import MultiViewGaze
    depth, V, B, hw = 50, 4, 128, 224
    sdn = synth.make_state_dict(depth, 0, 3)
    inp = synth.make_inputs(B, V, 1234, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    rot_d = rotation_matrix_2d(hp.reshape(-1, 2).to(dev())).reshape(B, V, 3, 3)
    views = [img[:, v].contiguous().to(dev()) for v in range(V)]
    runs = []
    for split in (True, False):
        m = MultiViewGaze(depth, 3)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}, strict=True)
        m.to(dev()).train()
        m.ensure_layout()
        m._backbone.split = split
        out = m.forward_multiview(views, rot_d)
        assert m._backbone._split_now is split
        loss = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)(out, gt.to(dev()))
        loss.backward()
        runs.append((loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
        del m, out, loss
        torch.cuda.empty_cache()
    rel_close(runs[0][0], runs[1][0], 1e-5, "C3 loss, split vs fp32-MFMA kernels")
    errs = []
    for k, g in runs[1][1].items():
        if not k.startswith("_feat_extractor"):
            continue
        ref = g.double()
        errs.append((((runs[0][1][k].double() - ref).norm() / ref.norm().clamp_min(1e-30)).item(), k))
    errs.sort(reverse=True)
    assert len(errs) == 159, len(errs)                      # 53 conv weights + 53 x (gamma, beta)
    assert errs[0][0] < 5e-2, errs[:5]


def test_fused_adam_matches_torch_adam():
    """mvg_adam_step over the arenas == torch.optim.Adam (trainer.py:54: lr, weight_decay 1e-6),
    three steps on the same gradients; CyclicLR drives it like the reference scheduler does."""
    from rot_mvgaze_amd.optim import Adam
    m = build(18)
    opt = Adam(m.parameters(), lr=0, weight_decay=1e-6)
    sched = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=1e-6, max_lr=1e-3, step_size_up=2, step_size_down=2,
                                              mode="triangular2", cycle_momentum=False)
    ref_p = {k: p.detach().cpu().clone().contiguous().requires_grad_(True) for k, p in m.named_parameters()}
    ref_opt = torch.optim.Adam(list(ref_p.values()), lr=0, weight_decay=1e-6)
    ref_sched = torch.optim.lr_scheduler.CyclicLR(ref_opt, base_lr=1e-6, max_lr=1e-3, step_size_up=2, step_size_down=2,
                                                  mode="triangular2", cycle_momentum=False)
    losses = []
    for it in range(3):
        opt.zero_grad()
        d = m(inputs(4, 64, seed=10 + it))
        loss = metrics()(d)
        loss.backward()
        losses.append(loss.item())
        for k, p in m.named_parameters():                # same gradients on the reference side
            ref_p[k].grad = None if p.grad is None else p.grad.detach().cpu().contiguous().clone()
        opt.step()
        ref_opt.step()
        sched.step()
        ref_sched.step()
        assert abs(opt.param_groups[0]["lr"] - ref_opt.param_groups[0]["lr"]) < 1e-12
    for k, p in m.named_parameters():
        rel_close(p.detach(), ref_p[k].detach().numpy(), 2e-6, "param after Adam " + k)
    sd = m.state_dict()
    assert torch.equal(sd["_feat_extractor.0.fc.weight"].cpu(), ref_p["_feat_extractor.0.fc.weight"].detach())
    assert all(np.isfinite(losses))


def test_raw_uint8_input_pipeline():
    """raw uint8 HWC patches normalised on the GPU == the float NCHW path fed
    with ((u8/255) - mean)/std.  The torchvision transforms the reference uses (ToTensor, Normalize,
    main.py:50-55) are not importable here: this row's host restatement follows their documented
    arithmetic - parity unpinned against the reference itself."""
    rng = np.random.default_rng(0)
    B, hw = 3, 64
    u8 = [torch.from_numpy(rng.integers(0, 256, size=(B, hw, hw, 3), dtype=np.uint8)) for _ in range(2)]
    mean = torch.tensor([0.485, 0.456, 0.406])
    std = torch.tensor([0.229, 0.224, 0.225])

    def host(x, bgr):
        x = x.flip(-1) if bgr else x
        return ((x.float().div(255) - mean) / std).permute(0, 3, 1, 2).contiguous()
    d = inputs(B, hw)
    for bgr in (False, True):
        m = build(18, train=False)
        m.input_bgr = bgr
        m.input_size = None                       # keep the patch size (the reference resizes to 224: next test)
        with torch.no_grad():
            a = m({"img_0": u8[0].to(dev()), "img_1": u8[1].to(dev()), "rot_0": d["rot_0"], "rot_1": d["rot_1"]})
            b = m({"img_0": host(u8[0], bgr).to(dev()), "img_1": host(u8[1], bgr).to(dev()),
                   "rot_0": d["rot_0"], "rot_1": d["rot_1"]})
        assert torch.equal(a["pred_gaze"], b["pred_gaze"]) and torch.equal(a["img_feat_1"], b["img_feat_1"])


def test_raw_uint8_input_is_resized_like_the_reference_transform():
    """test_transform of main.py:50-55 on the GPU: uint8 patches of another size go through
    Resize((input_size, input_size), antialias=True) between ToTensor and Normalize.  The float path is
    fed with the oracle's restatement of that transform (pinned to the ATen op torchvision calls:
    tests/golden/resize_aa.npz); the resize kernel itself is held to 3e-6 in tests/test_kernels_gpu.py,
    here the predictions must agree within the model tolerance."""
    from oracle import restatement as R
    rng = np.random.default_rng(1)
    B, size = 3, 64
    u8 = [rng.integers(0, 256, size=(B, 100, 90, 3), dtype=np.uint8) for _ in range(2)]
    d = inputs(B, size)
    m = build(18, train=False)
    m.input_size = size
    assert build(18, train=False).input_size == 224          # main.py:40
    host = [torch.from_numpy(R.preprocess_u8(x, size, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))) for x in u8]
    with torch.no_grad():
        a = m({"img_0": torch.from_numpy(u8[0]).to(dev()), "img_1": torch.from_numpy(u8[1]).to(dev()),
               "rot_0": d["rot_0"], "rot_1": d["rot_1"]})
        b = m({"img_0": host[0].to(dev()), "img_1": host[1].to(dev()), "rot_0": d["rot_0"], "rot_1": d["rot_1"]})
    assert a["pred_gaze"].shape == b["pred_gaze"].shape
    rel = (a["pred_gaze"] - b["pred_gaze"]).abs().max() / b["pred_gaze"].abs().max()
    assert rel < 1e-4, rel


def test_benchmark_configuration_c2_full_size_against_oracle():
    """The exact workload `bench.py` times (C2: ResNet-18, V = 2, B = 64, 224 x 224, bench.py's weights and
    inputs) against the CPU oracle on the same inputs: loss and every iteration's gaze predictions within
    the north star's 1e-4, weight gradients of one layer per stage in relative L2 (ReLU flips, see above).
    Full size means full-size launches: stream-K grids, merged stride-2 classes, split-K Linears."""
    from oracle import restatement as R
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.losses import MultiViewIterationLoss
    from rot_mvgaze_amd.model import MultiViewGaze
    depth, B, V, hw = 18, 64, 2, 224
    m = MultiViewGaze(depth, 3)
    sdn = synth.make_state_dict(depth, 0, 3)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}, strict=True)
    m.to(dev()).train()
    inp = synth.make_inputs(B, V, 1234, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    rot_d = rotation_matrix_2d(hp.reshape(-1, 2).to(dev())).reshape(B, V, 3, 3)
    out = m.forward_multiview([img[:, v].contiguous().to(dev()) for v in range(V)], rot_d)
    crit = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)
    loss = crit(out, gt.to(dev()))
    loss.backward()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and "running" not in k}
    rot = R.rotation_matrix_2d(hp.reshape(-1, 2)).reshape(B, V, 3, 3)
    oo = R.multiview_forward(sd, img, rot, depth, 3, True)
    ol = R.multiview_loss(oo, gt, iter_decay=0.5, rel_weight=0.01, reference_decay=1.0)
    ol.backward()
    rel_close(loss, ol.item(), TOL, "C2 loss")
    for it in range(3):
        for k in ("pred_gaze_0", "pred_gaze_1", "feat_0", "feat_1"):
            rel_close(out["pairs"][(0, 1)][f"iter_{it}"][k], oo["pairs"][(0, 1)][f"iter_{it}"][k].detach().numpy(), TOL,
                      f"C2 iter {it} {k}")
    params = dict(m.named_parameters())
    for k in ("_feat_extractor.0.conv1.weight", "_feat_extractor.0.layer1.0.conv1.weight",
              "_feat_extractor.0.layer2.0.downsample.0.weight", "_feat_extractor.0.layer3.1.conv2.weight",
              "_feat_extractor.0.layer4.1.conv2.weight", "_lifter._lifter.blocks.1.0.weight",
              "_img_fusers.2._fuser.blocks.0.0.weight", "_gaze_estimators.0.blocks.0.0.weight"):
        # measured: fusion block 1e-6, backbone 3e-3 (layer4) .. 6e-3 (stem) - a few of layer4's 1.6 M
        # outputs sit within fp32 rounding of the ReLU threshold (same-mask comparison: 2e-4, strict test above)
        l2_close(params[k].grad, leaves[k].grad.numpy(), GTOL_L2_FLIPS, "C2 grad " + k)


def test_training_step_does_not_synchronise_the_host():
    """Neither API's training step may contain a host/GPU synchronisation (a device value read on the
    host, a copy from pageable memory - e.g. indexing a device tensor with a Python list): one such copy
    in the loss held the whole step back by 5-7 %.  torch's sync debug mode raises on the synchronising
    calls torch itself would make; the library's own launches never synchronise."""
    from rot_mvgaze_amd.losses import MultiViewIterationLoss
    from rot_mvgaze_amd.optim import Adam
    m = build(18)
    crit, crit_mv = metrics(), MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)
    opt = Adam(m.parameters(), lr=1e-4, weight_decay=1e-6)
    d = inputs(4, 64)
    img = [d["img_0"], d["img_1"]]
    rot = torch.stack([d["rot_0"], d["rot_1"]], 1).contiguous()
    gt = torch.stack([d["gt_gaze"], d["gt_gaze_1"]], 1).contiguous()

    def step_dict():
        opt.zero_grad()
        data = m(dict(d))
        crit(data).backward()
        opt.step()

    def step_mv():
        m.zero_grad(set_to_none=True)
        crit_mv(m.forward_multiview(img, rot), gt).backward()
        opt.step()
    for fn in (step_dict, step_mv):
        fn()                                     # first call: arenas, index tensors, scratch buffers
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            fn()
            fn()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()


def test_inference_weight_cache_follows_parameter_updates():
    """The inference path keeps sp copies of the conv weights between calls; an optimizer step (torch's or the fused
    one, which writes through raw pointers), load_state_dict or an in-place edit must invalidate them."""
    from rot_mvgaze_amd.optim import Adam
    m = build(18)
    x = inputs(3, 64, seed=4)

    def infer():
        m.eval()
        with torch.no_grad():
            return m(dict(x))["iter_2"]["pred_gaze_1"].clone()
    p0 = infer()
    assert torch.equal(infer(), p0) and len(m._backbone._wk_cache) > 0          # second call: cache hits, same result
    m.train()
    opt = Adam(m.parameters(), lr=1e-2)
    metrics()(m(dict(x))).backward()
    opt.step()
    p1 = infer()
    assert not torch.equal(p1, p0), "weights changed by the fused Adam step: the cached copies must not be used"
    ref = build(18)
    ref.load_state_dict(m.state_dict())
    ref.eval()
    with torch.no_grad():
        want = ref(dict(x))["iter_2"]["pred_gaze_1"]
    rel_close(p1, want.cpu().numpy(), 1e-6, "inference after an optimizer step vs a fresh model with the same weights")
    with torch.no_grad():
        next(p for n, p in m.named_parameters() if n.endswith("layer1.0.conv1.weight")).mul_(1.5)
    assert not torch.equal(infer(), p1)


def test_inference_weight_cache_misses_writes_through_data_until_invalidated():
    """The documented limit of that cache: it is keyed on the parameters' version counters, which a write through
    ``p.data`` does not move.  ``model.invalidate_weight_cache()`` (or ``train()``) is the remedy."""
    m = build(18)
    x = inputs(3, 64, seed=4)
    m.eval()

    def infer():
        with torch.no_grad():
            return m(dict(x))["iter_2"]["pred_gaze_1"].clone()
    p0 = infer()
    w = next(p for n, p in m.named_parameters() if n.endswith("layer1.0.conv1.weight"))
    w.data.mul_(1.5)                                 # bypasses the version counter
    assert torch.equal(infer(), p0), "expected the stale cached copy here - if this fails the cache key got stronger: update the docs"
    m.invalidate_weight_cache()
    p1 = infer()
    assert not torch.equal(p1, p0)
    w.data.mul_(1.5)
    m.train()                                        # entering train mode drops the copies as well
    m.eval()
    assert not torch.equal(infer(), p1)


def test_backward_of_a_tape_whose_weight_copies_were_overwritten_raises():
    """Tapes point into the persistent buffers that hold a step's sp / bf16 weight copies.  forward A, weights change,
    forward B, backward A would run backward-data of A with B's weights: it raises instead (like PyTorch's
    version-counter check).  Without a weight change in between the older tape is still good."""
    m = build(18)
    xa, xb = inputs(2, 64, seed=1), inputs(2, 64, seed=2)
    da = m(dict(xa))
    db = m(dict(xb))                                 # same weights: rewrites the copies with identical values
    metrics()(db).backward()
    metrics()(da).backward()                         # fine
    da = m(dict(xa))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("conv1.weight"):
                p.mul_(1.0001)                       # an optimizer step (bumps the version counters)
    db = m(dict(xb))
    with pytest.raises(RuntimeError, match="DIFFERENT weights"):
        metrics()(da).backward()


def test_scratch_workspace_is_registered_by_the_caller_not_allocated_by_the_library():
    """the caller owns every buffer incl. workspace.  The first launch on a stream registers a torch-owned
    workspace (mvg_set_scratch); without one the stream-K / two-level launches run their scratch-free forms and
    give the same results to summation order."""
    from rot_mvgaze_amd import ops
    from rot_mvgaze_amd._lib import ConvDesc, lib
    st = torch.cuda.current_stream()
    d = ConvDesc.make(2, 8, 112, 112, 4, 64, 7, 2, 3)
    x = torch.randn(2, 8, 112, 112, 4, device=dev())
    w = torch.randn(64, 7, 7, 4, device=dev()) * 0.05
    y1 = torch.empty(2, 8, 56, 56, 64, device=dev())
    ops.conv_fprop(d, x, w, y1)
    key = (st.device_index, st.cuda_stream)
    assert key in ops._workspaces and ops._workspaces[key].numel() == lib().mvg_scratch_bytes() > 0
    assert torch.cuda.memory_allocated() >= ops._workspaces[key].numel()      # visible to PyTorch's accounting
    ops.release_workspaces()
    assert key not in ops._workspaces
    import ctypes as C
    raw = ops._s
    try:
        ops._s = lambda scratch=False: C.c_void_p(torch.cuda.current_stream().cuda_stream)     # NO workspace registered
        y2 = torch.empty_like(y1)
        ops.conv_fprop(d, x, w, y2)
    finally:
        ops._s = raw
    rel_close(y2, y1.cpu().numpy(), 1e-5, "stem conv without a scratch workspace")


def test_view_swap_symmetry_eval():
    m = build(18, train=False)
    with torch.no_grad():
        a = m(inputs(2, 64))
        d = inputs(2, 64)
        b = m({"img_0": d["img_1"], "img_1": d["img_0"], "rot_0": d["rot_1"], "rot_1": d["rot_0"]})
    for i in range(3):
        assert torch.equal(a[f"iter_{i}"]["pred_gaze_0"], b[f"iter_{i}"]["pred_gaze_1"])
        assert torch.equal(a[f"iter_{i}"]["feat_1"], b[f"iter_{i}"]["feat_0"])
