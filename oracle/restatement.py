"""CPU restatement of the Rot-MVGaze hot path (torch-CPU functional ops, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) - the checker, never the product.
Parity: PINNED against fixtures generated from the reference's own Python by
tests/golden/make_golden.py (tests/test_oracle_golden.py).

Everything is written functionally over a ``state_dict`` (name -> tensor) that uses the
reference's checkpoint key names, so the same dict drives the reference model (via
``load_state_dict``), this oracle and the HIP module.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

import rot_mvgaze_amd  # noqa: F401  (only the pure-python arch description is used)
from rot_mvgaze_amd.arch import NUM_FEAT_VEC, BackboneSpec, ConvSpec, backbone_spec

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------------
def rotation_matrix_2d(pitch_yaw: Tensor, inverse: bool = False) -> Tensor:
    """(pitch, yaw) -> R = Ry(yaw) @ Rx(-pitch); inverse -> transpose.

    Follows /root/reference/utils/math.py:188-219 (pitch negated :199, Rx :206-209, Ry :210-213,
    product :216, transpose :217-218; 1-D input promoted to [1,2] :196-197).
    """
    hp = pitch_yaw
    if hp.dim() == 1:
        hp = hp.unsqueeze(0)
    p = -hp[:, 0]
    y = hp[:, 1]
    cp, sp, cy, sy = torch.cos(p), torch.sin(p), torch.cos(y), torch.sin(y)
    one, zero = torch.ones_like(cp), torch.zeros_like(cp)
    rx = torch.stack([one, zero, zero, zero, cp, -sp, zero, sp, cp], dim=1).view(-1, 3, 3)
    ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).view(-1, 3, 3)
    r = torch.matmul(ry, rx)
    return r.transpose(1, 2) if inverse else r


def pitchyaw_to_vector(py: Tensor) -> Tensor:
    """v = (cos p sin y, sin p, cos p cos y); /root/reference/utils/math.py:52-60.
    The reference allocates the output with torch.empty in the default dtype (fp32)."""
    s, c = torch.sin(py), torch.cos(py)
    return torch.stack([c[:, 0] * s[:, 1], s[:, 0], c[:, 0] * c[:, 1]], dim=1).to(torch.float32)


def angular_error_numpy(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Evaluation metric in degrees; /root/reference/utils/math.py:105-120 (float64 numpy)."""
    def vec(x):
        if x.shape[1] != 2:
            return x
        s, c = np.sin(x), np.cos(x)
        out = np.empty((x.shape[0], 3))
        out[:, 0] = c[:, 0] * s[:, 1]
        out[:, 1] = s[:, 0]
        out[:, 2] = c[:, 0] * c[:, 1]
        return out
    a, b = vec(a), vec(b)
    ab = np.sum(a * b, axis=1)
    na = np.clip(np.linalg.norm(a, axis=1), 1e-7, None)
    nb = np.clip(np.linalg.norm(b, axis=1), 1e-7, None)
    return np.arccos(ab / (na * nb)) * 180.0 / np.pi


# --------------------------------------------------------------------------------------------
# backbone
# --------------------------------------------------------------------------------------------
def _relu(x: Tensor, masks) -> Tensor:
    """ReLU; with ``masks`` (an iterator of 0/1 tensors shaped like x) the activation pattern is
    IMPOSED (x * mask) instead of derived from the sign of x.  Test aid: fp32 reduction-order noise
    flips a handful of ReLU decisions on elements within rounding of 0, which changes gradients
    discontinuously; imposing the checked implementation's pattern makes gradients comparable to
    1e-4 (tests/test_model_gpu.py::test_backward_strict_with_imposed_relu_pattern)."""
    if masks is None:
        return F.relu(x)
    return x * next(masks).to(x.dtype)


def bf16_round(x: Tensor) -> Tensor:
    """Round to bfloat16 and back (round-to-nearest-even); differentiable like any dtype cast."""
    return x.to(torch.bfloat16).to(x.dtype)


def _conv_bn(sd: SD, x: Tensor, c: ConvSpec, training: bool, q=None) -> Tensor:
    """conv (bias=False) -> BatchNorm2d(eps 1e-5, momentum 0.1).
    Train mode: batch statistics over this call's (B,H,W); running stats updated in place in
    ``sd`` (unbiased var, num_batches_tracked += 1) exactly like nn.BatchNorm2d.
    /root/reference/models/resnet.py:31-47 (convs), :187,73 (BN use).

    ``q`` (None for the reference arithmetic) emulates the repo's bf16 STORAGE path, which the reference
    does not have: the conv reads bf16 weights, its fp32 result gives the batch statistics, the result is
    then stored rounded (``q``) and the normalisation y*scale + shift is applied to the stored values -
    the rounding points of rot-mvgaze_amd/csrc/conv_bf16.hip + bn.hip, so that the bf16 kernels can be
    compared with the same arithmetic instead of with fp32 (tests/test_bf16_gpu.py)."""
    w = sd[c.name + ".weight"]
    if q is None:
        y = F.conv2d(x, w, None, c.stride, c.pad)
        if training:
            sd[c.bn + ".num_batches_tracked"] += 1
        return F.batch_norm(y, sd[c.bn + ".running_mean"], sd[c.bn + ".running_var"],
                            sd[c.bn + ".weight"], sd[c.bn + ".bias"], training, 0.1, 1e-5)
    y = F.conv2d(x, q(w), None, c.stride, c.pad)
    rm, rv = sd[c.bn + ".running_mean"], sd[c.bn + ".running_var"]
    if training:
        sd[c.bn + ".num_batches_tracked"] += 1
        mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
        n = y.numel() // y.shape[1]
        with torch.no_grad():
            rm.mul_(0.9).add_(0.1 * mean.detach())
            rv.mul_(0.9).add_(0.1 * var.detach() * (n / max(n - 1, 1)))
    else:
        mean, var = rm, rv
    scale = sd[c.bn + ".weight"] * torch.rsqrt(var + 1e-5)
    shift = sd[c.bn + ".bias"] - mean * scale
    return q(y) * scale[None, :, None, None] + shift[None, :, None, None]


def backbone_forward(sd: SD, x: Tensor, spec: BackboneSpec, training: bool, trace: Optional[list] = None,
                     relu_masks=None, q=None) -> Tensor:
    """One ResNet pass over ONE view's images [B,3,H,W] -> pooled feature [B, fc_dim].

    /root/reference/models/resnet.py:261-275 (stem, maxpool, layer1..4, avgpool) wrapped by
    /root/reference/models/rot_mv.py:124-128 (avgpool again - identity on 1x1 - and flatten);
    residual blocks :80-96 (basic) and :128-148 (bottleneck).  ``q``: see _conv_bn (every stored
    activation - input image, unit outputs, block outputs - is rounded; None = reference arithmetic)."""
    s = (lambda t: t) if q is None else q
    x = s(_relu(_conv_bn(sd, s(x), spec.stem, training, q), relu_masks))
    x = F.max_pool2d(x, 3, 2, 1)
    if trace is not None:
        trace.append(x)
    for blk in spec.blocks:
        identity = x
        out = x
        for i, c in enumerate(blk.convs):
            out = _conv_bn(sd, out, c, training, q)
            if i + 1 < len(blk.convs):
                out = s(_relu(out, relu_masks))
        if blk.downsample is not None:
            identity = _conv_bn(sd, x, blk.downsample, training, q)      # normalised on the fly by the consumer: never stored
        x = s(_relu(out + identity, relu_masks))
        if trace is not None:       # debugging aid: per-block activations (tests may retain_grad them)
            trace.append(x)
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


# --------------------------------------------------------------------------------------------
# MLP blocks, lifter, fuser, head
# --------------------------------------------------------------------------------------------
_LINEAR_Q = None      # set by multiview_forward(storage=...): operand rounding of the fusion block's Linear layers


def mlp(sd: SD, prefix: str, x: Tensor, n_layers: int, relu_masks=None) -> Tensor:
    """[Linear, ReLU]*(n-1), Linear.  /root/reference/models/backbones/blocks.py:27-82.
    With the repo's bf16 path emulated (_LINEAR_Q), the operands of every Linear wider than 4 outputs are
    rounded (activations and weights), the product accumulates in fp32 and the result stays fp32 - the
    arithmetic of mvg_linear_*_mixed; the 512 -> 2 head layer stays fp32."""
    q = _LINEAR_Q
    for i in range(n_layers):
        w, b = sd[f"{prefix}blocks.{i}.0.weight"], sd[f"{prefix}blocks.{i}.0.bias"]
        if q is not None and w.shape[0] > 4:
            x = F.linear(q(x), q(w), b)
        else:
            x = F.linear(x, w, b)
        if i + 1 < n_layers:
            x = _relu(x, relu_masks)
    return x


def _one(mask):
    return None if mask is None else iter([mask])


def lift(sd: SD, img_feat: Tensor, mask=None) -> Tensor:
    """Feat3dLifter: Mlp(C_f,[1536,1536]) -> [B,3,512].  /root/reference/models/rot_mv.py:91-98."""
    return mlp(sd, "_lifter._lifter.", img_feat, 2, _one(mask)).reshape(-1, 3, NUM_FEAT_VEC)


def fuse(sd: SD, it: int, img_feat: Tensor, rotated: Tensor, mask=None) -> Tensor:
    """ImageFeatFuser: cat([img_feat, rotated.flatten(-2,-1)]) -> Mlp(K_in,[K_in,1536]) -> [B,3,512].
    /root/reference/models/rot_mv.py:35-50, reshape at :234-239."""
    x = torch.cat([img_feat, rotated.flatten(-2, -1)], dim=-1)
    return mlp(sd, f"_img_fusers.{it}._fuser.", x, 2, _one(mask)).reshape(-1, 3, NUM_FEAT_VEC)


def fuse_rotmat(sd: SD, it: int, img_feat: Tensor, feat: Tensor, rot: Tensor) -> Tensor:
    """ImageRotmatFeatFuser (encode_rotmat): the partner's feature is NOT rotated; the relative
    rotation is appended as 9 more inputs of a 3-layer Mlp.  /root/reference/models/rot_mv.py:53-69,219-225."""
    x = torch.cat([img_feat, feat.flatten(-2, -1), rot.flatten(-2, -1)], dim=-1)
    return mlp(sd, f"_img_fusers.{it}._fuser.", x, 3).reshape(-1, 3, NUM_FEAT_VEC)


def intensity_batchnorm(sd: SD, prefix: str, x: Tensor, training: bool, momentum: float = 0.05,
                        eps: float = 1e-4) -> Tensor:
    """IntensityBatchNorm: x [B,3,F] divided by a running mean of the batch std of the column
    norms; the running buffer is updated BEFORE it is used (train).  /root/reference/models/rot_mv.py:13-32."""
    key = prefix + "running_mean"
    intensity = torch.norm(x, dim=-2, keepdim=True).detach()
    var = torch.var(intensity, unbiased=False, dim=0, keepdim=True)
    std = torch.sqrt(var.clamp_min(eps))
    if training:
        sd[key] = sd[key] * (1 - momentum) + std * momentum
    return x / (sd[key] + eps)


def fuse_rotfeat(sd: SD, it: int, feat_0: Tensor, feat_1: Tensor, training: bool) -> Tensor:
    """RotFeatFuser (share_feature): both [B,3,F] inputs pass the fuser's ONE IntensityBatchNorm
    (feat_0 first), are concatenated along F, flattened axis-major and go through a 3-layer Mlp.
    /root/reference/models/rot_mv.py:72-86."""
    pre = f"_img_fusers.{it}._batchnorm."
    n0 = intensity_batchnorm(sd, pre, feat_0, training)
    n1 = intensity_batchnorm(sd, pre, feat_1, training)
    x = torch.cat([n0, n1], dim=-1).flatten(-2, -1)
    return mlp(sd, f"_img_fusers.{it}._fuser.", x, 3).reshape(-1, 3, NUM_FEAT_VEC)


def gaze_head(sd: SD, it: int, img_feat: Tensor, feat: Tensor, mask=None) -> Tensor:
    """Mlp(K_in,[512,2]) on cat([img_feat, feat.flatten(1)]).  /root/reference/models/rot_mv.py:179-184,249-254.
    share_feature (img_feat is [B,3,F]): cat along F, then flatten (:241-247)."""
    if img_feat.dim() == 3:
        x = torch.cat([img_feat, feat], dim=-1).flatten(1, -1)
    else:
        x = torch.cat([img_feat, feat.flatten(1, -1)], dim=-1)
    return mlp(sd, f"_gaze_estimators.{it}.", x, 2, _one(mask))


def fuse_pair(sd: SD, num_iter: int, img_feat_0: Tensor, img_feat_1: Tensor, f0: Tensor, f1: Tensor,
              rot_0: Tensor, rot_1: Tensor, masks: Optional[Dict[Any, Any]] = None, variant=None,
              training: bool = False) -> Dict[str, Any]:
    """The two-view recurrence of /root/reference/models/rot_mv.py:193-194,205-265 given the
    per-view pooled and lifted features.  ``masks`` (test aid, see _relu, default variant only):
    {("fuse", it): [m0, m1], ("head", it): [m0, m1]} imposed hidden-layer ReLU patterns.
    ``variant``: arch.Variant (ablations, :136-171,219-247); share_weights needs nothing here -
    the state_dict holds the same tensors under every iteration's names."""
    mk = (lambda kind, it, v: masks[(kind, it)][v]) if masks is not None else (lambda kind, it, v: None)
    enc = variant is not None and variant.encode_rotmat
    ign = variant is not None and variant.ignore_rotmat
    shf = variant is not None and variant.share_feature
    assert masks is None or not (enc or shf)
    rot_10 = rot_0 @ rot_1.transpose(-1, -2)
    rot_01 = rot_1 @ rot_0.transpose(-1, -2)
    if shf:                                            # :199-201
        img_feat_0, img_feat_1 = f0, f1
    pred: Dict[str, Any] = {
        "num_iter": num_iter,
        "img_feat_0": img_feat_0, "img_feat_1": img_feat_1,
        "initial_rot_feat_0": f0, "initial_rot_feat_1": f1,
    }
    for it in range(num_iter):
        f0_prev = f0                                   # rot_mv.py:217 - view 1 reads view 0's OLD feature
        if enc:
            f0 = fuse_rotmat(sd, it, img_feat_0, f1, rot_10)
            f1 = fuse_rotmat(sd, it, img_feat_1, f0_prev, rot_01)
        elif ign:
            f0 = fuse(sd, it, img_feat_0, f1, mk("fuse", it, 0))
            f1 = fuse(sd, it, img_feat_1, f0_prev, mk("fuse", it, 1))
        elif shf:
            f0 = fuse_rotfeat(sd, it, img_feat_0, rot_10 @ f1, training)
            f1 = fuse_rotfeat(sd, it, img_feat_1, rot_01 @ f0_prev, training)
        else:
            f0 = fuse(sd, it, img_feat_0, rot_10 @ f1, mk("fuse", it, 0))
            f1 = fuse(sd, it, img_feat_1, rot_01 @ f0_prev, mk("fuse", it, 1))
        pred[f"iter_{it}"] = {
            "feat_0": f0, "feat_1": f1,
            "pred_gaze_0": gaze_head(sd, it, img_feat_0, f0, mk("head", it, 0)),
            "pred_gaze_1": gaze_head(sd, it, img_feat_1, f1, mk("head", it, 1)),
        }
    pred["pred_gaze"] = pred[f"iter_{num_iter - 1}"]["pred_gaze_0"]
    return pred


def model_forward(sd: SD, data: Dict[str, Any], depth: int, num_iter: int = 3,
                  training: bool = False, masks: Optional[Dict[Any, Any]] = None, variant=None) -> Dict[str, Any]:
    """FeatRotationSymm.forward - /root/reference/models/rot_mv.py:187-269.
    Mutates and returns ``data`` like the reference (:266).  ``masks`` (test aid, see _relu):
    {"backbone": [iter_view0, iter_view1], "lift": [m0, m1], ("fuse", it): ..., ("head", it): ...}."""
    spec = backbone_spec(depth)
    bm = masks["backbone"] if masks is not None else [None, None]
    lm = masks["lift"] if masks is not None else [None, None]
    img_feat_0 = backbone_forward(sd, data["img_0"], spec, training, None, bm[0])   # view 0 first: BN running
    img_feat_1 = backbone_forward(sd, data["img_1"], spec, training, None, bm[1])   # stats update order :196-197
    f0, f1 = lift(sd, img_feat_0, lm[0]), lift(sd, img_feat_1, lm[1])
    data.update(fuse_pair(sd, num_iter, img_feat_0, img_feat_1, f0, f1, data["rot_0"], data["rot_1"], masks, variant,
                          training))
    return data


# --------------------------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------------------------
def gaze_angular_loss(pred: Tensor, gt: Tensor) -> Tensor:
    """mean over the batch of acos(clamp(cos_sim(v(gt), v(pred), eps=1e-6), -1, 1)) in degrees.
    /root/reference/losses/gaze_loss.py:42-52."""
    sim = F.cosine_similarity(pitchyaw_to_vector(gt), pitchyaw_to_vector(pred), eps=1e-6)
    return torch.mean(torch.acos(F.hardtanh(sim, -1.0, 1.0)) * (180 / np.pi))


def stereo_loss(pred0: Tensor, pred1: Tensor, gt0: Tensor, gt1: Tensor,
                rel_weight: float = 0.01, reference_decay: float = 1.0) -> Tensor:
    """/root/reference/losses/stereo_loss.py:46-54 with main.py:239's constants."""
    return (gaze_angular_loss(pred0, gt0) + gaze_angular_loss(pred1, gt1) * reference_decay) * rel_weight


def iteration_loss(data: Dict[str, Any], iter_decay: float = 0.5, rel_weight: float = 0.01,
                   reference_decay: float = 1.0) -> Tensor:
    """Horner accumulation over iterations; /root/reference/losses/stereo_loss.py:65-84
    (``additional_decay`` is None on the path, main.py:240)."""
    total: Any = 0
    for i in range(data["num_iter"]):
        it = data[f"iter_{i}"]
        total = total * iter_decay + stereo_loss(it["pred_gaze_0"], it["pred_gaze_1"],
                                                 data["gt_gaze"], data["gt_gaze_1"],
                                                 rel_weight, reference_decay)
    return total


# --------------------------------------------------------------------------------------------
# V > 2 views (SURVEY.md §8(a) A9 - NOT in the reference; defined so that every pair equals the
# two-view reference recurrence on that pair given the shared per-view backbone features)
# --------------------------------------------------------------------------------------------
def view_pairs(views: int) -> List[Tuple[int, int]]:
    return [(i, j) for i in range(views) for j in range(i + 1, views)]


def multiview_forward(sd: SD, img: Tensor, rot: Tensor, depth: int, num_iter: int = 3,
                      training: bool = False, masks: Optional[Dict[Any, Any]] = None, storage=None) -> Dict[str, Any]:
    """img [B,V,3,H,W], rot [B,V,3,3].  Backbone + lifter once per view in view order (BN
    statistics per view call); fusion over every unordered pair i<j in lexicographic order.
    ``masks`` (test aid, see _relu): {"backbone": [iterator per view], "lift": [mask per view],
    ("fuse", it) / ("head", it): [mask per DIRECTED pair d = 2p (i<-j), 2p+1 (j<-i)]}.
    ``storage`` = bf16_round emulates the repo's bf16 backbone storage (see _conv_bn); the fusion block is fp32."""
    global _LINEAR_Q
    spec = backbone_spec(depth)
    V = img.shape[1]
    _LINEAR_Q = storage
    try:
        return _multiview_forward(sd, img, rot, depth, num_iter, training, masks, storage, spec, V)
    finally:
        _LINEAR_Q = None


def _multiview_forward(sd, img, rot, depth, num_iter, training, masks, storage, spec, V):
    bm = masks["backbone"] if masks is not None else [None] * V
    lm = masks["lift"] if masks is not None else [None] * V
    feats = [backbone_forward(sd, img[:, v], spec, training, None, bm[v], storage) for v in range(V)]
    lifted = [lift(sd, f, lm[v]) for v, f in enumerate(feats)]
    out: Dict[str, Any] = {"num_iter": num_iter, "views": V, "img_feat": feats, "initial_rot_feat": lifted,
                           "pairs": {}}
    for p, (i, j) in enumerate(view_pairs(V)):
        pm = None
        if masks is not None:
            pm = {k: v[2 * p:2 * p + 2] for k, v in masks.items() if isinstance(k, tuple)}
        out["pairs"][(i, j)] = fuse_pair(sd, num_iter, feats[i], feats[j], lifted[i], lifted[j],
                                         rot[:, i], rot[:, j], pm)
    out["pred_gaze"] = out["pairs"][(0, 1)][f"iter_{num_iter - 1}"]["pred_gaze_0"]
    return out


def multiview_loss(out: Dict[str, Any], gt: Tensor, iter_decay: float = 0.5, rel_weight: float = 0.01,
                   reference_decay: float = 1.0) -> Tensor:
    """mean over pairs of the two-view iteration loss (gt [B,V,2])."""
    total: Any = 0
    pairs = out["pairs"]
    for (i, j), pd in pairs.items():
        d = dict(pd)
        d["gt_gaze"], d["gt_gaze_1"] = gt[:, i], gt[:, j]
        total = total + iteration_loss(d, iter_decay, rel_weight, reference_decay)
    return total / len(pairs)


# --------------------------------------------------------------------------------------------
# stereo pair index (integer path, bit-exact)
# --------------------------------------------------------------------------------------------
CAMERA_TAGS = {
    "all": list(range(18)),
    "novel_test": list(range(2, 18, 3)),
    "novel_train": [c for c in range(18) if c not in range(2, 18, 3)],
}


class MT19937:
    """Mersenne Twister restated from the published algorithm (Matsumoto & Nishimura 1998) with
    CPython 3.10's seeding (``random.seed(int)`` = init_by_array over the 32-bit words of |seed|)
    and ``Random._randbelow_with_getrandbits`` rejection sampling - what ``random.choice`` runs.
    Used by /root/reference/dataset/gaze.py:72 through Python's global ``random``."""

    N, M = 624, 397

    def __init__(self, seed: int):
        key = []
        s = abs(int(seed))
        while True:
            key.append(s & 0xFFFFFFFF)
            s >>= 32
            if s == 0:
                break
        mt = [0] * self.N
        mt[0] = 19650218
        for i in range(1, self.N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        i, j = 1, 0
        for _ in range(max(self.N, len(key))):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525)) + key[j] + j) & 0xFFFFFFFF
            i += 1
            j += 1
            if i >= self.N:
                mt[0] = mt[self.N - 1]
                i = 1
            if j >= len(key):
                j = 0
        for _ in range(self.N - 1):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941)) - i) & 0xFFFFFFFF
            i += 1
            if i >= self.N:
                mt[0] = mt[self.N - 1]
                i = 1
        mt[0] = 0x80000000
        self.mt, self.idx = mt, self.N

    def u32(self) -> int:
        if self.idx >= self.N:
            mt, N, M = self.mt, self.N, self.M
            for k in range(N):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % N] & 0x7FFFFFFF)
                mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (0x9908B0DF if (y & 1) else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def randbelow(self, n: int) -> int:
        k = n.bit_length()          # n <= 17 here, so k <= 32: one word per draw
        r = self.u32() >> (32 - k)
        while r >= n:
            r = self.u32() >> (32 - k)
        return r


def build_pair_index(file_rows: Sequence[int], camera_tag: str, rng: MT19937) -> List[Tuple[int, int, int]]:
    """(file, idx, partner) tuples - /root/reference/dataset/gaze.py:39-73.  O(n) restatement of
    the reference's O(n^2) list-membership loop; ``rng`` carries state across calls because the
    reference builds the train and test sets from one global stream (main.py:130-147)."""
    cams = CAMERA_TAGS[camera_tag]
    camset = set(cams)
    out: List[Tuple[int, int, int]] = []
    for fi, n in enumerate(file_rows):
        for idx in range(n):
            if idx % 18 not in camset:
                continue
            start = (idx // 18) * 18
            cand = [i for i in range(start, min(start + 18, n)) if (i % 18) in camset and i != idx]
            if cand:
                out.append((fi, idx, cand[rng.randbelow(len(cand))]))
    return out


# --------------------------------------------------------------------------------------------
# input pipeline: ToTensor -> Resize((S, S), antialias=True) -> Normalize   (main.py:38-56)
# --------------------------------------------------------------------------------------------
def _aa_axis_weights(in_size: int, out_size: int):
    """Per output index: (first input index, weights) of the antialiased triangle filter that
    torchvision.transforms.Resize(antialias=True) applies to float tensors.  torchvision is not
    installed here: its tensor path (transforms/_functional_tensor.py: resize) is one call to
    torch.nn.functional.interpolate(mode='bilinear', antialias=True, align_corners=False), i.e. ATen
    `_upsample_bilinear2d_aa` (aten/src/ATen/native/cpu/UpSampleKernel.cpp,
    `_compute_indices_min_size_weights_aa`), restated here in float32 and pinned against that op's
    outputs (tests/golden/resize_aa.npz, generated by make_golden.py --resize)."""
    f32 = np.float32
    scale = f32(in_size) / f32(out_size)
    support = scale if scale >= 1.0 else f32(1.0)
    invscale = f32(1.0) / scale if scale >= 1.0 else f32(1.0)
    max_k = int(math.ceil(float(support))) * 2 + 1
    out = []
    for i in range(out_size):
        center = scale * f32(i + 0.5)
        xmin = max(int(center - support + f32(0.5)), 0)
        xsize = min(int(center + support + f32(0.5)), in_size) - xmin
        xsize = min(max(xsize, 0), max_k)
        w = np.zeros(xsize, dtype=f32)
        for j in range(xsize):
            t = (f32(j + xmin) - center + f32(0.5)) * invscale
            w[j] = max(f32(0.0), f32(1.0) - abs(t))
        tot = f32(0.0)
        for j in range(xsize):
            tot = f32(tot + w[j])
        if tot != 0.0:
            w = (w / tot).astype(f32)
        out.append((xmin, w))
    return out


def resize_bilinear_aa(x: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """x [..., H, W] float32 -> [..., oh, ow]; width pass first, then height (the CPU kernel's order)."""
    x = np.asarray(x, dtype=np.float32)
    H, W = x.shape[-2:]
    if (H, W) == (oh, ow):
        return x.copy()                                  # torchvision returns the input unchanged
    wx = _aa_axis_weights(W, ow)
    tmp = np.zeros(x.shape[:-1] + (ow,), dtype=np.float32)
    for o, (x0, w) in enumerate(wx):
        acc = np.zeros(x.shape[:-1], dtype=np.float32)
        for j in range(len(w)):
            acc = (acc + w[j] * x[..., x0 + j]).astype(np.float32)
        tmp[..., o] = acc
    wy = _aa_axis_weights(H, oh)
    out = np.zeros(x.shape[:-2] + (oh, ow), dtype=np.float32)
    for o, (y0, w) in enumerate(wy):
        acc = np.zeros(tmp.shape[:-2] + (ow,), dtype=np.float32)
        for j in range(len(w)):
            acc = (acc + w[j] * tmp[..., y0 + j, :]).astype(np.float32)
        out[..., o, :] = acc
    return out


def preprocess_u8(img_u8: np.ndarray, size: int, mean, std, swap_rb: bool = False) -> np.ndarray:
    """uint8 [N,H,W,3] -> float32 [N,3,size,size]: (BGR->RGB, dataset/gaze.py:108-109) -> ToTensor
    (/255, CHW) -> Resize((size, size), antialias=True) -> Normalize(mean, std)   (main.py:50-55)."""
    x = np.asarray(img_u8)
    if swap_rb:
        x = x[..., ::-1]
    x = (x.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2)
    x = resize_bilinear_aa(x, size, size)
    m = np.asarray(mean, dtype=np.float32).reshape(1, 3, 1, 1)
    s = np.asarray(std, dtype=np.float32).reshape(1, 3, 1, 1)
    return ((x - m) / s).astype(np.float32)
