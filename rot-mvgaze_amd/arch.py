"""Architecture description of the hot-path network (shapes + state_dict names).

Shared by the product modules and (read-only) by the oracle.  Written from the structure of
the reference, not copied from it:

* ResNet-18 / ResNet-50 v1.5 layer plan ........ /root/reference/models/resnet.py:151-275
  (BasicBlock :50-96, Bottleneck :99-148 with the stride on the 3x3)
* feature extractor / lifter / fusers / heads ... /root/reference/models/rot_mv.py:113-184
* ``Mlp`` = Linear(+ReLU) ... Linear, the ``.blocks.N.0`` nesting
  ................................................ /root/reference/models/backbones/blocks.py:27-82
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

NUM_FEAT_VEC = 512          # rot_mv.py:117
ROT_DIM = 3 * NUM_FEAT_VEC  # 1536


@dataclass
class ConvSpec:
    name: str               # state_dict prefix of the conv weight (without ".weight")
    bn: str                 # state_dict prefix of the BatchNorm that follows it
    cin: int
    cout: int
    k: int
    stride: int
    pad: int


@dataclass
class BlockSpec:
    prefix: str
    convs: List[ConvSpec]
    downsample: Optional[ConvSpec]


@dataclass
class BackboneSpec:
    depth: int
    stem: ConvSpec
    blocks: List[BlockSpec] = field(default_factory=list)
    fc_dim: int = 512

    def all_convs(self) -> List[ConvSpec]:
        out = [self.stem]
        for b in self.blocks:
            out.extend(b.convs)
            if b.downsample is not None:
                out.append(b.downsample)
        return out


def backbone_spec(depth: int, prefix: str = "_feat_extractor.0.") -> BackboneSpec:
    """Layer plan of ResNet-18 / -50 (resnet.py:284-313: [2,2,2,2] basic / [3,4,6,3] bottleneck)."""
    if depth == 18:
        counts, bottleneck, expansion = [2, 2, 2, 2], False, 1
    elif depth == 50:
        counts, bottleneck, expansion = [3, 4, 6, 3], True, 4
    else:
        raise ValueError("hot path covers backbone_depth 18 and 50 only (rot_mv.py:119-122)")
    spec = BackboneSpec(depth, ConvSpec(prefix + "conv1", prefix + "bn1", 3, 64, 7, 2, 3),
                        fc_dim=512 * expansion)
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], counts), start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 1) else 1
            p = f"{prefix}layer{li}.{bi}."
            outplanes = planes * expansion
            if bottleneck:
                convs = [
                    ConvSpec(p + "conv1", p + "bn1", inplanes, planes, 1, 1, 0),
                    ConvSpec(p + "conv2", p + "bn2", planes, planes, 3, stride, 1),
                    ConvSpec(p + "conv3", p + "bn3", planes, outplanes, 1, 1, 0),
                ]
            else:
                convs = [
                    ConvSpec(p + "conv1", p + "bn1", inplanes, planes, 3, stride, 1),
                    ConvSpec(p + "conv2", p + "bn2", planes, planes, 3, 1, 1),
                ]
            ds = None
            if stride != 1 or inplanes != outplanes:
                ds = ConvSpec(p + "downsample.0", p + "downsample.1", inplanes, outplanes, 1, stride, 0)
            spec.blocks.append(BlockSpec(p, convs, ds))
            inplanes = outplanes
    return spec


def mlp_names(prefix: str, n_layers: int) -> List[str]:
    return [f"{prefix}blocks.{i}.0" for i in range(n_layers)]


@dataclass(frozen=True)
class Variant:
    """Constructor flags of FeatRotationSymm (/root/reference/models/rot_mv.py:102-184).  All False =
    the variant main.py builds; the others are the paper's ablations."""
    share_weights: bool = False      # one fuser / head module reused by every iteration (:148-156)
    encode_rotmat: bool = False      # ImageRotmatFeatFuser: rotation matrix concatenated, not applied (:53-69)
    share_feature: bool = False      # RotFeatFuser + IntensityBatchNorm on the lifted features (:72-86,157-169)
    ignore_rotmat: bool = False      # ImageFeatFuser on the partner's feature without rotating it (:219-225)

    def check(self) -> "Variant":
        assert not (self.ignore_rotmat and self.encode_rotmat)                       # rot_mv.py:133
        if self.share_feature and (self.share_weights or self.encode_rotmat or self.ignore_rotmat):
            # share_weights builds image-feature fusers (:148-156) that the share_feature forward then
            # feeds lifted features (:199-201): the reference itself fails on the shapes.  With
            # encode/ignore the fuser call passes a third argument / skips the rotation RotFeatFuser
            # does not support (:219-233).
            raise ValueError("share_feature cannot be combined with share_weights / encode_rotmat / ignore_rotmat "
                             "(the reference model raises on these combinations)")
        return self

    @property
    def default(self) -> bool:
        return not (self.share_weights or self.encode_rotmat or self.share_feature or self.ignore_rotmat)


DEFAULT_VARIANT = Variant()


def head_dims(depth: int, variant: Variant = DEFAULT_VARIANT):
    """(fuser input width, fuser Mlp widths, gaze-head input width) of a variant."""
    fc_dim = backbone_spec(depth).fc_dim
    if variant.share_feature:
        k = 6 * NUM_FEAT_VEC                                   # RotFeatFuser (:72-80); head :163
        return k, [k, k, ROT_DIM], k
    k_in = fc_dim + ROT_DIM
    if variant.encode_rotmat:
        return k_in + 9, [k_in + 9, k_in + 9, ROT_DIM], k_in   # ImageRotmatFeatFuser (:53-60)
    return k_in, [k_in, ROT_DIM], k_in                          # ImageFeatFuser (:35-43)


def head_layers(depth: int, num_iter: int, variant: Variant = DEFAULT_VARIANT) -> List[Tuple[str, int, int]]:
    """(state_dict prefix, in_features, out_features) of every Linear on the path, in state_dict
    order.  With share_weights the entries of iterations 1.. name the same tensors as iteration 0
    (``nn.ModuleList([module] * num_iter)``)."""
    fc_dim = backbone_spec(depth).fc_dim
    fin, widths, hin = head_dims(depth, variant)
    out = [
        ("_lifter._lifter.blocks.0.0", fc_dim, ROT_DIM),
        ("_lifter._lifter.blocks.1.0", ROT_DIM, ROT_DIM),
    ]
    for i in range(num_iter):
        d = fin
        for l, w in enumerate(widths):
            out.append((f"_img_fusers.{i}._fuser.blocks.{l}.0", d, w))
            d = w
    for i in range(num_iter):
        out.append((f"_gaze_estimators.{i}.blocks.0.0", hin, 512))
        out.append((f"_gaze_estimators.{i}.blocks.1.0", 512, 2))
    return out


def state_dict_shapes(depth: int, num_iter: int = 3, variant: Variant = DEFAULT_VARIANT):
    """Ordered (name, shape, kind) of the model's state_dict.

    kind in {"conv", "bn_weight", "bn_bias", "bn_mean", "bn_var", "bn_count", "lin_weight",
    "lin_bias", "ibn_mean"} or "alias:<name>" (share_weights: the same tensor under a second
    name).  Includes the never-used ``fc`` of the torchvision-style backbone (resnet.py:201), which
    is part of the checkpoint contract.
    """
    spec = backbone_spec(depth)
    out = []

    def add_conv(c: ConvSpec):
        out.append((c.name + ".weight", (c.cout, c.cin, c.k, c.k), "conv"))
        out.append((c.bn + ".weight", (c.cout,), "bn_weight"))
        out.append((c.bn + ".bias", (c.cout,), "bn_bias"))
        out.append((c.bn + ".running_mean", (c.cout,), "bn_mean"))
        out.append((c.bn + ".running_var", (c.cout,), "bn_var"))
        out.append((c.bn + ".num_batches_tracked", (), "bn_count"))

    add_conv(spec.stem)
    for b in spec.blocks:
        for c in b.convs:
            add_conv(c)
        if b.downsample is not None:
            add_conv(b.downsample)
    out.append(("_feat_extractor.0.fc.weight", (1000, spec.fc_dim), "lin_weight"))
    out.append(("_feat_extractor.0.fc.bias", (1000,), "lin_bias"))

    def shared(name: str) -> Optional[str]:
        """share_weights: iteration i > 0 of the fusers / heads is iteration 0's module."""
        if not variant.share_weights:
            return None
        for pre in ("_img_fusers.", "_gaze_estimators."):
            if name.startswith(pre):
                idx, rest = name[len(pre):].split(".", 1)
                if idx != "0":
                    return pre + "0." + rest
        return None

    seen_ibn = set()
    for name, fin, fout in head_layers(depth, num_iter, variant):
        if variant.share_feature and name.startswith("_img_fusers.") and name.endswith("blocks.0.0"):
            # module order inside RotFeatFuser: _fuser (parameters) then _batchnorm (buffer); state_dict
            # lists a module's own tensors before its children, children in registration order
            pass
        for suffix, shape, kind in ((".weight", (fout, fin), "lin_weight"), (".bias", (fout,), "lin_bias")):
            tgt = shared(name + suffix)
            out.append((name + suffix, shape, kind if tgt is None else "alias:" + tgt))
        if variant.share_feature and name.startswith("_img_fusers.") and name.endswith("blocks.2.0"):
            i = name.split(".")[1]
            if i not in seen_ibn:
                seen_ibn.add(i)
                out.append((f"_img_fusers.{i}._batchnorm.running_mean", (1, 1, NUM_FEAT_VEC), "ibn_mean"))
    return out
