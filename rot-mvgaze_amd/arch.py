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


def head_layers(depth: int, num_iter: int) -> List[Tuple[str, int, int]]:
    """(state_dict prefix, in_features, out_features) of every Linear on the path."""
    fc_dim = backbone_spec(depth).fc_dim
    k_in = fc_dim + ROT_DIM
    out = [
        ("_lifter._lifter.blocks.0.0", fc_dim, ROT_DIM),
        ("_lifter._lifter.blocks.1.0", ROT_DIM, ROT_DIM),
    ]
    for i in range(num_iter):
        out.append((f"_img_fusers.{i}._fuser.blocks.0.0", k_in, k_in))
        out.append((f"_img_fusers.{i}._fuser.blocks.1.0", k_in, ROT_DIM))
    for i in range(num_iter):
        out.append((f"_gaze_estimators.{i}.blocks.0.0", k_in, 512))
        out.append((f"_gaze_estimators.{i}.blocks.1.0", 512, 2))
    return out


def state_dict_shapes(depth: int, num_iter: int = 3):
    """Ordered (name, shape, kind) for the default FeatRotationSymm variant.

    kind in {"conv", "bn_weight", "bn_bias", "bn_mean", "bn_var", "bn_count",
    "lin_weight", "lin_bias"}.  Includes the never-used ``fc`` of the torchvision-style
    backbone (resnet.py:201), which is part of the checkpoint contract.
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
    for name, fin, fout in head_layers(depth, num_iter):
        out.append((name + ".weight", (fout, fin), "lin_weight"))
        out.append((name + ".bias", (fout,), "lin_bias"))
    return out
