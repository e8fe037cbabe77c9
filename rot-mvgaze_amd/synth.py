"""Deterministic synthetic weights and inputs (counter-based, numpy only).

The reference's only weight initialisation on the hot path is an ImageNet download
(/root/reference/models/resnet.py:278-283), unavailable offline; parity fixtures and the
benchmark therefore use seeded random init with the reference's *distributions*
(kaiming-normal fan_out convs, resnet.py:203-208; default ``nn.Linear`` init) produced by a
counter-based generator that does not depend on torch's RNG stream, so the golden generator,
the oracle, the tests and the benchmark all rebuild bit-identical tensors from
``(depth, seed)`` alone.  SURVEY.md §7 "hard part 1", §8(d) "Synthetic inputs".
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

from .arch import state_dict_shapes

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream_key(seed: int, tag: str) -> np.uint64:
    h = zlib.crc32(tag.encode()) & 0xFFFFFFFF
    with np.errstate(over="ignore"):
        return _splitmix64(np.array([(int(seed) << 32) ^ h], dtype=np.uint64))[0]


def uniform01(n: int, seed: int, tag: str, lane: int = 0) -> np.ndarray:
    """n doubles in (0,1): element i depends only on (seed, tag, lane, i)."""
    key = _stream_key(seed, tag)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(lane)
        bits = _splitmix64(ctr ^ key)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(n: int, seed: int, tag: str) -> np.ndarray:
    """n standard normals (Box-Muller on two independent counter lanes), float64."""
    u1 = uniform01(n, seed, tag, 0)
    u2 = uniform01(n, seed, tag, 1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


CONDITIONED_RESIDUAL_GAMMA = 0.1


def make_state_dict(depth: int, seed: int = 0, num_iter: int = 3, perturb_bn: bool = False, variant=None,
                    conditioned: bool = False):
    """name -> numpy array, keys/shapes = the reference checkpoint contract.

    perturb_bn=True draws BN gamma in U(0.5,1.5), beta in U(-0.2,0.2), running_mean in
    U(-0.1,0.1), running_var in U(0.5,1.5) so that affine/running-stat paths are exercised by
    parity tests; False gives the reference's init (gamma=1, beta=0, mean=0, var=1).

    conditioned=True is the second, WELL-CONDITIONED recipe: the same tensors, with the gamma of the last
    BatchNorm of every residual block (bn2 of a BasicBlock, bn3 of a Bottleneck - the one torchvision's
    ``zero_init_residual`` zeroes, resnet.py:213-218 of the reference) scaled by 0.1, so that each block is a
    small correction of its identity branch, as in a trained network.  A randomly initialised ResNet-50 without it
    amplifies a 1e-5 relative input perturbation (or one flipped bf16 rounding) into a 10-20 % change of
    its predictions under bf16 storage; with it the same perturbation moves them by 2-6e-3
    (tests/test_oracle_golden.py::test_conditioned_recipe_is_well_conditioned measures that on the CPU oracle
    alone), which is what lets tests/test_bf16_gpu.py hold ResNet-50 end to end to its declared 3e-2.
    """
    sd = OrderedDict()
    from .arch import DEFAULT_VARIANT
    variant = variant or DEFAULT_VARIANT
    shapes = state_dict_shapes(depth, num_iter, variant)
    damped = set()
    if conditioned:
        from .arch import backbone_spec
        damped = set(blk.convs[-1].bn + ".weight" for blk in backbone_spec(depth).blocks)
    fan_ins = dict((nm, sh) for nm, sh, _ in shapes)
    for name, shape, kind in shapes:
        n = int(np.prod(shape)) if len(shape) else 1
        if kind == "conv":
            cout, _cin, kh, kw = shape
            std = np.sqrt(2.0 / (cout * kh * kw))      # kaiming_normal_(mode="fan_out", relu)
            a = (normal(n, seed, name) * std).astype(np.float32).reshape(shape)
        elif kind == "lin_weight":
            bound = 1.0 / np.sqrt(shape[1])            # nn.Linear default: U(-1/sqrt(fan_in), +)
            a = ((uniform01(n, seed, name) * 2.0 - 1.0) * bound).astype(np.float32).reshape(shape)
        elif kind == "lin_bias":
            fan_in = fan_ins[name[: -len("bias")] + "weight"][1]
            bound = 1.0 / np.sqrt(fan_in)
            a = ((uniform01(n, seed, name) * 2.0 - 1.0) * bound).astype(np.float32)
        elif kind == "bn_weight":
            a = (0.5 + uniform01(n, seed, name)).astype(np.float32) if perturb_bn else np.ones(n, np.float32)
            if name in damped:
                a = (a * np.float32(CONDITIONED_RESIDUAL_GAMMA)).astype(np.float32)
        elif kind == "bn_bias":
            a = ((uniform01(n, seed, name) - 0.5) * 0.4).astype(np.float32) if perturb_bn else np.zeros(n, np.float32)
        elif kind == "bn_mean":
            a = ((uniform01(n, seed, name) - 0.5) * 0.2).astype(np.float32) if perturb_bn else np.zeros(n, np.float32)
        elif kind == "bn_var":
            a = (0.5 + uniform01(n, seed, name)).astype(np.float32) if perturb_bn else np.ones(n, np.float32)
        elif kind == "bn_count":
            a = np.array(0, dtype=np.int64)
        elif kind == "ibn_mean":          # IntensityBatchNorm.running_mean, init ones (rot_mv.py:16)
            a = ((0.5 + uniform01(n, seed, name)) if perturb_bn else np.ones(n)).astype(np.float32).reshape(shape)
        elif kind.startswith("alias:"):   # share_weights: the same array under a second name
            a = sd[kind[len("alias:"):]]
        else:
            raise AssertionError(kind)
        sd[name] = a
    return sd


def make_inputs(batch: int, views: int, seed: int = 1234, hw: int = 224):
    """Synthetic batch of SURVEY §8(d): img ~ N(0,1) [B,V,3,hw,hw]; head pose and gaze (pitch,yaw)
    ~ U(-0.5,0.5) rad [B,V,2].  float32."""
    n_img = batch * views * 3 * hw * hw
    img = normal(n_img, seed, "img").astype(np.float32).reshape(batch, views, 3, hw, hw)
    hp = (uniform01(batch * views * 2, seed, "head_pose") - 0.5).astype(np.float32).reshape(batch, views, 2)
    gz = (uniform01(batch * views * 2, seed, "gt_gaze") - 0.5).astype(np.float32).reshape(batch, views, 2)
    return {"img": img, "head_pose": hp, "gt_gaze": gz}
