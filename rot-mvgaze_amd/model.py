"""Drop-in ``FeatRotationSymm`` on the MI355X kernels.

Mirrors the public surface of /root/reference/models/rot_mv.py:102-269 - same constructor
signature (:103-111), ``forward(data: dict) -> dict`` that reads ``img_0/img_1/rot_0/rot_1``
(:188-191), updates the caller's dict in place and returns it (:266-269) with the same keys, and
the same ``state_dict`` key names/shapes so reference checkpoints load with
``strict=True``.  The arithmetic is done by librotmvgaze_hip.so; autograd sees two nodes
(backbone, fusion head) whose backward runs the hand-written backward kernels.

Parameter gradients are written by the kernels straight into one flat *gradient arena* laid out
in grad-ready order (heads/fusers iter I-1..0, lifter, layer4 ... stem) and ``param.grad`` is set to
a view of it: no per-parameter copies, and the data-parallel reducer all-reduces arena slices.
"""
from __future__ import annotations

import weakref
from typing import Any, Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .arch import DEFAULT_VARIANT, NUM_FEAT_VEC, Variant, backbone_spec, state_dict_shapes
from .backbone import Backbone, GradSink
from .heads import FusionHead, directed_pairs

Tensor = torch.Tensor


class _Node(nn.Module):
    """Pure parameter container (the compute lives in Backbone / FusionHead)."""


def _init_tensor(shape, kind: str) -> Tensor:
    """Random init with the reference's distributions (resnet.py:203-208; nn.Linear defaults).
    The reference then loads ImageNet weights from a URL (resnet.py:278-283) - unavailable offline;
    load a checkpoint with ``load_state_dict`` instead."""
    if kind == "conv":
        t = torch.empty(shape)
        nn.init.kaiming_normal_(t, mode="fan_out", nonlinearity="relu")
        return t.contiguous(memory_format=torch.channels_last)
    if kind == "lin_weight":
        t = torch.empty(shape)
        nn.init.kaiming_uniform_(t, a=5 ** 0.5)
        return t
    if kind == "lin_bias":
        return torch.empty(shape)          # filled by the caller (needs fan_in)
    if kind in ("bn_weight", "bn_var"):
        return torch.ones(shape)
    if kind in ("bn_bias", "bn_mean"):
        return torch.zeros(shape)
    if kind == "bn_count":
        return torch.tensor(0, dtype=torch.long)
    raise AssertionError(kind)


class _ArenaSink(GradSink):
    def __init__(self, model: "MultiViewGaze"):
        self.m = model
        self._acc: Dict[int, bool] = {}
        self._foreign: Dict[int, bool] = {}
        self.active = False

    def begin(self):
        """Decide, per parameter, whether this backward accumulates into an existing .grad."""
        if self.active:
            return
        self.active = True
        self._acc.clear()
        self._foreign.clear()
        for p in self.m._grad_order:
            v = self.m._grad_views[id(p)]
            if p.grad is None:
                self._acc[id(p)] = False
            elif p.grad.data_ptr() == v.data_ptr():
                self._acc[id(p)] = True                      # kernels accumulate in place
            else:
                self._acc[id(p)] = False                     # foreign .grad: compute fresh, add at publish
                self._foreign[id(p)] = True

    def view(self, p):
        return self.m._grad_views[id(p)]

    def accumulate(self, p):
        return self._acc[id(p)]

    def publish(self, ps):
        for p in ps:
            v = self.m._grad_views[id(p)]
            if self._foreign.get(id(p)):
                p.grad.add_(v)
            elif p.grad is None:
                p.grad = v
        hook = self.m._on_grads_ready
        if hook is not None:
            hook(ps)


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: "MultiViewGaze", training: bool, n_views: int, *tensors: Tensor):
        imgs = list(tensors[:n_views])
        ctx.set_materialize_grads(False)
        # needs_input_grad reports requires_grad of the inputs whatever the grad mode is (and grad mode is always off
        # inside forward): run_views records the caller's grad mode, so that torch.no_grad() inference takes the
        # tape-less path (BatchNorm folded into the conv epilogues) even though the parameters require grad
        keep = model._grad_mode and any(ctx.needs_input_grad)
        ctx.need_dimg = any(ctx.needs_input_grad[3:3 + n_views])
        feat, tape = model._backbone.forward(imgs, training, keep, model.input_bgr, model.input_size, need_dimg=ctx.need_dimg)
        ctx.model, ctx.tape, ctx.n_views = model, tape, n_views
        if model._debug_keep_tapes:
            model._last_backbone_tape = tape
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        model, tape = ctx.model, ctx.tape
        n_in = ctx.n_views + len(model._backbone_params)
        if tape is None:
            raise RuntimeError("FeatRotationSymm: backward through the backbone a second time - the saved activations "
                               "were released by the first backward (retain_graph is not supported; run forward again)")
        ctx.tape = None
        if dfeat is None:                         # nothing upstream depends on the image features
            model._sink.active = False
            model._finish_backward()
            return (None, None, None) + (None,) * n_in
        model._sink.begin()
        dimgs = model._backbone.backward(tape, dfeat.contiguous(), model._sink, ctx.need_dimg)
        model._sink.active = False
        model._finish_backward()
        img_grads = tuple(dimgs) if dimgs is not None else (None,) * ctx.n_views
        return (None, None, None) + img_grads + (None,) * len(model._backbone_params)


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: "MultiViewGaze", img_feat: Tensor, rot: Tensor, *params: Tensor):
        ctx.set_materialize_grads(False)
        keep = model._grad_mode and any(ctx.needs_input_grad)
        lifted, feats, preds, tape = model._head.forward(img_feat.detach().contiguous(), rot, keep, model.training)
        ctx.model, ctx.tape = model, tape
        if model._debug_keep_tapes:
            model._last_head_tape = tape
        return lifted, feats, preds

    @staticmethod
    def backward(ctx, d_lifted, d_feats, d_preds):
        model, tape = ctx.model, ctx.tape
        nparam = len(model._head_params)
        if tape is None:
            raise RuntimeError("FeatRotationSymm: backward through the fusion head a second time - the saved "
                               "activations were released by the first backward (retain_graph is not supported; "
                               "run forward again)")
        ctx.tape = None
        model._sink.begin()                       # the head's backward is the first node to run
        bb = model._backbone
        side = bb._side(tape["img_feat"].device) if (bb.overlap_wgrad and bb.overlap_head and tape["img_feat"].is_cuda) else None
        dimg = model._head.backward(tape, d_lifted, d_feats, d_preds, model._sink, side)
        if not ctx.needs_input_grad[1]:                        # no backbone backward will follow
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            model._sink.active = False
            model._finish_backward()
        return (None, dimg, None) + (None,) * nparam


class MultiViewGaze(nn.Module):
    """V-view engine (V >= 2).  ``FeatRotationSymm`` below is its two-view, reference-shaped face."""

    def __init__(self, backbone_depth: int = 50, num_iter: int = 3, variant: Variant = DEFAULT_VARIANT) -> None:
        super().__init__()
        if backbone_depth not in (18, 50):
            raise ValueError("backbone_depth must be 18 or 50 (rot_mv.py:119-122)")
        self._variant = variant.check()
        self._num_iter = num_iter
        self._output_index = num_iter - 1
        self._num_feat_vec = NUM_FEAT_VEC
        self._depth = backbone_depth
        self._fc_dim = backbone_spec(backbone_depth).fc_dim
        shapes = state_dict_shapes(backbone_depth, num_iter, self._variant)
        fan_in = {n: s[1] for n, s, k in shapes if k == "lin_weight"}
        made: Dict[str, nn.Parameter] = {}
        for name, shape, kind in shapes:
            node: nn.Module = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            if kind.startswith("alias:"):             # share_weights: the same Parameter under a second name
                node.register_parameter(parts[-1], made[kind[len("alias:"):]])
                continue
            if kind == "ibn_mean":                    # IntensityBatchNorm.running_mean (rot_mv.py:16)
                node.register_buffer(parts[-1], torch.ones(shape))
                continue
            t = _init_tensor(shape, kind)
            if kind == "lin_bias":
                bound = 1.0 / fan_in[name[:-4] + "weight"] ** 0.5
                nn.init.uniform_(t, -bound, bound)
            if kind in ("bn_mean", "bn_var", "bn_count"):
                node.register_buffer(parts[-1], t)
            else:
                made[name] = nn.Parameter(t)
                node.register_parameter(parts[-1], made[name])
        owner = weakref.ref(self)
        for prm in self.parameters():
            prm._mvg_owner = owner            # lets rot_mvgaze_amd.optim.Adam find the arenas
        self._on_grads_ready: Optional[Callable[[List[nn.Parameter]], None]] = None
        self._on_backward_done: Optional[Callable[[], None]] = None
        self._layout_sig = None
        self._debug_keep_tapes = False            # tests: keep references to the saved activations
        self._grad_mode = True                    # the caller's grad mode at the last forward (see _BackboneFn.forward)
        self.input_bgr = False                    # raw uint8 inputs: swap B and R first (dataset color_type 'bgr')
        self.input_size: Optional[int] = 224      # raw uint8 inputs: Resize((S, S), antialias=True), main.py:40,53; None = keep
        self._sink = _ArenaSink(self)
        self._wgrad_low_priority = True           # data-parallel runs set False (dp.GradAllReducer._configure)
        # storage / matrix-core type of the backbone: torch.float32 (default: the path held to 1e-4 against the
        # reference) or torch.bfloat16 (BASELINE config C5: bf16 activations + weights copies, fp32 everything else)
        self.compute_dtype = torch.float32

    # ---------------------------------------------------------------- plumbing
    def _named_tensors(self) -> Dict[str, Tensor]:
        d: Dict[str, Tensor] = dict(self.named_parameters(remove_duplicate=False))
        d.update(dict(self.named_buffers()))
        return d

    def _ensure_layout(self, dev: torch.device) -> None:
        """(Re)build the engine views over the live parameters: KRSC conv weights, gradient arena."""
        first = next(self.parameters())
        sig = (first.data_ptr(), str(first.device), sum(1 for _ in self.parameters()))
        if sig == self._layout_sig:
            return
        if not first.is_cuda:
            raise RuntimeError("FeatRotationSymm (MI355X build) needs its parameters on the GPU: model.to('cuda'). "
                               "There is no CPU fallback.")
        named = self._named_tensors()
        for name, p in self.named_parameters():
            if p.dim() == 4 and not p.is_contiguous(memory_format=torch.channels_last):
                p.data = p.data.contiguous(memory_format=torch.channels_last)
            elif p.dim() != 4 and not p.is_contiguous():
                p.data = p.data.contiguous()
            if p.dtype != torch.float32:
                raise RuntimeError("fp32 parameters only")
        self._backbone = Backbone(self._depth, named)
        self._backbone.wgrad_low_priority = self._wgrad_low_priority
        self._backbone.act_dtype = self.compute_dtype
        self._grad_streams = self._backbone.grad_streams   # streams besides the caller's that write gradients
        self._head = FusionHead(named, self._fc_dim, self._num_iter, self._variant)
        # grad-ready order: heads+fusers I-1..0 (shared weights: once), lifter, backbone blocks last..first, stem
        order: List[nn.Parameter] = []
        for m in self._head.unique_modules():
            order += m.parameters()
        order += self._head.lifter.parameters()
        self._head_params = list(order)
        bb: List[nn.Parameter] = []
        spec = self._backbone.spec
        for blk in reversed(spec.blocks):
            cs = [blk.convs[-1]] + list(reversed(blk.convs[:-1])) + ([blk.downsample] if blk.downsample else [])
            for c in cs:
                bb += [named[c.name + ".weight"], named[c.bn + ".weight"], named[c.bn + ".bias"]]
        bb += [named[spec.stem.name + ".weight"], named[spec.stem.bn + ".weight"], named[spec.stem.bn + ".bias"]]
        self._backbone_params = bb
        order += bb
        self._grad_order = order
        # every entry starts on a 16-byte boundary (the kernels move float4 / 16-byte buffer loads; the
        # 2-element head biases would otherwise leave every later slice 8-byte aligned); gaps stay zero in
        # both arenas, so the one-launch optimizer leaves them at zero
        total = sum((p.numel() + 3) // 4 * 4 for p in order)
        self._grad_arena = torch.zeros(total, dtype=torch.float32, device=first.device)
        # parameters live in a second arena with the SAME offsets (one-launch optimizer, flat
        # broadcast / checkpoint staging); each Parameter becomes a view of its slice
        self._param_arena = torch.zeros(total, dtype=torch.float32, device=first.device)
        self._grad_views: Dict[int, Tensor] = {}
        self._grad_offsets: Dict[int, int] = {}

        def view_of(arena, off, p):
            flat = arena[off:off + p.numel()]
            if p.dim() == 4:
                o, i, kh, kw = p.shape
                return flat.view(o, kh, kw, i).permute(0, 3, 1, 2)       # channels_last view of the slice
            return flat.view(p.shape)
        off = 0
        for p in order:
            self._grad_views[id(p)] = view_of(self._grad_arena, off, p)
            self._grad_offsets[id(p)] = off
            pv = view_of(self._param_arena, off, p)
            pv.copy_(p.data)
            p.data = pv
            p.grad = None
            assert off % 4 == 0 and pv.data_ptr() % 16 == 0
            off += (p.numel() + 3) // 4 * 4
        first = next(self.parameters())
        self._layout_sig = (first.data_ptr(), str(first.device), sum(1 for _ in self.parameters()))

    def grad_arena(self):
        """(flat fp32 gradient buffer, [(param, offset, numel)] in grad-ready order)."""
        return self._grad_arena, [(p, self._grad_offsets[id(p)], p.numel()) for p in self._grad_order]

    def invalidate_weight_cache(self) -> None:
        """``torch.no_grad()`` inference keeps sp copies of the conv weights between calls, keyed on every parameter's
        version counter.  The counter misses writes through ``p.data`` (``p.data.copy_`` / ``mul_``: EMA, clipping) and
        writes to ``param_arena()``: call this after such a write (or ``torch.autograd.graph.increment_version(p)``).
        ``load_state_dict``, the fused Adam, the data-parallel broadcast, ``train()`` and training forwards are covered."""
        bb = getattr(self, "_backbone", None)
        if bb is not None:
            bb.invalidate_weight_cache()

    def train(self, mode: bool = True):
        if mode:                                   # weights are about to change; eval() keeps the cache (it is what uses it)
            self.invalidate_weight_cache()
        return super().train(mode)

    def param_arena(self) -> Tensor:
        """Flat fp32 buffer holding every trainable parameter at the offsets of ``grad_arena``."""
        return self._param_arena

    def ensure_layout(self) -> None:
        self._ensure_layout(next(self.parameters()).device)

    def _finish_backward(self):
        if self._on_backward_done is not None:
            self._on_backward_done()

    # ---------------------------------------------------------------- compute
    def run_views(self, imgs: List[Tensor], rot: Tensor):
        """imgs: V tensors [B,3,H,W]; rot [B,V,3,3] -> (img_feat [V,B,Cf], lifted [V,B,3,512],
        feats [I,D,B,3,512], preds [I,D,B,2]) with D = V(V-1) directed pairs (heads.directed_pairs)."""
        dev = imgs[0].device
        if not imgs[0].is_cuda:
            raise RuntimeError("inputs must be device tensors (no CPU fallback)")
        self._ensure_layout(dev)
        if self.compute_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 or torch.bfloat16")
        self._backbone.act_dtype = self.compute_dtype
        self._head.mixed = self.compute_dtype == torch.bfloat16
        self._head.split = self._backbone.split          # one switch (MVG_SPLIT) selects the kernel family everywhere
        self._sink.active = False
        self._grad_mode = torch.is_grad_enabled()
        img_feat = _BackboneFn.apply(self, self.training, len(imgs), *imgs, *self._backbone_params)
        lifted, feats, preds = _HeadFn.apply(self, img_feat, rot, *self._head_params)
        return img_feat, lifted, feats, preds

    def forward_multiview(self, img: Tensor, rot: Tensor) -> Dict[str, Any]:
        """img [B,V,3,H,W] (or a list of V tensors [B,3,H,W]), rot [B,V,3,3].  Output: per-view ``img_feat`` /
        ``initial_rot_feat`` and per-pair dicts shaped like the reference's two-view output."""
        if isinstance(img, (list, tuple)):
            imgs = list(img)                       # V tensors [B,3,H,W] (already one tensor per view)
            V = len(imgs)
        else:
            V = img.shape[1]
            imgs = [img[:, v] for v in range(V)]
        img_feat, lifted, feats, preds = self.run_views(imgs, rot)
        out: Dict[str, Any] = {"num_iter": self._num_iter, "views": V, "img_feat": img_feat,
                               "initial_rot_feat": lifted, "pairs": {}, "_mvg_preds": preds}
        p = 0
        for i in range(V):
            for j in range(i + 1, V):
                pd = {"num_iter": self._num_iter}
                for it in range(self._num_iter):
                    pd[f"iter_{it}"] = {"feat_0": feats[it, 2 * p], "feat_1": feats[it, 2 * p + 1],
                                        "pred_gaze_0": preds[it, 2 * p], "pred_gaze_1": preds[it, 2 * p + 1]}
                out["pairs"][(i, j)] = pd
                p += 1
        out["pred_gaze"] = preds[self._output_index, 0]
        return out


class FeatRotationSymm(MultiViewGaze):
    def __init__(
        self,
        backbone_depth: int = 50,
        num_iter: Optional[int] = None,
        share_weights: bool = False,
        encode_rotmat: bool = False,
        share_feature: bool = False,
        ignore_rotmat: bool = False,
    ) -> None:
        assert not (ignore_rotmat and encode_rotmat)                       # rot_mv.py:133
        super().__init__(backbone_depth, num_iter, Variant(share_weights, encode_rotmat, share_feature, ignore_rotmat))

    def forward(self, data: Dict[str, Any]) -> Dict[str, Any]:
        img_0: Tensor = data["img_0"]
        img_1: Tensor = data["img_1"]
        rot = torch.stack([data["rot_0"], data["rot_1"]], dim=1)          # [B,2,3,3] (16-byte rows: plumbing)
        img_feat, lifted, feats, preds = self.run_views([img_0, img_1], rot)
        pred: Dict[str, Any] = {
            "num_iter": self._num_iter,
            "img_feat_0": img_feat[0], "img_feat_1": img_feat[1],
            "initial_rot_feat_0": lifted[0], "initial_rot_feat_1": lifted[1],
        }
        if self._variant.share_feature:                                    # rot_mv.py:199-201
            pred["img_feat_0"], pred["img_feat_1"] = lifted[0], lifted[1]
        for it in range(self._num_iter):
            pred[f"iter_{it}"] = {"feat_0": feats[it, 0], "feat_1": feats[it, 1],
                                  "pred_gaze_0": preds[it, 0], "pred_gaze_1": preds[it, 1]}
        pred["pred_gaze"] = pred[f"iter_{self._output_index}"]["pred_gaze_0"]
        pred["_mvg_preds"] = preds                                         # fused-loss fast path handle
        data.update(pred)
        return data
