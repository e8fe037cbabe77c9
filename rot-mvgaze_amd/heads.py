"""Lifter + rotation-constrained cross-view fusion + gaze heads, forward and backward, on the HIP
kernels - generalised from the reference's two views to V views (every unordered pair runs the
reference's two-view recurrence).

What it replaces (for V = 2 exactly):
  * Feat3dLifter.forward ............ /root/reference/models/rot_mv.py:91-98
  * rot_10 / rot_01 .................. :193-194
  * the fusion loop .................. :213-263 (ImageFeatFuser.forward :44-50, heads :249-254)

Row space: D = V(V-1) *directed* pairs d = 2p (i<-j), 2p+1 (j<-i) for the p-th unordered pair
(i<j) in lexicographic order; every GEMM of an iteration runs over all D*B rows at once because
the two directions (and all pairs) share the iteration's weights.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import ops
from ._lib import ConvDesc
from .arch import NUM_FEAT_VEC, ROT_DIM
from .backbone import GradSink

Tensor = torch.Tensor


def directed_pairs(views: int) -> Tuple[List[int], List[int]]:
    vi, vj = [], []
    for i in range(views):
        for j in range(i + 1, views):
            vi += [i, j]
            vj += [j, i]
    return vi, vj


class _Mlp2:
    """Linear+ReLU -> Linear (every Mlp on the path has exactly two layers,
    /root/reference/models/backbones/blocks.py:27-60 with rot_mv.py:95,41-43,179-184)."""

    def __init__(self, params: Dict[str, Tensor], prefix: str):
        self.w0, self.b0 = params[prefix + "blocks.0.0.weight"], params[prefix + "blocks.0.0.bias"]
        self.w1, self.b1 = params[prefix + "blocks.1.0.weight"], params[prefix + "blocks.1.0.bias"]
        self.fin, self.hid, self.fout = self.w0.shape[1], self.w0.shape[0], self.w1.shape[0]

    def parameters(self):
        return [self.w0, self.b0, self.w1, self.b1]

    def forward(self, x: Tensor, out: Optional[Tensor] = None):
        rows = x.shape[0]
        dev = x.device
        h = torch.empty(rows, self.hid, dtype=torch.float32, device=dev)
        ops.linear_fprop(x, self.w0.detach(), self.b0.detach(), True, h, rows, self.fin, self.hid)
        y = out if out is not None else torch.empty(rows, self.fout, dtype=torch.float32, device=dev)
        if self.fout <= 4:
            ops.linear_skinny_fwd(h, self.w1.detach(), self.b1.detach(), y, rows, self.hid, self.fout)
        else:
            ops.linear_fprop(h, self.w1.detach(), self.b1.detach(), False, y, rows, self.hid, self.fout)
        return h, y

    def backward(self, x: Tensor, h: Tensor, gy: Tensor, sink: GradSink, dx_addend: Optional[Tensor] = None,
                 dx_out: Optional[Tensor] = None) -> Tensor:
        """gy = grad wrt the output; returns grad wrt x (optionally fused `+ dx_addend`)."""
        rows = x.shape[0]
        dev = x.device
        dh = torch.empty(rows, self.hid, dtype=torch.float32, device=dev)
        if self.fout <= 4:
            acc = sink.accumulate(self.w1)
            ops.linear_skinny_bwd(gy, h, self.w1.detach(), h, dh, sink.view(self.w1), sink.view(self.b1), rows,
                                  self.hid, self.fout, acc)
        else:
            d1 = ConvDesc.linear(rows, self.hid, self.fout)
            ops.linear_dgrad(gy, self.w1.detach(), h, None, dh, rows, self.hid, self.fout)   # (gy @ W1) * (h > 0)
            ops.conv_wgrad(d1, h, gy, sink.view(self.w1), sink.accumulate(self.w1))
            ops.colsum(gy, sink.view(self.b1), rows, self.fout, sink.accumulate(self.b1))
        d0 = ConvDesc.linear(rows, self.fin, self.hid)
        ops.conv_wgrad(d0, x, dh, sink.view(self.w0), sink.accumulate(self.w0))
        ops.colsum(dh, sink.view(self.b0), rows, self.hid, sink.accumulate(self.b0))
        dx = dx_out if dx_out is not None else torch.empty(rows, self.fin, dtype=torch.float32, device=dev)
        ops.linear_dgrad(dh, self.w0.detach(), None, dx_addend, dx, rows, self.fin, self.hid)
        return dx


class FusionHead:
    def __init__(self, params: Dict[str, Tensor], fc_dim: int, num_iter: int):
        self.cf, self.I = fc_dim, num_iter
        self.kin = fc_dim + ROT_DIM
        self.lifter = _Mlp2(params, "_lifter._lifter.")
        self.fusers = [_Mlp2(params, f"_img_fusers.{i}._fuser.") for i in range(num_iter)]
        self.heads = [_Mlp2(params, f"_gaze_estimators.{i}.") for i in range(num_iter)]
        self._idx_cache: Dict[Tuple[int, str], dict] = {}

    def _indices(self, views: int, dev) -> dict:
        key = (views, str(dev))
        if key not in self._idx_cache:
            vi, vj = directed_pairs(views)
            D = len(vi)
            mk = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)
            self._idx_cache[key] = {"vi": mk(vi), "vj": mk(vj), "partner": mk([d ^ 1 for d in range(D)]),
                                    "ident": mk(list(range(D))), "D": D}
        return self._idx_cache[key]

    # ---------------------------------------------------------------- forward
    def forward(self, img_feat: Tensor, rot: Tensor, keep_tape: bool):
        """img_feat [V,B,Cf]; rot [B,V,3,3].  Returns (lifted [V,B,3,512], feats [I,D,B,3,512],
        preds [I,D,B,2], tape)."""
        V, B, cf = img_feat.shape
        dev = img_feat.device
        ix = self._indices(V, dev)
        D, I, NV = ix["D"], self.I, NUM_FEAT_VEC
        hl, lifted = self.lifter.forward(img_feat.reshape(V * B, cf))
        rel = torch.empty(D, B, 3, 3, dtype=torch.float32, device=dev)
        ops.relative_rotation(rot.detach().to(torch.float32).contiguous(), ix["vi"], ix["vj"], rel, B, V, D)
        feats = torch.empty(I, D * B, ROT_DIM, dtype=torch.float32, device=dev)
        preds = torch.empty(I, D * B, 2, dtype=torch.float32, device=dev)
        saved = []
        src, src_idx = lifted, ix["vj"]                  # iteration 0 reads the partner VIEW's lifted feature
        for it in range(I):
            X = torch.empty(D * B, self.kin, dtype=torch.float32, device=dev)
            ops.rotcat_fwd(img_feat, src, rel, ix["vi"], src_idx, X, B, D, cf, NV)
            H1, Fn = self.fusers[it].forward(X, feats[it])
            Xh = torch.empty(D * B, self.kin, dtype=torch.float32, device=dev)
            ops.rotcat_fwd(img_feat, Fn, None, ix["vi"], ix["ident"], Xh, B, D, cf, NV)
            Hh, _ = self.heads[it].forward(Xh, preds[it])
            if keep_tape:
                saved.append((X, H1, Xh, Hh))
            src, src_idx = Fn, ix["partner"]             # view j's feature of the SAME pair, previous iteration
        tape = {"img_feat": img_feat, "hl": hl, "rel": rel, "saved": saved, "V": V, "B": B} if keep_tape else None
        return lifted.view(V, B, 3, NV), feats.view(I, D, B, 3, NV), preds.view(I, D, B, 2), tape

    # ---------------------------------------------------------------- backward
    def backward(self, tape: dict, d_lifted: Optional[Tensor], d_feats: Optional[Tensor], d_preds: Optional[Tensor],
                 sink: GradSink) -> Tensor:
        """Gradients wrt the three outputs (None = zero) -> d(img_feat) [V,B,Cf]; parameter gradients
        go to ``sink`` and are published iteration I-1 ... 0, then the lifter."""
        V, B, cf = tape["V"], tape["B"], self.cf
        img_feat, rel = tape["img_feat"], tape["rel"]
        dev = img_feat.device
        ix = self._indices(V, dev)
        D, NV = ix["D"], NUM_FEAT_VEC
        dimg = torch.empty(V, B, cf, dtype=torch.float32, device=dev)
        dimg_live = False
        dF_next: Optional[Tensor] = None
        dlift = torch.empty(V * B, ROT_DIM, dtype=torch.float32, device=dev)
        dlift_live = False
        for it in range(self.I - 1, -1, -1):
            X, H1, Xh, Hh = tape["saved"][it]
            dF = dF_next
            if d_feats is not None:
                ext = d_feats[it].reshape(D * B, ROT_DIM).contiguous()
                if dF is None:
                    dF = ext.clone()
                else:
                    ops.axpby(ext, dF, 1.0, 1.0)
            if d_preds is not None:
                gp = d_preds[it].reshape(D * B, 2).contiguous()
                dXh = self.heads[it].backward(Xh, Hh, gp, sink)
                ops.segment_sum(dXh, self.kin, cf, ix["vi"], dimg, B, D, V, dimg_live)
                dimg_live = True
                if dF is None:
                    dF = torch.empty(D * B, ROT_DIM, dtype=torch.float32, device=dev)
                    ops.rotcat_bwd(dXh, None, ix["ident"], dF, B, D, cf, NV)
                else:
                    dFh = torch.empty(D * B, ROT_DIM, dtype=torch.float32, device=dev)
                    ops.rotcat_bwd(dXh, None, ix["ident"], dFh, B, D, cf, NV)
                    ops.axpby(dFh, dF, 1.0, 1.0)
            else:
                self._zero_grads(self.heads[it], sink)
            dF_next = None
            if dF is None:
                self._zero_grads(self.fusers[it], sink)
            else:
                dX = self.fusers[it].backward(X, H1, dF, sink)
                ops.segment_sum(dX, self.kin, cf, ix["vi"], dimg, B, D, V, dimg_live)
                dimg_live = True
                if it > 0:
                    dF_next = torch.empty(D * B, ROT_DIM, dtype=torch.float32, device=dev)
                    ops.rotcat_bwd(dX, rel, ix["partner"], dF_next, B, D, cf, NV)
                else:
                    tmp = torch.empty(D * B, ROT_DIM, dtype=torch.float32, device=dev)
                    ops.rotcat_bwd(dX, rel, ix["ident"], tmp, B, D, cf, NV)
                    ops.segment_sum(tmp, ROT_DIM, ROT_DIM, ix["vj"], dlift, B, D, V, False)
                    dlift_live = True
            sink.publish(self.heads[it].parameters() + self.fusers[it].parameters())
        if d_lifted is not None:
            ext = d_lifted.reshape(V * B, ROT_DIM).contiguous()
            if dlift_live:
                ops.axpby(ext, dlift, 1.0, 1.0)
            else:
                dlift, dlift_live = ext.clone(), True
        if dlift_live:
            self.lifter.backward(img_feat.reshape(V * B, cf), tape["hl"], dlift, sink,
                                 dx_addend=dimg.view(V * B, cf) if dimg_live else None, dx_out=dimg.view(V * B, cf))
            dimg_live = True
        else:
            self._zero_grads(self.lifter, sink)
        sink.publish(self.lifter.parameters())
        if not dimg_live:
            dimg.zero_()
        return dimg

    @staticmethod
    def _zero_grads(m: _Mlp2, sink: GradSink):
        for p in m.parameters():
            if not sink.accumulate(p):
                sink.view(p).zero_()
