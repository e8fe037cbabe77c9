"""Lifter + cross-view fusion + gaze heads, forward and backward, on the HIP kernels - generalised
from the reference's two views to V views (every unordered pair runs the reference's two-view
recurrence) and covering the constructor variants of the reference model.

What it replaces (for V = 2 exactly):
  * Feat3dLifter.forward ............ /root/reference/models/rot_mv.py:91-98
  * rot_10 / rot_01 .................. :193-194
  * the fusion loop .................. :205-263 with, per variant,
      default        ImageFeatFuser on rot @ F_partner ............ :35-50, :234-239
      ignore_rotmat  ImageFeatFuser on F_partner ................... :226-232
      encode_rotmat  ImageRotmatFeatFuser(img, F_partner, rot) ..... :53-69, :219-225
      share_feature  RotFeatFuser + IntensityBatchNorm on the lifted features :13-32, :72-86, :199-201, :241-247
      share_weights  one fuser / head reused by every iteration .... :148-156

Row space: D = V(V-1) *directed* pairs d = 2p (i<-j), 2p+1 (j<-i) for the p-th unordered pair
(i<j) in lexicographic order; every GEMM of an iteration runs over all D*B rows at once because
the two directions (and all pairs) share the iteration's weights.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from ._lib import ConvDesc
from .arch import DEFAULT_VARIANT, NUM_FEAT_VEC, ROT_DIM, Variant
from .backbone import GradSink

Tensor = torch.Tensor
IBN_MOMENTUM, IBN_EPS = 0.05, 1e-4          # IntensityBatchNorm defaults, rot_mv.py:14
SPLIT_MIN_ROWS = 1024                       # Linears with at least this many rows run on the split-operand kernels


def directed_pairs(views: int) -> Tuple[List[int], List[int]]:
    vi, vj = [], []
    for i in range(views):
        for j in range(i + 1, views):
            vi += [i, j]
            vj += [j, i]
    return vi, vj


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class _Written:
    """Parameters whose gradient this backward has already written: a second write (weights shared
    between iterations) must accumulate whatever ``sink.accumulate`` said at the start."""

    def __init__(self, sink: GradSink, side=None):
        self.sink = sink
        self.seen = set()
        self.side = side          # low-priority stream for the weight / bias gradients (off the critical path)

    def off_path(self, fn, *tensors):
        """Run fn (kernels that only produce parameter gradients) on the side stream, after everything
        queued so far; `tensors` are its inputs (kept from the allocator until the side stream is done)."""
        if self.side is None:
            fn()
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            fn()
        for t in tensors:
            t.record_stream(self.side)

    def acc(self, p) -> bool:
        a = self.sink.accumulate(p) or id(p) in self.seen
        self.seen.add(id(p))
        return a


class _GenInput:
    """Rows [ img_feat[row_img[m]] | rel[m] @ feat[row_src[m]] ] of a fuser / head input, described instead of
    stored: the first Linear's kernels build them in their operand loaders (ops.fuser_fprop / fuser_wgrad)."""
    __slots__ = ("img", "feat", "rel", "row_img", "row_src", "cf", "nvec", "rows", "width")

    def __init__(self, img, feat, rel, row_img, row_src, cf, nvec):
        self.img, self.feat, self.rel, self.row_img, self.row_src, self.cf, self.nvec = img, feat, rel, row_img, row_src, cf, nvec
        self.rows, self.width = row_img.shape[0], cf + 3 * nvec


class _Mlp:
    """[Linear + ReLU] * (n - 1), Linear  (/root/reference/models/backbones/blocks.py:27-60; two layers
    for the lifter, ImageFeatFuser and the heads, three for ImageRotmatFeatFuser / RotFeatFuser).

    Feature widths that are not multiples of 4 (encode_rotmat: K_in + 9 = 2057 / 3593) run on
    zero-padded copies of the weights: the GEMM kernels move 16-byte vectors along K.  The copies are
    refreshed from the parameters every forward; padded hidden units are relu(0) = 0."""

    def __init__(self, params: Dict[str, Tensor], prefix: str, n_layers: int = 2):
        self.w = [params[f"{prefix}blocks.{l}.0.weight"] for l in range(n_layers)]
        self.b = [params[f"{prefix}blocks.{l}.0.bias"] for l in range(n_layers)]
        self.n = n_layers
        self.fin = [w.shape[1] for w in self.w]
        self.fout = [w.shape[0] for w in self.w]
        self.fin_p = [_pad4(f) for f in self.fin]
        self.fout_p = [f if (l == n_layers - 1 and f <= 4) else _pad4(f) for l, f in enumerate(self.fout)]
        self.padded = [self.fin_p[l] != self.fin[l] or self.fout_p[l] != self.fout[l] for l in range(n_layers)]
        self._wp: List[Optional[Tensor]] = [None] * n_layers
        self._bp: List[Optional[Tensor]] = [None] * n_layers
        self.in_width = self.fin_p[0]
        # bf16 path ("mixed"): fp32 tensors, matrix products on the bf16 cores against bf16 weight copies that
        # forward() makes once per step (w [fout][fin] and its transpose for backward-data)
        self.mixed = False
        self._wbf: List[Optional[tuple]] = [None] * n_layers
        # fp32 model, many rows (C3: 1536, C5's shapes: 3584): the fuser / head Linears run on the split-operand kernels
        # (FusionHead._forward_split / _backward_split drive them directly; this class's own forward / backward are the
        # fp32-MFMA and bf16-mixed paths).  `split` only marks which layers FusionHead._prepare_split_weights copies to sp.
        self.split = False
        self._wsp: List[Optional[tuple]] = [None] * n_layers

    def _use_mixed(self, l: int) -> bool:
        return self.mixed and not self.padded[l] and not (l == self.n - 1 and self.fout[l] <= 4)

    def _use_split(self, l: int, rows: int) -> bool:
        return (self.split and not self.mixed and rows >= SPLIT_MIN_ROWS and not self.padded[l] and self.fin[l] % 32 == 0
                and self.fout[l] % 32 == 0)

    def parameters(self):
        out = []
        for w, b in zip(self.w, self.b):
            out += [w, b]
        return out

    def _weights(self, l: int):
        if not self.padded[l]:
            return self.w[l].detach(), self.b[l].detach()
        if self._wp[l] is None or self._wp[l].device != self.w[l].device:
            self._wp[l] = torch.zeros(self.fout_p[l], self.fin_p[l], dtype=torch.float32, device=self.w[l].device)
            self._bp[l] = torch.zeros(self.fout_p[l], dtype=torch.float32, device=self.w[l].device)
        self._wp[l][: self.fout[l], : self.fin[l]].copy_(self.w[l].detach())       # layout plumbing
        self._bp[l][: self.fout[l]].copy_(self.b[l].detach())
        return self._wp[l], self._bp[l]

    def forward(self, x: Optional[Tensor], out: Optional[Tensor] = None, gen: Optional["_GenInput"] = None):
        """x [rows, in_width] -> (hidden activations [h_0 .. h_{n-2}], y [rows, fout_last]).
        gen (instead of x): the input rows [img_feat | R @ F] are generated inside the first layer's GEMM
        (mvg_fuser_fprop) - the concatenated / rotated operand is never materialised."""
        if gen is not None:
            assert x is None and not self.padded[0] and gen.width == self.fin_p[0]
            rows, dev = gen.rows, gen.img.device
        else:
            rows, dev = x.shape[0], x.device
            assert x.shape[1] == self.fin_p[0], (x.shape, self.fin_p[0])
        hs: List[Tensor] = []
        cur = x
        for l in range(self.n):
            w, b = self._weights(l)
            last = l == self.n - 1
            y = out if (last and out is not None) else torch.empty(rows, self.fout_p[l], dtype=torch.float32, device=dev)
            if self._use_mixed(l):
                assert gen is None
                if self._wbf[l] is None:                 # (made by FusionHead._prepare_split_weights(mixed=True) at the start of the step)
                    self._wbf[l] = ops.cast_weights_bf16(ConvDesc.linear(1, self.fin_p[l], self.fout_p[l]), w, self.fin_p[l], True)
                ops.linear_fprop_mixed(cur, self._wbf[l][0], b, not last, y, rows, self.fin_p[l], self.fout_p[l])
            elif l == 0 and gen is not None:
                ops.fuser_fprop(gen.img, gen.feat, gen.rel, gen.row_img, gen.row_src, w, b, not last, y, rows, gen.cf, gen.nvec,
                                self.fout_p[l])
            elif last and self.fout[l] <= 4:
                ops.linear_skinny_fwd(cur, w, b, y, rows, self.fin_p[l], self.fout[l])
            else:
                ops.linear_fprop(cur, w, b, not last, y, rows, self.fin_p[l], self.fout_p[l])
            if not last:
                hs.append(y)
            cur = y
        return hs, cur

    def backward(self, x: Optional[Tensor], hs: List[Tensor], gy: Tensor, sink: GradSink, wr: _Written,
                 dx_addend: Optional[Tensor] = None, dx_out: Optional[Tensor] = None, gen: Optional["_GenInput"] = None) -> Tensor:
        """gy = grad wrt the output; returns grad wrt x (optionally fused `+ dx_addend`).  gen: see forward
        (the first layer's weight gradient regenerates the input rows inside mvg_fuser_wgrad)."""
        rows, dev = gy.shape[0], gy.device
        g = gy
        for l in range(self.n - 1, -1, -1):
            inp = x if l == 0 else hs[l - 1]
            w, _ = self._weights(l) if self.padded[l] else (self.w[l].detach(), None)
            last = l == self.n - 1
            fin, fout = self.fin_p[l], self.fout_p[l]
            # ---- weight / bias gradients
            if self.padded[l]:
                accs = (wr.acc(self.w[l]), wr.acc(self.b[l]))

                def padded_grads(inp=inp, g=g, l=l, fin=fin, fout=fout, accs=accs):
                    dwp = torch.empty(fout, fin, dtype=torch.float32, device=dev)
                    dbp = torch.empty(fout, dtype=torch.float32, device=dev)
                    ops.linear_wgrad(inp, g, dwp, dbp, rows, fin, fout, False)
                    for p, src, a in ((self.w[l], dwp[: self.fout[l], : self.fin[l]], accs[0]),
                                      (self.b[l], dbp[: self.fout[l]], accs[1])):
                        if a:
                            sink.view(p).add_(src)
                        else:
                            sink.view(p).copy_(src)
                wr.off_path(padded_grads, inp, g)
            elif last and self.fout[l] <= 4:
                aw, ab = wr.acc(self.w[l]), wr.acc(self.b[l])
                assert aw == ab
                dh = torch.empty(rows, fin, dtype=torch.float32, device=dev)
                ops.linear_skinny_bwd(g, inp, w, inp, dh, sink.view(self.w[l]), sink.view(self.b[l]), rows, fin, self.fout[l], aw)
                g = dh                               # skinny_bwd already applied the ReLU mask of its input
                continue
            else:
                aw, ab = wr.acc(self.w[l]), wr.acc(self.b[l])

                assert aw == ab

                def grads(inp=inp, g=g, l=l, fin=fin, fout=fout, aw=aw):
                    # weight AND bias gradient in one launch: the bias gradient (column sums of g) rides on the
                    # kernel that streams g for the weight gradient
                    if self._use_mixed(l):
                        ops.linear_wgrad_mixed(inp, g, sink.view(self.w[l]), sink.view(self.b[l]), rows, fin, fout, aw)
                    elif l == 0 and gen is not None:
                        ops.fuser_wgrad(gen.img, gen.feat, gen.rel, gen.row_img, gen.row_src, g, sink.view(self.w[l]),
                                        sink.view(self.b[l]), rows, gen.cf, gen.nvec, fout, aw)
                    else:
                        ops.linear_wgrad(inp, g, sink.view(self.w[l]), sink.view(self.b[l]), rows, fin, fout, aw)
                if l == 0 and gen is not None:
                    # every per-step tensor the side-stream kernel reads must be recorded on that stream: `rel` is
                    # a per-forward allocation owned only by this tape (the row tables are cached for the model's life)
                    wr.off_path(grads, gen.img, gen.feat, g, *([gen.rel] if gen.rel is not None else []))
                else:
                    wr.off_path(grads, inp, g)
            # ---- input gradient
            mixed = self._use_mixed(l)
            if l == 0:
                dx = dx_out if dx_out is not None else torch.empty(rows, fin, dtype=torch.float32, device=dev)
                if mixed:
                    ops.linear_dgrad_mixed(g, self._wbf[l][1], None, dx_addend, dx, rows, fin, fout)
                else:
                    ops.linear_dgrad(g, w, None, dx_addend, dx, rows, fin, fout)
                return dx
            dh = torch.empty(rows, fin, dtype=torch.float32, device=dev)
            if mixed:
                ops.linear_dgrad_mixed(g, self._wbf[l][1], hs[l - 1], None, dh, rows, fin, fout)
            else:
                ops.linear_dgrad(g, w, hs[l - 1], None, dh, rows, fin, fout)        # (g @ W) * (h > 0)
            g = dh
        raise AssertionError("unreachable")


class FusionHead:
    def __init__(self, params: Dict[str, Tensor], fc_dim: int, num_iter: int, variant: Variant = DEFAULT_VARIANT):
        self.cf, self.I, self.v = fc_dim, num_iter, variant
        self.lifter = _Mlp(params, "_lifter._lifter.", 2)
        three = variant.encode_rotmat or variant.share_feature
        self.fusers = [_Mlp(params, f"_img_fusers.{i}._fuser.", 3 if three else 2) for i in range(num_iter)]
        self.heads = [_Mlp(params, f"_gaze_estimators.{i}.", 2) for i in range(num_iter)]
        if variant.share_weights:                     # nn.ModuleList([module] * num_iter): one module
            self.fusers = [self.fusers[0]] * num_iter
            self.heads = [self.heads[0]] * num_iter
        self.ibn = [params[f"_img_fusers.{i}._batchnorm.running_mean"] for i in range(num_iter)] \
            if variant.share_feature else None
        # MVG_FUSED_INPUT=0: materialise the fuser / head inputs with rotcat kernels instead (A/B switch)
        self.fused_input = True
        self.mixed = False                        # bf16 path: Linear products on the bf16 matrix cores (set by the model per call)
        self.split = True                         # fp32 path: Linears with >= SPLIT_MIN_ROWS rows on the split-operand kernels
        self.kin = self.fusers[0].in_width            # row length of the fuser input (zero-padded)
        self.hin = self.heads[0].in_width
        self._idx_cache: Dict[Tuple[int, str], dict] = {}
        self._wprep_state = None
        self._wsp_versions = None

    def _prepare_split_weights(self, rows: int, dev, mixed: bool = False):
        """The sp copies (KRSC for fprop, CRSK for backward-data) of every fuser / head Linear that runs on the split
        kernels this step - or, mixed (the bf16 path), their bf16 copies - made by ONE batched launch pair (like
        Backbone._prepare_weights: persistent destination buffers and a device-resident record table built once per
        parameter placement)."""
        mods = [m for m in self.unique_modules()]
        if mixed:
            mods = [self.lifter] + mods
            layers = [(m, l) for m in mods for l in range(m.n) if m._use_mixed(l)]
        else:
            layers = [(m, l) for m in mods for l in range(m.n) if m._use_split(l, rows)]
        key = (str(dev), mixed, tuple(m.w[l].data_ptr() for m, l in layers))
        if self._wprep_state is None or self._wprep_state[0] != key:
            wstat = torch.zeros(len(layers), 2, dtype=torch.float32, device=dev)
            rows_t, bufs = [], []
            for i, (m, l) in enumerate(layers):
                fin, fout = m.fin[l], m.fout[l]
                if mixed:
                    wk = torch.empty(fout, 1, 1, fin, dtype=torch.bfloat16, device=dev)
                    wt = torch.empty(fin, 1, 1, fout, dtype=torch.bfloat16, device=dev)
                else:
                    wk, wt = ops.sp_empty(fout, fin, device=dev), ops.sp_empty(fin, fout, device=dev)
                    wk.sinv = wt.sinv = wstat[i, 1:2]
                bufs.append((m, l, wk, wt))
                rows_t.append([m.w[l].data_ptr(), wk.data_ptr(), wt.data_ptr(), fout | (1 << 32), fin | (fin << 32), wstat[i].data_ptr()])
            table = torch.tensor(rows_t, dtype=torch.int64).to(dev) if rows_t else None
            self._wprep_state = (key, bufs, table, wstat)
        _, bufs, table, wstat = self._wprep_state
        for m in mods:
            m._wsp = [None] * m.n
            m._wbf = [None] * m.n
        if table is None:
            return
        if not mixed:
            wstat.zero_()
        ops.weights_prep_batch(table, len(bufs), 0 if mixed else 1, 256)
        for m, l, wk, wt in bufs:
            if mixed:
                m._wbf[l] = (wk, wt)
            else:
                m._wsp[l] = (wk, wt)
        self._wsp_versions = tuple(m.w[l]._version for m, l, _, _ in bufs)

    def unique_modules(self) -> List[_Mlp]:
        """Heads and fusers in grad-ready order (last iteration first), each module once."""
        out, seen = [], set()
        for it in range(self.I - 1, -1, -1):
            for m in (self.heads[it], self.fusers[it]):
                if id(m) not in seen:
                    seen.add(id(m))
                    out.append(m)
        return out

    def _indices(self, views: int, dev) -> dict:
        key = (views, str(dev))
        if key not in self._idx_cache:
            vi, vj = directed_pairs(views)
            D = len(vi)
            mk = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)
            self._idx_cache[key] = {"vi": mk(vi), "vj": mk(vj), "partner": mk([d ^ 1 for d in range(D)]),
                                    "ident": mk(list(range(D))), "D": D}
        return self._idx_cache[key]

    def _row_tables(self, views: int, batch: int, dev) -> dict:
        """Row-index tables (int32, device) of the generated fuser / head inputs: row (d, b) takes the image feature
        of view vi[d] and the source feature row given by the table (partner view / partner direction / itself)."""
        key = (views, batch, str(dev), "rows")
        if key not in self._idx_cache:
            vi, vj = directed_pairs(views)
            D = len(vi)
            b = torch.arange(batch, dtype=torch.int32)
            tab = lambda idx: (torch.tensor(idx, dtype=torch.int32)[:, None] * batch + b[None, :]).reshape(-1).contiguous().to(dev)
            self._idx_cache[key] = {"img": tab(vi), "view": tab(vj), "partner": tab([d ^ 1 for d in range(D)]),
                                    "ident": tab(list(range(D)))}
        return self._idx_cache[key]

    # ---------------------------------------------------------------- operand builders
    def _fuser_input(self, a: Tensor, src: Tensor, rel: Tensor, vi, src_idx, B: int, D: int, scales):
        """Rows (d, b) of the fuser's input.  a = image features [V,B,Cf] (lifted features [V,B,3,512]
        for share_feature); src = the partner features, addressed through src_idx."""
        dev, v = a.device, self.v
        X = torch.empty(D * B, self.kin, dtype=torch.float32, device=dev)
        if v.share_feature:
            ops.paircat_fwd(a, src, rel, scales, vi, src_idx, X, B, D, NUM_FEAT_VEC)
        elif v.encode_rotmat:
            ops.rotcat_ext_fwd(a, src, None, rel, vi, src_idx, X, self.kin, B, D, self.cf, NUM_FEAT_VEC)
        else:
            ops.rotcat_fwd(a, src, None if v.ignore_rotmat else rel, vi, src_idx, X, B, D, self.cf, NUM_FEAT_VEC)
        return X

    def _head_input(self, a: Tensor, feat: Tensor, vi, ident, B: int, D: int):
        Xh = torch.empty(D * B, self.hin, dtype=torch.float32, device=a.device)
        if self.v.share_feature:
            ops.paircat_fwd(a, feat, None, None, vi, ident, Xh, B, D, NUM_FEAT_VEC)
        else:
            ops.rotcat_fwd(a, feat, None, vi, ident, Xh, B, D, self.cf, NUM_FEAT_VEC)
        return Xh

    # ---------------------------------------------------------------- forward
    def forward(self, img_feat: Tensor, rot: Tensor, keep_tape: bool, training: bool = True):
        """img_feat [V,B,Cf]; rot [B,V,3,3].  Returns (lifted [V,B,3,512], feats [I,D,B,3,512],
        preds [I,D,B,2], tape)."""
        V, B, cf = img_feat.shape
        dev = img_feat.device
        ix = self._indices(V, dev)
        D, I, NV = ix["D"], self.I, NUM_FEAT_VEC
        # fp32 model, default variant, many rows: the fuser / head Linears on the split-operand kernels (their inputs
        # are then materialised: the span loader of those kernels reads plain rows)
        split_on = self.split and not self.mixed and not (self.v.share_feature or self.v.encode_rotmat) and D * B >= SPLIT_MIN_ROWS
        for m in [self.lifter] + self.fusers + self.heads:
            m.mixed = self.mixed
            m.split = split_on
            m._wbf = [None] * m.n                  # this step's bf16 copies: made below in one launch (or per layer on first use)
        self.lifter.split = False                  # V * B rows: stays on the fp32-MFMA kernels
        if split_on:
            return self._forward_split(img_feat, rot, keep_tape, ix)
        if self.mixed:
            self._prepare_split_weights(D * B, dev, mixed=True)
        hl, lifted = self.lifter.forward(img_feat.reshape(V * B, cf))
        rel = torch.empty(D, B, 3, 3, dtype=torch.float32, device=dev)
        ops.relative_rotation(rot.detach().to(torch.float32).contiguous(), ix["vi"], ix["vj"], rel, B, V, D)
        feats = torch.empty(I, D * B, ROT_DIM, dtype=torch.float32, device=dev)
        preds = torch.empty(I, D * B, 2, dtype=torch.float32, device=dev)
        a = lifted if self.v.share_feature else img_feat       # rot_mv.py:199-201
        saved = []
        src, src_idx = lifted, ix["vj"]                  # iteration 0 reads the partner VIEW's lifted feature
        # default / ignore_rotmat variants: rotate + concat run inside the first Linear's operand loader - the
        # fuser input X = [img_feat | R @ F] and the head input [img_feat | F] are never written (the other
        # variants have 9 extra columns / interleaved rows and keep the materialised form)
        fused_in = self.fused_input and not (self.v.share_feature or self.v.encode_rotmat) and not self.mixed and not split_on
        rt = self._row_tables(V, B, dev) if fused_in else None
        img2d = img_feat.reshape(V * B, cf)
        for it in range(I):
            scales = None
            if fused_in:
                gen_f = _GenInput(img2d, src.reshape(-1, ROT_DIM), None if self.v.ignore_rotmat else rel, rt["img"],
                                  rt["view"] if it == 0 else rt["partner"], cf, NV)
                Hf, Fn = self.fusers[it].forward(None, feats[it], gen=gen_f)
                gen_h = _GenInput(img2d, Fn, None, rt["img"], rt["ident"], cf, NV)
                Hh, _ = self.heads[it].forward(None, preds[it], gen=gen_h)
                if keep_tape:
                    saved.append((gen_f, Hf, gen_h, Hh, None))
                src, src_idx = Fn, ix["partner"]
                continue
            if self.v.share_feature:
                scales = torch.empty(2 * D, NV, dtype=torch.float32, device=dev)
                ops.ibn_scales(a, src, ix["vi"], src_idx, self.ibn[it], training, IBN_MOMENTUM, IBN_EPS, scales, B, D, NV)
            X = self._fuser_input(a, src, rel, ix["vi"], src_idx, B, D, scales)
            Hf, Fn = self.fusers[it].forward(X, feats[it])
            Xh = self._head_input(a, Fn, ix["vi"], ix["ident"], B, D)
            Hh, _ = self.heads[it].forward(Xh, preds[it])
            if keep_tape:
                saved.append((X, Hf, Xh, Hh, scales))
            src, src_idx = Fn, ix["partner"]             # view j's feature of the SAME pair, previous iteration
        # the tape remembers which kernel family (and which weight copies) its forward ran on: backward dispatches on
        # that, not on the modules' per-call state, which a later forward (a small validation batch, another
        # compute_dtype) may have changed in between
        tape = {"img_feat": img_feat, "lifted": lifted, "hl": hl, "rel": rel, "saved": saved, "V": V, "B": B, "mode": "mixed" if self.mixed else "fp32",
                "wbf": {id(m): list(m._wbf) for m in [self.lifter] + self.fusers + self.heads},
                "wsp_versions": self._wsp_versions if self.mixed else None} if keep_tape else None
        return lifted.view(V, B, 3, NV), feats.view(I, D, B, 3, NV), preds.view(I, D, B, 2), tape

    # ---------------------------------------------------------------- the split-operand path (fp32 model, >= SPLIT_MIN_ROWS rows)
    _SLOTS = 64

    def _forward_split(self, img_feat: Tensor, rot: Tensor, keep_tape: bool, ix: dict):
        """The fuser / head Linears on the split kernels (conv_split.hip), every operand with its own power-of-two scale.

        Per iteration: fuser L1 (hidden activation written in sp, scaled from a bound), fuser L2 (fp32 features + their
        abs-max from the epilogue), ONE builder launch that writes this iteration's head input and the next iteration's
        fuser input straight into sp (mvg_fuse_build_split), head L1, head L2 (512 -> 2).  No fp32 copy of an operand is
        ever made and no separate abs-max / split pass runs: the scales come from device slots that the producing launches
        fill (a 64-float arena owned by this call's tape)."""
        V, B, cf = img_feat.shape
        dev = img_feat.device
        D, I, NV = ix["D"], self.I, NUM_FEAT_VEC
        rows, kin = D * B, self.kin
        assert 2 + I <= 8 and 11 * I + 2 <= self._SLOTS
        self._prepare_split_weights(rows, dev)
        st = torch.zeros(self._SLOTS, dtype=torch.float32, device=dev)          # abs-max slots must start at zero
        names: Dict[str, int] = {}

        def slot(name: str) -> Tensor:
            i = names.setdefault(name, len(names))
            return st[i:i + 1]

        def sp(r: int, c: int, name: str) -> Tensor:
            t = ops.sp_empty(r, c, device=dev)
            t.sinv = slot(name)
            return t
        for m in [self.lifter] + self.fusers + self.heads:
            m.split = False                          # (the modules' own forward / backward are the fp32-MFMA path: the lifter's)
        hl, lifted = self.lifter.forward(img_feat.reshape(V * B, cf))
        rel = torch.empty(D, B, 3, 3, dtype=torch.float32, device=dev)
        ops.relative_rotation(rot.detach().to(torch.float32).contiguous(), ix["vi"], ix["vj"], rel, B, V, D)
        rel_fuse = None if self.v.ignore_rotmat else rel
        rt = self._row_tables(V, B, dev)
        img2d = img_feat.reshape(V * B, cf)
        ops.absmax_multi([img2d, lifted] + [self.fusers[it].b[0].detach() for it in range(I)],
                         [slot("am_img"), slot("am_lift")] + [slot(f"bam{it}") for it in range(I)])
        feats = torch.empty(I, rows, ROT_DIM, dtype=torch.float32, device=dev)
        preds = torch.empty(I, rows, 2, dtype=torch.float32, device=dev)
        hh = torch.empty(I, rows, self.heads[0].fout[0], dtype=torch.float32, device=dev)
        xf = sp(rows, kin, "xf0")
        ops.fuse_build_split(img2d, lifted, rel_fuse, rt["img"], rt["view"], None, xf, None, slot("am_img"), slot("am_lift"), rows, cf, NV)
        saved = []
        for it in range(I):
            fu, hd = self.fusers[it], self.heads[it]
            wf0, wf1, wh0 = fu._wsp[0], fu._wsp[1], hd._wsp[0]
            hid, hhid = fu.fout[0], hd.fout[0]
            h = sp(rows, hid, f"h{it}")
            ops.linear_fprop_split(xf, wf0[0], fu.b[0].detach(), True, h, rows, kin, hid, bias_absmax=slot(f"bam{it}"))
            ops.linear_fprop_split(h, wf1[0], fu.b[1].detach(), False, feats[it], rows, hid, ROT_DIM, out_absmax=slot(f"amF{it}"))
            xh = sp(rows, kin, f"xh{it}")
            xf_next = sp(rows, kin, f"xf{it + 1}") if it + 1 < I else None
            ops.fuse_build_split(img2d, feats[it], rel_fuse, rt["img"], rt["partner"], rt["ident"], xf_next, xh, slot("am_img"),
                                 slot(f"amF{it}"), rows, cf, NV)
            ops.linear_fprop_split(xh, wh0[0], hd.b[0].detach(), True, hh[it], rows, kin, hhid)
            ops.linear_skinny_fwd(hh[it], hd.w[1].detach(), hd.b[1].detach(), preds[it], rows, hhid, 2)
            if keep_tape:
                saved.append((xf, h, xh, hh[it], wf0, wf1, wh0))
            xf = xf_next
        tape = {"img_feat": img_feat, "lifted": lifted, "hl": hl, "rel": rel, "saved": saved, "V": V, "B": B, "mode": "split",
                "st": st, "slot": slot, "wsp_versions": self._wsp_versions} if keep_tape else None
        return lifted.view(V, B, 3, NV), feats.view(I, D, B, 3, NV), preds.view(I, D, B, 2), tape

    def _backward_split(self, tape: dict, d_lifted: Optional[Tensor], d_feats: Optional[Tensor], d_preds: Optional[Tensor],
                        sink: GradSink, side=None) -> Tensor:
        """Backward of _forward_split.  Per iteration, on the critical path: head L2 backward (its dx with the abs-max),
        [split + bias gradient -> head L1 dgrad], ONE mvg_fuse_unbuild (dF_it from the head input's and the next fuser
        input's gradients, the image-feature gradient sums), then for fuser L2 and L1: split + bias gradient (one launch,
        scaled from the slot the producer filled) -> dgrad (with the next abs-max in its epilogue).  Weight gradients run
        on the side stream."""
        V, B, cf, v = tape["V"], tape["B"], self.cf, self.v
        img_feat, rel, st, slot = tape["img_feat"], tape["rel"], tape["st"], tape["slot"]
        dev = img_feat.device
        ix = self._indices(V, dev)
        D, NV, I = ix["D"], NUM_FEAT_VEC, self.I
        rows, kin = D * B, self.kin
        wr = _Written(sink, side)
        rel_fuse = None if v.ignore_rotmat else rel
        da = torch.empty(V, B, cf, dtype=torch.float32, device=dev)
        da_live = False
        dXn: Optional[Tensor] = None                 # gradient of the NEXT iteration's fuser input

        def sp(r: int, c: int, name: str) -> Tensor:
            t = ops.sp_empty(r, c, device=dev)
            t.sinv = slot(name)
            return t

        def wgrad(x_sp, g_sp, p, fin, fout, acc):
            wr.off_path(lambda: ops.linear_wgrad_split(x_sp, g_sp, sink.view(p), rows, fin, fout, acc), x_sp, g_sp, st)

        for it in range(I - 1, -1, -1):
            xf, h, xh, hh, wf0, wf1, wh0 = tape["saved"][it]
            fu, hd = self.fusers[it], self.heads[it]
            hid, hhid = fu.fout[0], hd.fout[0]
            dXh = None
            if d_preds is not None:
                gp = d_preds[it].reshape(rows, 2).contiguous()
                a1, ab1 = wr.acc(hd.w[1]), wr.acc(hd.b[1])
                assert a1 == ab1
                dhh = torch.empty(rows, hhid, dtype=torch.float32, device=dev)
                ops.linear_skinny_bwd(gp, hh, hd.w[1].detach(), hh, dhh, sink.view(hd.w[1]), sink.view(hd.b[1]), rows, hhid, 2, a1,
                                      dx_absmax=slot(f"am_dhh{it}"))
                a0, ab0 = wr.acc(hd.w[0]), wr.acc(hd.b[0])
                assert a0 == ab0
                g_hh = sp(rows, hhid, f"g_hh{it}")
                ops.split_colsum(dhh, rows, hhid, slot(f"am_dhh{it}"), g_hh, sink.view(hd.b[0]), ab0)
                wgrad(xh, g_hh, hd.w[0], kin, hhid, a0)
                dXh = torch.empty(rows, kin, dtype=torch.float32, device=dev)
                ops.linear_dgrad_split(g_hh, wh0[1], dXh, rows, kin, hhid)
            else:
                self._zero_grads(hd, sink, wr)
            ext = d_feats[it].reshape(rows, ROT_DIM).contiguous() if d_feats is not None else None
            if dXh is None and dXn is None and ext is None:
                self._zero_grads(fu, sink, wr)
                if not v.share_weights:
                    sink.publish(hd.parameters() + fu.parameters())
                continue
            am_dF = slot(f"am_dF{it}")
            if dXh is None and dXn is None:
                dF = ext.clone()
            else:
                dF = torch.empty(rows, ROT_DIM, dtype=torch.float32, device=dev)
                ops.fuse_unbuild(dXh, dXn, rel_fuse, ix["partner"], ix["vi"], dF, da, da_live, D, V, D, B, cf, NV,
                                 absmax=am_dF if ext is None else None)
                da_live = True
                if ext is not None:
                    ops.axpby(ext, dF, 1.0, 1.0)
            if ext is not None:
                ops.absmax_multi([dF], [am_dF])
            # fuser layer 2: features <- hidden
            a1, ab1 = wr.acc(fu.w[1]), wr.acc(fu.b[1])
            assert a1 == ab1
            g_F = sp(rows, ROT_DIM, f"g_F{it}")
            ops.split_colsum(dF, rows, ROT_DIM, am_dF, g_F, sink.view(fu.b[1]), ab1)
            wgrad(h, g_F, fu.w[1], hid, ROT_DIM, a1)
            dh = torch.empty(rows, hid, dtype=torch.float32, device=dev)
            ops.linear_dgrad_split(g_F, wf1[1], dh, rows, hid, ROT_DIM, relu_mask_sp=h, out_absmax=slot(f"am_dh{it}"))     # (g W) * (h > 0)
            # fuser layer 1: hidden <- the built input
            a0, ab0 = wr.acc(fu.w[0]), wr.acc(fu.b[0])
            assert a0 == ab0
            g_h = sp(rows, hid, f"g_h{it}")
            ops.split_colsum(dh, rows, hid, slot(f"am_dh{it}"), g_h, sink.view(fu.b[0]), ab0)
            wgrad(xf, g_h, fu.w[0], kin, hid, a0)
            dXn = torch.empty(rows, kin, dtype=torch.float32, device=dev)
            ops.linear_dgrad_split(g_h, wf0[1], dXn, rows, kin, hid)
            if not v.share_weights:
                sink.publish(hd.parameters() + fu.parameters())
        if v.share_weights:
            sink.publish(self.heads[0].parameters() + self.fusers[0].parameters())
        dlift, dlift_live = None, False
        if dXn is not None:
            # iteration 0 read the partner VIEW's lifted feature: sum the directions per source view (and the last image-feature terms)
            dlift = torch.empty(V * B, ROT_DIM, dtype=torch.float32, device=dev)
            ops.fuse_unbuild(None, dXn, rel_fuse, ix["vj"], ix["vi"], dlift, da, da_live, V, V, D, B, cf, NV)
            da_live = dlift_live = True
        if d_lifted is not None:
            ext = d_lifted.reshape(V * B, ROT_DIM).contiguous()
            if dlift_live:
                ops.axpby(ext, dlift, 1.0, 1.0)
            else:
                dlift, dlift_live = ext.clone(), True
        dimg = da
        if dlift_live:
            self.lifter.split = self.lifter.mixed = False
            self.lifter.backward(img_feat.reshape(V * B, cf), tape["hl"], dlift, sink, wr,
                                 dx_addend=dimg.view(V * B, cf) if da_live else None, dx_out=dimg.view(V * B, cf))
            da_live = True
        else:
            self._zero_grads(self.lifter, sink, wr)
        sink.publish(self.lifter.parameters())
        if not da_live:
            dimg.zero_()
        return dimg

    # ---------------------------------------------------------------- backward
    def backward(self, tape: dict, d_lifted: Optional[Tensor], d_feats: Optional[Tensor], d_preds: Optional[Tensor],
                 sink: GradSink, side=None) -> Tensor:
        """Gradients wrt the three outputs (None = zero) -> d(img_feat) [V,B,Cf]; parameter gradients
        go to ``sink`` and are published iteration I-1 ... 0 (shared weights: once, after iteration 0),
        then the lifter."""
        V, B, cf, v = tape["V"], tape["B"], self.cf, self.v
        if tape.get("wsp_versions") is not None and tape["wsp_versions"] != self._wsp_versions:
            raise RuntimeError("backward of a tape whose sp weight copies were overwritten by a later forward with DIFFERENT "
                               "weights (forward, optimizer step, forward, then backward of the first call): run backward "
                               "before the weights change")
        if tape.get("mode") == "split":
            return self._backward_split(tape, d_lifted, d_feats, d_preds, sink, side)
        # the kernel family this tape's forward ran on (not whatever a later forward left on the modules)
        for m in [self.lifter] + self.fusers + self.heads:
            m.split = False
            m.mixed = tape.get("mode") == "mixed"
            if "wbf" in tape:
                m._wbf = list(tape["wbf"][id(m)])
        img_feat, rel = tape["img_feat"], tape["rel"]
        dev = img_feat.device
        ix = self._indices(V, dev)
        D, NV = ix["D"], NUM_FEAT_VEC
        wr = _Written(sink, side)
        # "a" = what every fuser / head input starts with: image features, or (share_feature) the lifted ones
        aw = ROT_DIM if v.share_feature else cf
        da = torch.empty(V, B, aw, dtype=torch.float32, device=dev)
        da_live = False
        dF_next: Optional[Tensor] = None
        dlift = torch.empty(V * B, ROT_DIM, dtype=torch.float32, device=dev)
        dlift_live = False
        rel_fuse = None if (v.ignore_rotmat or v.encode_rotmat) else rel

        def split_input_grad(dX: Tensor, ld: int, rel_b, scales, src_idx, dsrc: Tensor):
            """dX of a fuser / head input -> da (+= over directions) and the gradient of the feature operand."""
            nonlocal da_live
            if v.share_feature:
                da_dir = torch.empty(D * B, ROT_DIM, dtype=torch.float32, device=dev)
                ops.paircat_bwd(dX, rel_b, scales, src_idx, da_dir, dsrc, B, D, NV)
                ops.segment_sum(da_dir, ROT_DIM, ROT_DIM, ix["vi"], da, B, D, V, da_live)
            else:
                ops.segment_sum(dX, ld, cf, ix["vi"], da, B, D, V, da_live)
                if ld == cf + ROT_DIM:
                    ops.rotcat_bwd(dX, rel_b, src_idx, dsrc, B, D, cf, NV)
                else:
                    ops.rotcat_ext_bwd(dX, ld, rel_b, src_idx, dsrc, B, D, cf, NV)
            da_live = True

        for it in range(self.I - 1, -1, -1):
            X, Hf, Xh, Hh, scales = tape["saved"][it]
            dF = dF_next
            if d_feats is not None:
                ext = d_feats[it].reshape(D * B, ROT_DIM).contiguous()
                if dF is None:
                    dF = ext.clone()
                else:
                    ops.axpby(ext, dF, 1.0, 1.0)
            gen_f = X if isinstance(X, _GenInput) else None
            gen_h = Xh if isinstance(Xh, _GenInput) else None
            if d_preds is not None:
                gp = d_preds[it].reshape(D * B, 2).contiguous()
                dXh = self.heads[it].backward(None if gen_h else Xh, Hh, gp, sink, wr, gen=gen_h)
                dFh = torch.empty(D * B, ROT_DIM, dtype=torch.float32, device=dev)
                split_input_grad(dXh, self.hin, None, None, ix["ident"], dFh)
                if dF is None:
                    dF = dFh
                else:
                    ops.axpby(dFh, dF, 1.0, 1.0)
            else:
                self._zero_grads(self.heads[it], sink, wr)
            dF_next = None
            if dF is None:
                self._zero_grads(self.fusers[it], sink, wr)
            else:
                dX = self.fusers[it].backward(None if gen_f else X, Hf, dF, sink, wr, gen=gen_f)
                if it > 0:
                    dF_next = torch.empty(D * B, ROT_DIM, dtype=torch.float32, device=dev)
                    split_input_grad(dX, self.kin, rel_fuse, scales, ix["partner"], dF_next)
                else:
                    tmp = torch.empty(D * B, ROT_DIM, dtype=torch.float32, device=dev)
                    split_input_grad(dX, self.kin, rel_fuse, scales, ix["ident"], tmp)
                    ops.segment_sum(tmp, ROT_DIM, ROT_DIM, ix["vj"], dlift, B, D, V, False)
                    dlift_live = True
            if not v.share_weights:
                sink.publish(self.heads[it].parameters() + self.fusers[it].parameters())
        if v.share_weights:
            sink.publish(self.heads[0].parameters() + self.fusers[0].parameters())
        if d_lifted is not None:
            ext = d_lifted.reshape(V * B, ROT_DIM).contiguous()
            if dlift_live:
                ops.axpby(ext, dlift, 1.0, 1.0)
            else:
                dlift, dlift_live = ext.clone(), True
        if v.share_feature and da_live:                  # the lifted features ARE the "image features" here
            if dlift_live:
                ops.axpby(da.view(V * B, ROT_DIM), dlift, 1.0, 1.0)
            else:
                dlift, dlift_live = da.view(V * B, ROT_DIM), True
        dimg = da if not v.share_feature else torch.empty(V, B, cf, dtype=torch.float32, device=dev)
        dimg_live = da_live and not v.share_feature
        if dlift_live:
            self.lifter.backward(img_feat.reshape(V * B, cf), tape["hl"], dlift, sink, wr,
                                 dx_addend=dimg.view(V * B, cf) if dimg_live else None, dx_out=dimg.view(V * B, cf))
            dimg_live = True
        else:
            self._zero_grads(self.lifter, sink, wr)
        sink.publish(self.lifter.parameters())
        if not dimg_live:
            dimg.zero_()
        return dimg

    @staticmethod
    def _zero_grads(m: _Mlp, sink: GradSink, wr: _Written):
        for p in m.parameters():
            if not wr.acc(p):
                sink.view(p).zero_()
