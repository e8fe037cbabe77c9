"""ResNet-18/50 backbone over all views at once, forward and backward, on the HIP kernels.

What it replaces: ``self._feat_extractor(img_k)`` for every view k
(/root/reference/models/rot_mv.py:124-128,196-197 -> /root/reference/models/resnet.py:261-275 with
BasicBlock :80-96 / Bottleneck :128-148).  The reference runs one backbone pass per view; here
all V views go through each kernel launch as V *groups* - BatchNorm statistics stay per group and
running statistics are updated once per group in view order, so results equal V separate passes.

Data layout in HBM: activations NHWC fp32 ``[V][B][H][W][C]``; per conv+BN unit the tape keeps
the raw conv output ``y`` (BN backward needs x-hat) and the post-activation ``out`` (next conv's
input, ReLU mask); weights are the PyTorch parameters themselves in channels_last (= KRSC).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional

import torch

from . import ops
from ._lib import ConvDesc
from .arch import BackboneSpec, ConvSpec, backbone_spec

Tensor = torch.Tensor
BN_EPS, BN_MOMENTUM = 1e-5, 0.1          # nn.BatchNorm2d defaults (resnet.py:185)
IMAGE_MEAN, IMAGE_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # main.py:38-39


class GradSink:
    """Where parameter gradients go.  ``view(p)`` returns the tensor the kernels write into and
    ``accumulate(p)`` says whether they must add to it; ``publish(ps)`` is called once a group of
    parameters has its final gradient (in grad-ready order: DP buckets hang off this)."""

    def view(self, p: torch.nn.Parameter) -> Tensor:
        raise NotImplementedError

    def accumulate(self, p: torch.nn.Parameter) -> bool:
        raise NotImplementedError

    def publish(self, ps: List[torch.nn.Parameter]) -> None:
        raise NotImplementedError


class _Unit:
    """Saved state of one conv + BN (+ReLU) (+residual) for the backward pass."""
    __slots__ = ("spec", "desc", "x_in", "y", "out", "mean", "invstd", "relu", "rows", "w", "trained", "pool",
                 "relu_affine", "fused_s12", "relu_bits", "split", "stem_rw")

    def __init__(self):
        self.split = False        # conv operands (x_in, out, dy, w) in sp (two fp16 pieces), split-operand kernels (conv_split.hip)
        self.stem_rw = False      # the stem in row-window form: x_in is the window operand, dy goes out in sp
        self.relu_bits = None     # residual units: the ReLU mask as one byte per 16-byte access (ops.bn_apply_bits)
        self.fused_s12 = None     # BatchNorm-backward sums delivered by the backward-data launch that produced this unit's gradient


class Backbone:
    _guard_scale = 1          # tests raise it to exercise the 2 GiB guard of the split path without a 2 GiB tensor

    def __init__(self, depth: int, params: Dict[str, Tensor], prefix: str = "_feat_extractor.0."):
        self.spec: BackboneSpec = backbone_spec(depth, prefix)
        self.p = params                      # name -> Parameter / buffer (live objects)
        self.fc_dim = self.spec.fc_dim
        # Backward-weight kernels are off the critical path (bn_bwd -> dgrad -> bn_bwd ...): they run on a
        # side stream, where their MFMA work overlaps the HBM-bound BatchNorm-backward passes and the
        # stream-K fix-ups of the main stream.  grad_streams lists every stream that writes gradients
        # besides the caller's (the data-parallel reducer waits on them too).
        self.overlap_wgrad = True
        self.overlap_head = True      # the fusion block's weight gradients too (+1.6 %)
        self.wgrad_low_priority = True      # data-parallel runs use an ordinary stream: gradients must not finish last
        self.grad_streams: List[torch.cuda.Stream] = []
        self._wg_stream: Optional[torch.cuda.Stream] = None
        self._wg_low = True
        # storage type of activations / activation gradients / conv operands: fp32 = the parity path (1e-4 vs
        # the reference); bf16 = BASELINE config C5's "bf16 MFMA path" (fp32 master weights, statistics, gradients)
        self.act_dtype = torch.float32
        # residual units record their ReLU mask as bits in the forward apply pass; the backward reduce pass reads
        # those (1/16 of the activation's bytes) instead of the activation (attribute False: read the activation).
        self.relu_bits = True
        # fp32 training steps on the split-operand kernels (conv_split.hip): every conv but the 3-channel stem reads its
        # operands as two fp16 pieces per fp32 value and runs three fp16 MFMAs per product - fp32-accurate (the 1e-4
        # parity path).  MVG_SPLIT=0: the fp32-MFMA kernels everywhere.
        self.split = os.environ.get("MVG_SPLIT", "1") != "0"
        # split and bf16 paths: the BatchNorm-backward reduce pass of a unit rides on the backward-data launch that
        # produces its output gradient (every unit but the stem and the last one): the staged epilogue already holds 8
        # channels of a row per lane, reads y (and the mask bits) with 16-byte accesses, stores the masked gradient and
        # leaves one partial per row tile; a stride-2 launch's parity classes each bring theirs, the classes a 1x1
        # stride-2 filter never touches as epilogue-only tiles.  (attribute False: separate reduce passes - the tests
        # compare the two.)
        self.fuse_bn_split = True
        # split path, training: the 7x7 stem on the split kernels too, in its "row-window" form (mvg_stem_fprop_split: the
        # image rewritten as [.., W/2, 8 columns x 4 channels] windows, a 7 x 1 filter over 32 channels, K = 224) instead
        # of the fp32-MFMA kernel on 4-channel taps (K = 196 at a fifth of the matrix rate).  Not when the caller wants
        # d(loss)/d(img) (the backward-data launch runs on the fp32 kernel) or the width is odd.
        self.stem_rowwindow = True
        self.split_eval = True      # inference forward on the split kernels too
        self._wk_cache: Dict[str, tuple] = {}     # inference: conv name -> (data_ptr, version, sp weights)
        # training: one launch per step makes every conv's bf16 / sp weight copies
        self.batch_weight_prep = True
        self._wprep: Optional[Dict[str, tuple]] = None
        self._wprep_state = None
        self._wprep_versions = None
        self._wg_defer: Optional[list] = None # backward: (slabs, dw, splits, accumulate) of the split wgrads whose reduce is pending
        self._split_now = self.split          # per forward call: off when a view's largest sp tensor would exceed 2 GiB
        self._stem_rw = False                 # per forward call: the stem runs in row-window form on the split kernels
        self._stem_w8 = None

    @property
    def bf16(self) -> bool:
        return self.act_dtype == torch.bfloat16

    # ---------------------------------------------------------------- helpers
    def _weight(self, c: ConvSpec) -> Tensor:
        """KRSC device tensor the kernels read.  The 3-channel stem is padded to 4 channels."""
        w = self.p[c.name + ".weight"]
        if c.cin == 3:
            w4 = torch.zeros(c.cout, c.k, c.k, 4, dtype=torch.float32, device=w.device)
            w4[..., :3].copy_(w.detach().permute(0, 2, 3, 1))          # 9408 floats: layout plumbing
            return w4
        assert w.is_contiguous(memory_format=torch.channels_last) or (c.k == 1 and w.is_contiguous()), \
            f"{c.name}.weight must be channels_last (KRSC)"
        return w.detach()

    def _prepare_weights(self, dev) -> Dict[str, tuple]:
        """The per-step copies of every conv's weights the bf16 / split kernels read (KRSC for fprop, CRSK for
        backward-data), made by ONE launch: destination buffers and the launch's device-resident table of
        (source, destinations, shape) records are built once per parameter placement and reused every step."""
        mode = 0 if self.bf16 else 1
        stem_rw = self._stem_rw
        convs = [c for c in self.spec.all_convs() if not (mode == 1 and c.cin == 3 and not stem_rw)]
        key = (mode, stem_rw, str(dev), tuple(self.p[c.name + ".weight"].data_ptr() for c in convs))
        if self._wprep_state is None or self._wprep_state[0] != key:
            out, rows = {}, []
            wstat = torch.zeros(len(convs), 2, dtype=torch.float32, device=dev)      # per conv {max |w| bits, 2^-k} (split path)
            self._stem_w8 = None
            for ci, c in enumerate(convs):
                wsrc = self.p[c.name + ".weight"]
                assert wsrc.is_contiguous(memory_format=torch.channels_last) or (c.k == 1 and wsrc.is_contiguous())
                rs = c.k * c.k
                if mode == 1 and c.cin == 3:
                    # the stem in row-window form: w'[o][r][j][c] = w[o][r][j - 1][c] (j = 0 and c = 3 zero), refreshed from the
                    # parameter below (9408 floats: layout plumbing), then split like a [cout][7][1][32] filter
                    assert c.k == 7
                    self._stem_w8 = torch.zeros(c.cout, 7, 8, 4, dtype=torch.float32, device=dev)
                    wk = ops.sp_empty(c.cout, 7 * 32, device=dev)
                    wk.sinv = wstat[ci, 1:2]
                    out[c.name] = (wk, None)
                    rows.append([self._stem_w8.data_ptr(), wk.data_ptr(), 0, c.cout | (7 << 32), 32 | (32 << 32), wstat[ci].data_ptr()])
                    continue
                if mode == 0 and c.cin == 3 and stem_rw:
                    # bf16 path, folded windows (16 columns x 4 channels serve two output columns): two copies of the filter,
                    # w'[par * cout + o][r][j][c] = w[o][r][j - 1 - 2 par][c], cast like a [2 cout][7][1][64] filter
                    assert c.k == 7
                    self._stem_w8 = torch.zeros(2 * c.cout, 7, 16, 4, dtype=torch.float32, device=dev)
                    wk = torch.empty(2 * c.cout, 7, 1, 64, dtype=torch.bfloat16, device=dev)
                    out[c.name] = (wk, None)
                    rows.append([self._stem_w8.data_ptr(), wk.data_ptr(), 0, (2 * c.cout) | (7 << 32), 64 | (64 << 32), wstat[ci].data_ptr()])
                    continue
                if mode == 1:
                    wk = ops.sp_empty(c.cout, rs * c.cin, device=dev)
                    wt = ops.sp_empty(c.cin, rs * c.cout, device=dev)
                    wk.sinv = wt.sinv = wstat[ci, 1:2]
                    cin_pad = c.cin
                else:
                    cin_pad = 8 if c.cin == 3 else c.cin
                    wk = torch.empty(c.cout, c.k, c.k, cin_pad, dtype=torch.bfloat16, device=dev)
                    wt = None if c.cin == 3 else torch.empty(cin_pad, c.k, c.k, c.cout, dtype=torch.bfloat16, device=dev)
                out[c.name] = (wk, wt)
                rows.append([wsrc.data_ptr(), wk.data_ptr(), wt.data_ptr() if wt is not None else 0, c.cout | (rs << 32),
                             c.cin | (cin_pad << 32), wstat[ci].data_ptr()])
            table = torch.tensor(rows, dtype=torch.int64).to(dev)            # once per placement (a host-to-device copy)
            self._wprep_state = (key, out, table, len(rows), mode, wstat)
        _, out, table, n, mode, wstat = self._wprep_state
        if mode == 1:
            wstat.zero_()                                  # the max |w| slots are atomicMax targets
            if self._stem_w8 is not None:
                self._stem_w8[:, :, 1:, :3].copy_(self.p[self.spec.stem.name + ".weight"].detach().permute(0, 2, 3, 1))
        elif self._stem_w8 is not None:
            wsrc, co = self.p[self.spec.stem.name + ".weight"].detach().permute(0, 2, 3, 1), self.spec.stem.cout
            self._stem_w8[:co, :, 1:8, :3].copy_(wsrc)
            self._stem_w8[co:, :, 3:10, :3].copy_(wsrc)
        ops.weights_prep_batch(table, n, mode)
        # the copies live in persistent buffers that the next forward overwrites: remember which parameter versions
        # they hold, so that the backward of an OLDER tape can tell (tapes keep pointers into these buffers)
        self._wprep_versions = tuple(self.p[c.name + ".weight"]._version for c in self.spec.all_convs())
        return out

    def invalidate_weight_cache(self):
        """Forget the inference path's cached sp copies of the conv weights.  The cache is keyed on each parameter's
        (data_ptr, version counter); writes that bypass the counter - ``p.data.copy_()`` / ``p.data.mul_()`` (EMA,
        clipping) or writes to ``model.param_arena()`` - must be followed by this call (or by
        ``torch.autograd.graph.increment_version(p)``), otherwise ``torch.no_grad()`` inference keeps using the old
        weights.  ``model.train()`` and every training forward clear it too."""
        self._wk_cache.clear()

    def bn_count_buffers(self) -> List[Tensor]:
        return [self.p[c.bn + ".num_batches_tracked"] for c in self.spec.all_convs()]

    # ---------------------------------------------------------------- forward
    def _unit_fwd(self, c: ConvSpec, x: Tensor, G: int, N: int, H: int, W: int, training: bool, relu: bool,
                  residual: Optional[Tensor], tape: Optional[list], pool: bool = False, residual_affine=None,
                  defer_apply: bool = False):
        """conv -> BatchNorm (-> + residual) (-> ReLU).  pool=True (the stem): the 3x3/2 max pool is
        fused behind the ReLU and (pooled, argmax) is returned; the normalised map is not stored.
        defer_apply (the downsample branch): stop after the statistics and return (y, (scale, shift)) - the
        normalisation is applied by the consumer, the block's last unit, which takes them as
        residual / residual_affine; the normalised downsample map is never written."""
        bf = self.bf16
        cin = (8 if bf else 4) if c.cin == 3 else c.cin
        d = ConvDesc.make(G, N, H, W, cin, c.cout, c.k, c.stride, c.pad)
        dev = x.device
        # split path: training steps of the fp32 model; sp_in = this conv reads sp operands (all but the stem),
        # sp_out = its consumers do (every unit: the stem's pooled map feeds layer1)
        sp_out = self._split_now and training and not bf
        sp_in = sp_out and c.cin != 3
        stem_rw = c.cin == 3 and self._stem_rw and (sp_out or bf)  # x is then the row-window operand (ops.stem_rowwindow_split / _bf16)
        if (sp_in or bf or stem_rw) and self._wprep is not None:
            w, w_t = self._wprep[c.name]         # this step's copies, made by ONE launch at the start of forward()
        elif sp_in:
            wsrc = self.p[c.name + ".weight"].detach()
            assert wsrc.is_contiguous(memory_format=torch.channels_last) or (c.k == 1 and wsrc.is_contiguous())
            w, w_t = ops.split_weights(d, wsrc, need_transposed=tape is not None)
        elif bf:
            # one cast of the fp32 master weights per step: KRSC for fprop, CRSK (transposed) for backward-data
            wsrc = self.p[c.name + ".weight"].detach()
            assert wsrc.is_contiguous(memory_format=torch.channels_last) or (c.k == 1 and wsrc.is_contiguous())
            w, w_t = ops.cast_weights_bf16(d, wsrc, c.cin, need_transposed=(tape is not None and c.cin != 3))
        else:
            w, w_t = self._weight(c), None
        y = torch.empty(G, N, d.ho, d.wo, c.cout, dtype=self.act_dtype, device=dev)
        rows = N * d.ho * d.wo
        gamma, beta = self.p[c.bn + ".weight"].detach(), self.p[c.bn + ".bias"].detach()
        rm, rv = self.p[c.bn + ".running_mean"], self.p[c.bn + ".running_var"]
        aff = torch.empty(4, G, c.cout, dtype=torch.float32, device=dev)
        mean, invstd, scale, shift = aff[0], aff[1], aff[2], aff[3]

        def fprop(stats_buf):
            if stem_rw and bf:
                ops.stem_fprop_bf16(d, x, w, y, stats_buf)
            elif stem_rw:
                ops.stem_fprop_split(d, x, w, y, stats_buf)
            elif sp_in:
                ops.conv_fprop_split(d, x, w, y, stats_buf)
            else:
                ops.conv_fprop(d, x, w, y, None, False, stats_buf)
        if training:
            if stem_rw and bf:   # two output columns per GEMM row: twice the partials of a forward over n x ho x wo/2 rows
                P, rpp = ops.conv_stats_partials(ConvDesc.make(G, N, d.ho, d.wo // 2, 64, 2 * c.cout, 1, 1, 0), True)
                P *= 2
            elif stem_rw:    # the same row tiles as any split forward with these n, ho, wo
                P, rpp = ops.conv_stats_partials_split(ConvDesc.make(G, N, d.ho, d.wo, 32, c.cout, 1, 1, 0))
            else:
                P, rpp = ops.conv_stats_partials_split(d) if sp_in else ops.conv_stats_partials(d, bf)
            stats = torch.empty(G, P, 2, c.cout, dtype=torch.float32, device=dev)
            fprop(stats)
            ops.bn_finalize(stats, G, P, rpp, rows, c.cout, gamma, beta, rm, rv, BN_MOMENTUM, BN_EPS, mean, invstd,
                            scale, shift)
        elif tape is None and not bf:
            # inference: BN (running statistics) + residual + ReLU folded into the conv epilogue
            ops.bn_eval_affine(1, c.cout, gamma, beta, rm, rv, BN_EPS, scale[:1], shift[:1])
            sp_eval = self._split_now and self.split_eval
            if sp_eval and c.cin != 3:
                # ... on the split kernels: the epilogue writes the next conv's sp operand directly (the downsample
                # branch, read only as a residual, stays fp32)
                wsrc = self.p[c.name + ".weight"].detach()
                assert wsrc.is_contiguous(memory_format=torch.channels_last) or (c.k == 1 and wsrc.is_contiguous())
                # the sp copy of the weights is kept between calls while the parameter is unchanged (its version counter:
                # load_state_dict, optimizer steps - the fused Adam bumps it explicitly - and broadcasts all move it)
                hit = self._wk_cache.get(c.name)
                if hit is not None and hit[0] == wsrc.data_ptr() and hit[1] == wsrc._version:
                    wk = hit[2]
                else:
                    wk, _ = ops.split_weights(d, wsrc, need_transposed=False)
                    self._wk_cache[c.name] = (wsrc.data_ptr(), wsrc._version, wk)
                out = ops.sp_empty(G, N, d.ho, d.wo, c.cout, device=dev) if relu else y
                ops.conv_fprop_split_affine(d, x, wk, out, scale[0], shift[0], residual, relu)
                return out
            ops.conv_fprop_affine(d, x, w, y, scale[0], shift[0], residual, relu)
            if pool:
                pooled, am = self._pool_plain(y, G, N, d.ho, d.wo, c.cout)
                return (ops.split_f32(pooled), am) if sp_eval else (pooled, am)
            return y
        else:
            fprop(None)
            ops.bn_eval_affine(G, c.cout, gamma, beta, rm, rv, BN_EPS, scale, shift)
        keep = tape is not None
        if defer_apply:
            assert not relu and residual is None and not pool
            if keep:
                u = _Unit()
                u.spec, u.desc, u.x_in, u.y, u.out, u.mean, u.invstd, u.relu, u.rows, u.w = \
                    c, d, x, y, None, mean, invstd, False, rows, (w_t if (bf or sp_in) else w)
                u.trained = training
                u.split = sp_in
                u.relu_affine = None
                tape.append(u)
            return y, (scale, shift)
        if pool:
            assert relu and residual is None
            hp, wp_ = (d.ho + 2 - 3) // 2 + 1, (d.wo + 2 - 3) // 2 + 1
            argmax = torch.empty(G, N, hp, wp_, c.cout, dtype=torch.uint8, device=dev)
            if sp_out:
                out = ops.sp_empty(G, N, hp, wp_, c.cout, device=dev)
                ops.bn_relu_maxpool_fwd_split(y, scale, shift, out, argmax, G, N, d.ho, d.wo, c.cout, hp, wp_)
            else:
                out = torch.empty(G, N, hp, wp_, c.cout, dtype=self.act_dtype, device=dev)
                ops.bn_relu_maxpool_fwd(y, scale, shift, out, argmax, G, N, d.ho, d.wo, c.cout, hp, wp_)
        elif sp_out:
            out = ops.sp_empty(G, N, d.ho, d.wo, c.cout, device=dev)
            bits = ops.bn_apply_split(y, scale, shift, residual, relu, out, G, rows, c.cout, residual_affine,
                                      want_bits=keep and relu and residual is not None)
        else:
            out = torch.empty_like(y) if keep else y            # inference: normalise in place
            bits = None
            if keep and relu and residual is not None and self.relu_bits:
                bits = ops.bn_apply_bits(y, scale, shift, residual, out, G, rows, c.cout, residual_affine)
            else:
                ops.bn_apply(y, scale, shift, residual, relu, out, G, rows, c.cout, residual_affine)
        if keep:
            u = _Unit()
            u.spec, u.desc, u.x_in, u.y, u.out, u.mean, u.invstd, u.relu, u.rows, u.w = \
                c, d, x, y, (None if pool else out), mean, invstd, relu, rows, (w_t if (bf or sp_in) else w)
            u.trained = training
            u.split = sp_in
            u.stem_rw = stem_rw
            # ReLU without residual: the backward rebuilds the mask from y (saves reading `out` twice)
            u.relu_affine = (scale, shift) if (relu and residual is None and not pool) else None
            if pool:
                u.pool = (argmax, scale, shift, d.ho, d.wo, hp, wp_)
            else:
                u.relu_bits = bits
            tape.append(u)
        return (out, argmax) if pool else out

    @staticmethod
    def _pool_plain(a0: Tensor, G: int, N: int, h: int, w: int, c: int):
        hp, wp_ = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        x = torch.empty(G, N, hp, wp_, c, dtype=torch.float32, device=a0.device)
        argmax = torch.empty(G, N, hp, wp_, c, dtype=torch.uint8, device=a0.device)
        ops.maxpool_fwd(a0, x, argmax, G * N, h, w, c, hp, wp_)
        return x, argmax

    def forward(self, imgs: List[Tensor], training: bool, keep_tape: bool, input_bgr: bool = False,
                input_size: Optional[int] = None, need_dimg: bool = False):
        """imgs: V tensors [B,3,H,W] fp32 NCHW (the reference's input format, rot_mv.py:188-189), or
        V raw uint8 [B,H,W,3] face patches, put through test_transform of main.py:50-55 on the GPU
        (ToTensor, Resize((input_size, input_size), antialias=True) when the patch has another size,
        Normalize).  Returns (img_feat [V,B,fc_dim], tape or None)."""
        V = len(imgs)
        raw = imgs[0].dtype == torch.uint8
        if raw:
            B, Hin, Win, C = imgs[0].shape
            H, W = (input_size, input_size) if input_size else (Hin, Win)
        else:
            B, C, H, W = imgs[0].shape
        assert C == 3
        dev = imgs[0].device
        if self.bf16 and raw:
            raise NotImplementedError("raw uint8 input with the bf16 path: normalise to fp32 NCHW first")
        # The split kernels address one view of an sp tensor (4 bytes per element) with 32-bit offsets: the largest one
        # (layer1's output: (H/4) x (W/4) x 64 or 256 channels per image) must stay below 2 GiB, or this call runs on the
        # fp32-MFMA kernels (64-bit row offsets there; B < 668 per view at 224 x 224 with ResNet-50)
        biggest_view_elems = B * ((H + 3) // 4) * ((W + 3) // 4) * self.spec.blocks[0].convs[-1].cout
        self._split_now = self.split and 4 * biggest_view_elems * self._guard_scale < 0x7FFFFFF0
        # the stem's form for this call (see stem_rowwindow)
        if self.bf16:        # folded windows: width % 4, whole 64-row partials (n * ho * wo / 2), the stem's weights through the batch
            ho, wo = (H - 1) // 2 + 1, W // 2
            self._stem_rw = (self.stem_rowwindow and training and W % 4 == 0 and (B * ho * (wo // 2)) % 64 == 0 and self.batch_weight_prep
                             and self.spec.stem.k == 7)
        else:
            self._stem_rw = (self._split_now and self.stem_rowwindow and training and not need_dimg and W % 2 == 0
                             and self.batch_weight_prep and 32 * B * H * (W // 2) * 4 < 0x7FFFFFF0)
        direct = self._stem_rw and not raw              # windows straight from the NCHW input: no NHWC image is built
        if direct and self.bf16:
            x0 = torch.empty(V, B, H, W // 4, 64, dtype=self.act_dtype, device=dev)
        elif direct:
            x0 = ops.sp_empty(V, B, H, W // 2, 32, device=dev)
            x0.sinv = None
        else:
            x0 = torch.empty(V, B, H, W, 8 if self.bf16 else 4, dtype=self.act_dtype, device=dev)
        for v, im in enumerate(imgs):
            assert im.shape == imgs[0].shape and im.is_cuda and im.dtype == imgs[0].dtype
            if direct:
                assert im.dtype == torch.float32
                if self.bf16:
                    ops.stem_rowwindow_bf16(im.detach().contiguous(), x0[v])
                else:
                    ops.stem_rowwindow_split_nchw(im.detach().contiguous(), x0[v])
            elif self.bf16:
                assert im.dtype == torch.float32
                ops.nchw_to_nhwc8_bf16(im.detach().contiguous(), x0[v], B, 3, H, W)
            elif raw:
                ops.preprocess_u8hwc_resize(im.contiguous(), x0[v], B, Hin, Win, H, W, IMAGE_MEAN, IMAGE_STD, input_bgr)
            else:
                assert im.dtype == torch.float32
                ops.nchw_to_nhwc4(im.detach().contiguous(), x0[v], B, 3, H, W)
        if self._stem_rw and not direct:                # raw uint8 input: windows from the normalised NHWC4 image
            x0 = ops.stem_rowwindow_split(x0)
        self._wprep = None
        if training:
            self._wk_cache.clear()               # the weights are about to change: drop the inference copies
        if self.batch_weight_prep and (self.bf16 or (self._split_now and training)):
            self._wprep = self._prepare_weights(dev)
        tape: Optional[dict] = {"units": [], "blocks": [], "V": V, "B": B} if keep_tape else None
        if keep_tape and self._wprep is not None:
            tape["wprep_versions"] = self._wprep_versions
        ulist = tape["units"] if keep_tape else None
        if training:
            torch._foreach_add_(self.bn_count_buffers(), V)       # num_batches_tracked += 1 per view call
        s = self.spec
        x, argmax = self._unit_fwd(s.stem, x0, V, B, H, W, training, True, None, ulist, pool=True)
        Hc, Wc = x.shape[2], x.shape[3]
        for blk in s.blocks:
            first = len(ulist) if keep_tape else 0
            identity = x
            out = x
            h, w = Hc, Wc
            for c in blk.convs[:-1]:
                out = self._unit_fwd(c, out, V, B, h, w, training, True, None, ulist)
                h, w = out.shape[2], out.shape[3]
            ds_idx = None
            ident_affine = None
            if blk.downsample is not None:
                if training or keep_tape or self.bf16:
                    # raw downsample conv output + its (scale, shift): normalised inside the last unit's bn_apply
                    identity, ident_affine = self._unit_fwd(blk.downsample, x, V, B, Hc, Wc, training, False, None, ulist,
                                                            defer_apply=True)
                else:
                    identity = self._unit_fwd(blk.downsample, x, V, B, Hc, Wc, training, False, None, ulist)
                ds_idx = len(ulist) - 1 if keep_tape else None
            out = self._unit_fwd(blk.convs[-1], out, V, B, h, w, training, True, identity, ulist,
                                 residual_affine=ident_affine)
            if keep_tape:
                n_main = len(blk.convs)
                idx = list(range(first, first + n_main - 1)) + [len(ulist) - 1]
                tape["blocks"].append((idx, ds_idx))
            x = out
            Hc, Wc = out.shape[2], out.shape[3]
        feat = torch.empty(V, B, self.fc_dim, dtype=torch.float32, device=dev)
        if x.dtype == torch.float16:                                # sp activation of the split path
            ops.avgpool_fwd_split(x, feat, V * B, Hc * Wc, self.fc_dim)
        else:
            ops.avgpool_fwd(x, feat, V * B, Hc * Wc, self.fc_dim)
        if keep_tape:
            tape["final_hw"] = (Hc, Wc)
        return feat, tape

    # ---------------------------------------------------------------- backward
    def _bn_bwd(self, u: _Unit, g: Tensor, need_dz: bool, sink: GradSink):
        """g = grad wrt the unit's output.  Returns (dy, dz): dy = grad wrt the conv output;
        dz = g masked by the unit's ReLU (written in place into g) when the residual branch needs it."""
        c = u.spec
        G = u.y.shape[0]
        gp, bp = self.p[c.bn + ".weight"], self.p[c.bn + ".bias"]
        if u.fused_s12 is not None and u.split:
            # split path, fused: g arrived masked and the sums came with it; dy goes out in sp
            (s12, sinv), u.fused_s12 = u.fused_s12, None
            dy = ops.sp_empty(*u.y.shape, device=g.device)
            ops.bn_bwd_apply_split(g, u.y, u.mean, u.invstd, gp.detach(), s12[0], s12[1], G, u.rows, c.cout, dy, None, s12[2], sinv)
            return dy, (g if need_dz else None)
        if u.fused_s12 is not None:
            # g arrived masked by this unit's ReLU and its sums (incl. dgamma / dbeta) came with it
            s12, u.fused_s12 = u.fused_s12, None
            dy = torch.empty_like(g) if need_dz else g
            ops.bn_bwd_apply(g, None, u.y, u.mean, u.invstd, gp.detach(), s12[0], s12[1], G, u.rows, c.cout, dy, None, None)
            return dy, (g if need_dz else None)
        s12 = torch.empty(3 if u.split else 2, G, c.cout, dtype=torch.float32, device=g.device)     # s1, s2 (, max |dz| per channel)
        ra = u.relu_affine
        act = u.out if (u.relu and ra is None) else None
        acc = sink.accumulate(gp)
        assert acc == sink.accumulate(bp)
        if u.split:
            # split path: g and y are fp32, dy goes to the conv kernels in sp; residual units carry their mask as bits
            assert not (u.relu and ra is None) or u.relu_bits is not None
            sinv = torch.empty(1, dtype=torch.float32, device=g.device)       # dy's 2^-k: left by the reduce pass's finalize launch
            if u.relu_bits is not None:
                ops.bn_bwd_reduce_split(g, u.relu_bits, u.y, u.mean, u.invstd, G, u.rows, c.cout, s12[0], s12[1], sink.view(gp),
                                        sink.view(bp), acc, s12[2], None, dz_out=g, gamma=gp.detach(), dy_sinv=sinv)
            else:
                ops.bn_bwd_reduce_split(g, None, u.y, u.mean, u.invstd, G, u.rows, c.cout, s12[0], s12[1], sink.view(gp),
                                        sink.view(bp), acc, s12[2], ra, gamma=gp.detach(), dy_sinv=sinv)
            dy = ops.sp_empty(*u.y.shape, device=g.device)
            ops.bn_bwd_apply_split(g, u.y, u.mean, u.invstd, gp.detach(), s12[0], s12[1], G, u.rows, c.cout, dy,
                                   None if u.relu_bits is not None else ra, s12[2], sinv)
            return dy, (g if need_dz else None)
        if need_dz:
            # the reduce pass writes the masked gradient dz over g: the apply pass then reads (dz, y) only - no
            # second look at the ReLU mask, no second dz store - and the residual branch takes dz from g
            if u.relu_bits is not None:
                ops.bn_bwd_reduce_bits(g, u.relu_bits, u.y, u.mean, u.invstd, G, u.rows, c.cout, s12[0], s12[1], sink.view(gp),
                                       sink.view(bp), acc, dz_out=g)
            else:
                ops.bn_bwd_reduce(g, act, u.y, u.mean, u.invstd, G, u.rows, c.cout, s12[0], s12[1], sink.view(gp),
                                  sink.view(bp), acc, ra, dz_out=g)
            dy = torch.empty_like(g)
            ops.bn_bwd_apply(g, None, u.y, u.mean, u.invstd, gp.detach(), s12[0], s12[1], G, u.rows, c.cout, dy, None, None)
            return dy, g
        ops.bn_bwd_reduce(g, act, u.y, u.mean, u.invstd, G, u.rows, c.cout, s12[0], s12[1], sink.view(gp), sink.view(bp),
                          acc, ra)
        ops.bn_bwd_apply(g, act, u.y, u.mean, u.invstd, gp.detach(), s12[0], s12[1], G, u.rows, c.cout, g, None, ra)
        return g, None

    def _side(self, dev) -> "torch.cuda.Stream":
        if self._wg_stream is None or self._wg_stream.device != dev or self._wg_low != self.wgrad_low_priority:
            old = self._wg_stream if (self._wg_stream is not None and self._wg_stream.device == dev) else None
            self._wg_low = self.wgrad_low_priority
            try:
                if not self._wg_low:
                    raise RuntimeError
                self._wg_stream = ops.low_priority_stream(dev)   # fills what the critical path leaves idle
            except RuntimeError:                                 # ordinary side stream (also: no priority support)
                self._wg_stream = torch.cuda.Stream(device=dev)
            if old is not None:
                self._wg_stream.wait_stream(old)                 # whoever joins the new stream also joins the old one's work
            self.grad_streams[:] = [self._wg_stream]
        return self._wg_stream

    def _conv_bwd(self, u: _Unit, dy: Tensor, need_dx: bool, addend: Optional[Tensor], sink: GradSink,
                  fuse_for: Optional[_Unit] = None):
        if self.overlap_wgrad and dy.is_cuda:
            side = self._side(dy.device)
            side.wait_stream(torch.cuda.current_stream())         # dy (and everything before it) is ready
            with torch.cuda.stream(side):
                self._wgrad(u, dy, sink)
            if getattr(dy, "sinv", None) is not None:
                dy.sinv.record_stream(side)
            dy.record_stream(side)                                # the allocator must not recycle these while
            u.x_in.record_stream(side)                            # the side stream still reads them
        else:
            self._wgrad(u, dy, sink)
        dx = None
        if need_dx:
            dx = torch.empty(ops.sp_shape(u.x_in), dtype=torch.float32, device=dy.device) if u.split else torch.empty_like(u.x_in)
            self._dgrad(u, dy, dx, addend, fuse_for, sink)
        return dx

    def _wgrad(self, u: _Unit, dy: Tensor, sink: GradSink):
        c = u.spec
        wp = self.p[c.name + ".weight"]
        if u.split:
            # (slabs now, their sums in ONE launch per residual block: _flush_wgrad_reduces)
            ops.conv_wgrad_split(u.desc, u.x_in, dy, sink.view(wp), sink.accumulate(wp), defer=self._wg_defer)
        elif u.stem_rw and self.bf16:
            dw16 = torch.empty(2 * c.cout, 7, 16, 4, dtype=torch.float32, device=dy.device)
            ops.stem_wgrad_bf16(u.desc, u.x_in, dy, dw16, False)
            gv = sink.view(wp).permute(0, 2, 3, 1)                    # [cout, 7, 7, 3] view of the gradient
            if sink.accumulate(wp):
                gv.add_(dw16[:c.cout, :, 1:8, :3])
            else:
                gv.copy_(dw16[:c.cout, :, 1:8, :3])
            gv.add_(dw16[c.cout:, :, 3:10, :3])                       # the odd output columns' taps
        elif u.stem_rw:
            dw8 = torch.empty(c.cout, 7, 8, 4, dtype=torch.float32, device=dy.device)
            ops.stem_wgrad_split(u.desc, u.x_in, dy, dw8, False)
            gv = sink.view(wp).permute(0, 2, 3, 1)                    # [cout, 7, 7, 3] view of the gradient
            if sink.accumulate(wp):
                gv.add_(dw8[:, :, 1:, :3])
            else:
                gv.copy_(dw8[:, :, 1:, :3])
        elif c.cin == 3:
            dw4 = torch.empty(c.cout, c.k, c.k, u.desc.cin, dtype=torch.float32, device=dy.device)
            ops.conv_wgrad(u.desc, u.x_in, dy, dw4, False)
            gv = sink.view(wp).permute(0, 2, 3, 1)
            if sink.accumulate(wp):
                gv.add_(dw4[..., :3])
            else:
                gv.copy_(dw4[..., :3])
        else:
            ops.conv_wgrad(u.desc, u.x_in, dy, sink.view(wp), sink.accumulate(wp), defer=self._wg_defer if self.bf16 else None)

    def _flush_wgrad_reduces(self, dev):
        """The slab sums of the weight gradients launched since the last flush, in one launch on the stream that wrote the slabs."""
        if not self._wg_defer:
            return
        if self.overlap_wgrad and dev.type == "cuda":
            with torch.cuda.stream(self._side(dev)):
                ops.wgrad_reduce_batch(self._wg_defer)
        else:
            ops.wgrad_reduce_batch(self._wg_defer)

    def _dgrad(self, u: _Unit, dy: Tensor, dx: Tensor, addend: Optional[Tensor], fuse_for: Optional[_Unit] = None,
               sink: Optional[GradSink] = None):
        """dx = backward-data of unit u (+ addend).  fuse_for = the unit whose OUTPUT gradient dx is, when dx
        is final with this launch: its ReLU mask is applied and its BatchNorm-backward sums (s1, s2, dgamma,
        dbeta) are produced by the same launch (mvg_conv_dgrad_split_bnreduce / mvg_conv_dgrad_bf16_bnreduce; stride-2 launches
        too: every parity class brings its partials) instead of a pass over (g, act, y)."""
        U = fuse_for
        if (self.bf16 and U is not None and self.fuse_bn_split and u.spec.cin != 3 and u.desc.cin % 64 == 0 and u.desc.cout % 64 == 0
                and (not U.relu or U.relu_bits is not None or U.relu_affine is not None)):     # (relu_bits off: mask from the activation)
            c = U.spec
            gp, bp = self.p[c.bn + ".weight"], self.p[c.bn + ".bias"]
            acc = sink.accumulate(gp)
            assert acc == sink.accumulate(bp)
            s12 = torch.empty(2, dx.shape[0], c.cout, dtype=torch.float32, device=dx.device)
            ops.conv_dgrad_bf16_bnreduce(u.desc, dy, u.w, dx, addend, U.y, U.relu_bits, U.mean, U.invstd,
                                         None if U.relu_bits is not None else U.relu_affine, s12[0], s12[1], sink.view(gp),
                                         sink.view(bp), acc)
            U.fused_s12 = s12
            return
        if u.split:
            if U is not None and U.split and self.fuse_bn_split:
                c = U.spec
                gp, bp = self.p[c.bn + ".weight"], self.p[c.bn + ".bias"]
                acc = sink.accumulate(gp)
                assert acc == sink.accumulate(bp)
                s12 = torch.empty(3, dx.shape[0], c.cout, dtype=torch.float32, device=dx.device)     # s1, s2, max |dz| per channel
                sinv = torch.empty(1, dtype=torch.float32, device=dx.device)      # 2^-k of the dy that U's apply pass will write
                ops.conv_dgrad_split_bnreduce(u.desc, dy, u.w, dx, addend, U.y, U.relu_bits, U.mean, U.invstd,
                                              None if U.relu_bits is not None else U.relu_affine, s12[0], s12[1], sink.view(gp),
                                              sink.view(bp), acc, s12[2], gp.detach(), sinv)
                U.fused_s12 = (s12, sinv)
                return
            ops.conv_dgrad_split(u.desc, dy, u.w, dx, addend)
            return
        ops.conv_dgrad(u.desc, dy, u.w, dx, None, addend)

    def backward(self, tape: dict, dfeat: Tensor, sink: GradSink, need_dimg: bool = False):
        """dfeat [V,B,fc_dim] -> parameter gradients into ``sink`` (published layer4 ... stem, the
        order they become final); returns d(img) as V NCHW tensors when need_dimg."""
        V, B = tape["V"], tape["B"]
        units: List[_Unit] = tape["units"]
        if not all(u.trained for u in units):
            raise NotImplementedError("backward through eval-mode BatchNorm (running statistics) is not implemented: "
                                      "call model.train() for gradient steps")
        Hc, Wc = tape["final_hw"]
        if need_dimg and self.bf16:
            raise NotImplementedError("d(loss)/d(img) is not produced by the bf16 path")
        if tape.get("wprep_versions") is not None and tape["wprep_versions"] != getattr(self, "_wprep_versions", None):
            raise RuntimeError("backward of a tape whose bf16 / sp weight copies were overwritten by a later forward with "
                               "DIFFERENT weights (forward, optimizer step, forward, then backward of the first call): "
                               "backward-data would run with the new weights.  Run backward before the weights change "
                               "(PyTorch raises its version-counter error in the same situation)")
        g = torch.empty(V, B, Hc, Wc, self.fc_dim, dtype=self.act_dtype, device=dfeat.device)
        ops.avgpool_bwd(dfeat.contiguous(), g, V * B, Hc * Wc, self.fc_dim)
        self._wg_defer = []
        P = self.p
        blocks = tape["blocks"]
        for bi in range(len(blocks) - 1, -1, -1):
            idx, ds_idx = blocks[bi]
            last = units[idx[-1]]
            # the unit that receives this block's input gradient: the previous block's last unit (the stem's
            # fused pool backward takes it for the first block)
            prev_last = units[blocks[bi - 1][0][-1]] if bi > 0 else None
            done: List[torch.nn.Parameter] = []
            dy, dz = self._bn_bwd(last, g, True, sink)
            d = self._conv_bwd(last, dy, True, None, sink, fuse_for=units[idx[-2]])
            done += [P[last.spec.name + ".weight"], P[last.spec.bn + ".weight"], P[last.spec.bn + ".bias"]]
            last.y = last.out = None
            del dy
            for k in range(len(idx) - 2, -1, -1):
                u = units[idx[k]]
                dy, _ = self._bn_bwd(u, d, False, sink)
                if k == 0:
                    # with a downsample branch the block-input gradient is final only after that branch's launch
                    d = self._conv_bwd(u, dy, True, dz if ds_idx is None else None, sink,
                                       fuse_for=prev_last if ds_idx is None else None)
                else:
                    d = self._conv_bwd(u, dy, True, None, sink, fuse_for=units[idx[k - 1]])
                done += [P[u.spec.name + ".weight"], P[u.spec.bn + ".weight"], P[u.spec.bn + ".bias"]]
                u.y = u.out = None
            if ds_idx is not None:
                ud = units[ds_idx]
                dyd, _ = self._bn_bwd(ud, dz, False, sink)
                self._conv_bwd(ud, dyd, False, None, sink)                # wgrad only
                self._dgrad(ud, dyd, d, d, prev_last, sink)               # d += dgrad (aliasing addend): now final
                done += [P[ud.spec.name + ".weight"], P[ud.spec.bn + ".weight"], P[ud.spec.bn + ".bias"]]
                ud.y = ud.out = None
            self._flush_wgrad_reduces(g.device)               # this block's weight gradients are final once their slabs are summed
            sink.publish(done)
            g = d
            if "debug" in tape:
                tape["debug"].append(g.clone())
        # stem: max pool + ReLU + BatchNorm backward fused (the 112x112 gradient map is never built)
        stem = units[0]
        argmax, scale, shift, H1, W1, Hp, Wp = stem.pool
        sc = stem.spec
        gp, bp = P[sc.bn + ".weight"], P[sc.bn + ".bias"]
        stem_sp = stem.stem_rw and not self.bf16
        s12 = torch.empty(3 if stem_sp else 2, V, sc.cout, dtype=torch.float32, device=g.device)
        acc = sink.accumulate(gp)
        assert acc == sink.accumulate(bp)
        if stem_sp:
            # the stem's weight gradient runs on the split kernels: dy goes out in sp, scaled by a bound from the reduce pass
            sinv = torch.empty(1, dtype=torch.float32, device=g.device)
            ops.bn_relu_maxpool_bwd_reduce_split(g, argmax, stem.y, stem.mean, stem.invstd, scale, shift, V, B, H1, W1, sc.cout, Hp, Wp,
                                                 s12[0], s12[1], sink.view(gp), sink.view(bp), acc, s12[2], gp.detach(), sinv)
            dy = ops.sp_empty(*stem.y.shape, device=g.device)
            ops.bn_relu_maxpool_bwd_apply_split(g, argmax, stem.y, stem.mean, stem.invstd, gp.detach(), scale, shift, s12[0], s12[1],
                                                V, B, H1, W1, sc.cout, Hp, Wp, dy, s12[2], sinv)
        else:
            ops.bn_relu_maxpool_bwd_reduce(g, argmax, stem.y, stem.mean, stem.invstd, scale, shift, V, B, H1, W1, sc.cout, Hp, Wp,
                                           s12[0], s12[1], sink.view(gp), sink.view(bp), acc)
            dy = torch.empty_like(stem.y)
            ops.bn_relu_maxpool_bwd_apply(g, argmax, stem.y, stem.mean, stem.invstd, gp.detach(), scale, shift, s12[0], s12[1],
                                          V, B, H1, W1, sc.cout, Hp, Wp, dy)
        dx0 = self._conv_bwd(stem, dy, need_dimg, None, sink)
        self._flush_wgrad_reduces(g.device)
        self._wg_defer = None
        sink.publish([P[stem.spec.name + ".weight"], P[stem.spec.bn + ".weight"], P[stem.spec.bn + ".bias"]])
        if self._wg_stream is not None and dy.is_cuda and (self.overlap_wgrad or not torch.cuda.is_current_stream_capturing()):
            torch.cuda.current_stream().wait_stream(self._wg_stream)      # gradients complete for the optimizer
        if not need_dimg:
            return None
        H, W = dx0.shape[2], dx0.shape[3]
        outs = []
        for v in range(V):
            o = torch.empty(B, 3, H, W, dtype=torch.float32, device=dx0.device)
            ops.nhwc4_to_nchw(dx0[v], o, B, 3, H, W)
            outs.append(o)
        return outs
