"""Tensor-level wrappers over the C ABI: raw device pointers + the current HIP stream.

PyTorch is used here for device memory and streams only; every computation is a launch of a
hand-written gfx950 kernel inside librotmvgaze_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from ._lib import ConvDesc, K_FAMILIES, ProfEntry, check, lib

Tensor = torch.Tensor


def _p(t: Optional[Tensor]):
    """Device pointer of t for a `void *` argument (ctypes converts the int; None = NULL).  Kept minimal: a ResNet-50 step makes
    ~6000 of these from the host."""
    if t is None:
        return None
    assert t.is_cuda and t.dtype in _PTR_DTYPES, (t.device, t.dtype)
    return t.data_ptr()


_PTR_DTYPES = frozenset((torch.float32, torch.int32, torch.uint8, torch.int16, torch.bfloat16, torch.float16))
_workspaces = {}      # (device index, stream handle) -> the uint8 tensor registered with mvg_set_scratch (insertion = LRU order)
_MAX_WORKSPACES = 8
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the current stream's handle without building a Stream object


def _s(scratch: bool = False):
    """The current HIP stream as the library's `void *stream`.  scratch=True (the entry points that can use scratch:
    fp32 conv / Linear fprop and dgrad - stream-K pieces - and bn_finalize): the first such launch on a
    (device, stream) allocates that stream's workspace from PyTorch's caching allocator and registers it
    (mvg_set_scratch) - the library itself never allocates device memory ("caller owns every buffer
    incl. workspace").  At most _MAX_WORKSPACES stay registered; the least recently used one is dropped (uses are
    stream-ordered and the tensor was allocated on that stream, so queued kernels are safe)."""
    if _raw_stream is not None:
        dev = torch.cuda.current_device()
        handle = _raw_stream(dev)
    else:
        st = torch.cuda.current_stream()
        dev, handle = st.device_index, st.cuda_stream
    if scratch:
        key = (dev, handle)
        ws = _workspaces.pop(key, None)
        if ws is None:
            while len(_workspaces) >= _MAX_WORKSPACES:
                (odev, ohandle), _old = next(iter(_workspaces.items()))
                with torch.cuda.device(odev):
                    check(lib().mvg_set_scratch(None, 0, C.c_void_p(ohandle)), "set_scratch")
                del _workspaces[(odev, ohandle)]
            with torch.cuda.device(dev):
                ws = torch.empty(int(lib().mvg_scratch_bytes()), dtype=torch.uint8, device=torch.device("cuda", dev))
                check(lib().mvg_set_scratch(C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(handle)), "set_scratch")
        _workspaces[key] = ws                      # (re)insert as most recently used
    return handle


def release_workspaces():
    """Drop every registered scratch workspace (the library forgets the pointers before PyTorch frees them)."""
    for (dev, handle), ws in list(_workspaces.items()):
        with torch.cuda.device(dev):
            check(lib().mvg_set_scratch(None, 0, C.c_void_p(handle)), "set_scratch")
    _workspaces.clear()


def _f32c(t: Tensor) -> Tensor:
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), "expected contiguous fp32 device tensor"
    return t


def _fn(name: str, t: Tensor):
    """The entry point for t's storage type: activations are fp32 (parity path) or bf16 (config C5's path)."""
    return getattr(lib(), name + "_bf16" if t.dtype == torch.bfloat16 else name)


# ---------------------------------------------------------------- conv / linear
def conv_stats_partials(d: ConvDesc, bf16: bool = False):
    rpp = C.c_int32(0)
    n = (lib().mvg_conv_stats_partials_bf16 if bf16 else lib().mvg_conv_stats_partials)(C.byref(d), C.byref(rpp))
    if n < 0:
        check(1, "conv_stats_partials")
    return n, rpp.value


def conv_fprop(d: ConvDesc, x: Tensor, w: Tensor, y: Tensor, bias: Optional[Tensor] = None, relu: bool = False,
               stats: Optional[Tensor] = None):
    check(_fn("mvg_conv_fprop", x)(C.byref(d), _p(x), _p(w), _p(y), _p(bias), int(relu), _p(stats), _s(True)), "conv_fprop")


def conv_fprop_affine(d: ConvDesc, x: Tensor, w: Tensor, out: Tensor, scale: Tensor, shift: Tensor,
                      residual: Optional[Tensor], relu: bool):
    check(lib().mvg_conv_fprop_affine(C.byref(d), _p(x), _p(w), _p(out), _p(scale), _p(shift), _p(residual), int(relu),
                                      _s(True)), "conv_fprop_affine")


def conv_dgrad(d: ConvDesc, dy: Tensor, w: Tensor, dx: Tensor, mask: Optional[Tensor] = None,
               addend: Optional[Tensor] = None):
    check(_fn("mvg_conv_dgrad", dy)(C.byref(d), _p(dy), _p(w), _p(dx), _p(mask), _p(addend), _s(True)), "conv_dgrad")


def cast_weights_bf16(d: ConvDesc, w: Tensor, cin_src: int, need_transposed: bool = True):
    """fp32 KRSC weights -> (bf16 KRSC [cout][r][s][d.cin], bf16 CRSK [d.cin][r][s][cout] or None)."""
    wk = torch.empty(d.cout, d.r, d.s, d.cin, dtype=torch.bfloat16, device=w.device)
    wt = torch.empty(d.cin, d.r, d.s, d.cout, dtype=torch.bfloat16, device=w.device) if need_transposed else None
    check(lib().mvg_cast_weights_bf16(C.byref(d), _p(w), cin_src, _p(wk), _p(wt), _s()), "cast_weights_bf16")
    return wk, wt


def conv_wgrad(d: ConvDesc, x: Tensor, dy: Tensor, dw: Tensor, accumulate: bool = False, defer: Optional[list] = None):
    """defer (bf16 storage only): see conv_wgrad_split - slabs now, their sums in one wgrad_reduce_batch launch per residual block."""
    splits = _fn("mvg_conv_wgrad_splits", x)(C.byref(d))
    if splits < 1:
        check(1, "conv_wgrad_splits")
    ws = None
    if splits > 1:
        ws = torch.empty(splits * d.cout * d.r * d.s * d.cin, dtype=torch.float32, device=x.device)
    if defer is not None and splits > 1 and x.dtype == torch.bfloat16 and dw.numel() == d.cout * d.r * d.s * d.cin and dw.numel() % 4 == 0:
        check(lib().mvg_conv_wgrad_bf16_slabs(C.byref(d), _p(x), _p(dy), _p(ws), splits, _s()), "conv_wgrad_bf16_slabs")
        defer.append((ws, dw, splits, bool(accumulate)))
        return
    check(_fn("mvg_conv_wgrad", x)(C.byref(d), _p(x), _p(dy), _p(dw), _p(ws), splits, int(accumulate), _s()), "conv_wgrad")


def linear_wgrad(x: Tensor, dy: Tensor, dw: Tensor, db: Optional[Tensor], rows: int, fin: int, fout: int, accumulate: bool = False):
    """dw (+)= dy^T x and db (+)= column sums of dy in one launch (+ the fixed-order slab reduce)."""
    d = ConvDesc.linear(rows, fin, fout)
    splits = lib().mvg_conv_wgrad_splits(C.byref(d))
    if splits < 1:
        check(1, "conv_wgrad_splits")
    ws = torch.empty(splits * (fout * fin + fout), dtype=torch.float32, device=x.device) if splits > 1 else None
    check(lib().mvg_linear_wgrad(_p(x), _p(dy), _p(dw), _p(db), rows, fin, fout, _p(ws), splits, int(accumulate), _s()),
          "linear_wgrad")


def fuser_fprop(img_feat, feat, rel, row_img, row_src, w, bias, relu, y, rows, cf, nvec, fout):
    """y = [relu](W @ [img_feat[row_img] | rel @ feat[row_src]] + bias): the concatenated / rotated rows are built by
    the GEMM's operand loader (mvg_fuser_fprop) - no X tensor."""
    ws, n = _linear_ws(img_feat, rows, cf + 3 * nvec, fout)
    check(lib().mvg_fuser_fprop(_p(img_feat), _p(feat), _p(rel), _p(row_img), _p(row_src), _p(w), _p(bias), int(relu), _p(y), rows,
                                cf, nvec, fout, img_feat.numel() // cf, feat.numel() // (3 * nvec), _p(ws), n, _s(True)), "fuser_fprop")


def fuser_wgrad(img_feat, feat, rel, row_img, row_src, dy, dw, db, rows, cf, nvec, fout, accumulate=False):
    fin = cf + 3 * nvec
    d = ConvDesc.linear(rows, fin, fout)
    splits = lib().mvg_conv_wgrad_splits(C.byref(d))
    if splits < 1:
        check(1, "conv_wgrad_splits")
    ws = torch.empty(splits * (fout * fin + fout), dtype=torch.float32, device=dy.device) if splits > 1 else None
    check(lib().mvg_fuser_wgrad(_p(img_feat), _p(feat), _p(rel), _p(row_img), _p(row_src), _p(dy), _p(dw), _p(db), rows, cf, nvec,
                                fout, img_feat.numel() // cf, feat.numel() // (3 * nvec), _p(ws), splits, int(accumulate), _s()),
          "fuser_wgrad")


# ---- "split" operands: fp32 values as two fp16 pieces (+ a per-tensor power-of-two scale), fp32-accurate products on
# the fp16 matrix cores.  An sp tensor is a float16 tensor [..., C/8, 2, 8]; its scale travels as the attribute
# ``sinv`` - a 1-element fp32 DEVICE tensor holding 2^-k, or absent / None for unscaled tensors (activations).
def _sinv(t: Optional[Tensor]):
    return _p(getattr(t, "sinv", None)) if t is not None else None


def sp_empty(*shape, device) -> Tensor:
    """An uninitialised sp tensor for fp32 shape [..., C] (C % 8 == 0): float16 [..., C/8, 2, 8]."""
    assert shape[-1] % 8 == 0
    return torch.empty(*shape[:-1], shape[-1] // 8, 2, 8, dtype=torch.float16, device=device)


def sp_shape(x_sp: Tensor):
    return tuple(x_sp.shape[:-3]) + (x_sp.shape[-3] * 8,)


def is_sp(t: Optional[Tensor]) -> bool:
    return t is not None and t.dtype == torch.float16


def split_f32(x: Tensor, scale: float = 1.0) -> Tensor:
    """fp32 [..., C] (C % 8 == 0) -> sp tensor of x * scale (scale: a power of two that puts max |x| * scale below 65504)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 8 == 0
    out = sp_empty(*x.shape, device=x.device)
    check(lib().mvg_split_f32(_p(x), _p(out), x.numel(), float(scale), _s()), "split_f32")
    out.sinv = None if scale == 1.0 else torch.full((1,), 1.0 / scale, dtype=torch.float32, device=x.device)
    return out


def merge_sp(x: Tensor) -> Tensor:
    """sp -> fp32 (the sum of the pieces times the tensor's 2^-k; test plumbing: reads sinv on the host)."""
    assert is_sp(x) and x.is_contiguous() and x.shape[-1] == 8 and x.shape[-2] == 2
    out = torch.empty(sp_shape(x), dtype=torch.float32, device=x.device)
    sinv = getattr(x, "sinv", None)
    check(lib().mvg_merge_sp(_p(x), _p(out), out.numel(), float(sinv.item()) if sinv is not None else 1.0, _s()), "merge_sp")
    return out


_items_scratch = {}


def split_weights(d: ConvDesc, w: Tensor, need_transposed: bool = True):
    """fp32 KRSC weights -> (sp KRSC, sp CRSK or None), both carrying ``sinv`` (2^-k from max |w|)."""
    assert w.dtype == torch.float32 and w.is_cuda
    rs = d.r * d.s
    wk = sp_empty(d.cout, rs * d.cin, device=w.device)
    wt = sp_empty(d.cin, rs * d.cout, device=w.device) if need_transposed else None
    stat = torch.empty(2, dtype=torch.float32, device=w.device)
    items = torch.empty(64, dtype=torch.uint8, device=w.device)
    check(lib().mvg_split_weights(C.byref(d), _p(w), _p(wk), _p(wt), _p(stat), _p(items), _s()), "split_weights")
    wk.sinv = stat[1:2]
    if wt is not None:
        wt.sinv = stat[1:2]
    return wk, wt


def weights_prep_batch(table: Tensor, n: int, mode: int, blocks_per_item: int = 0):
    """table: [n, 6] int64 DEVICE tensor of (w, wk, wt, cout | rs << 32, cin | cin_pad << 32, stat) records (48 bytes each)."""
    assert table.is_cuda and table.dtype == torch.int64 and table.is_contiguous() and table.shape == (n, 6)
    check(lib().mvg_weights_prep_batch(C.c_void_p(table.data_ptr()), n, mode, blocks_per_item, _s()), "weights_prep_batch")


def conv_stats_partials_split(d: ConvDesc):
    rpp = C.c_int32(0)
    n = lib().mvg_conv_stats_partials_split(C.byref(d), C.byref(rpp))
    if n < 0:
        check(1, "conv_stats_partials_split")
    return n, rpp.value


def conv_fprop_split(d: ConvDesc, x_sp: Tensor, w_sp: Tensor, y: Tensor, stats: Optional[Tensor] = None):
    check(lib().mvg_conv_fprop_split(C.byref(d), _p(x_sp), _sinv(x_sp), _p(w_sp), _sinv(w_sp), _p(y), _p(stats), _s()), "conv_fprop_split")


def conv_fprop_split_affine(d: ConvDesc, x_sp: Tensor, w_sp: Tensor, out: Tensor, scale: Tensor, shift: Tensor,
                            residual: Optional[Tensor], relu: bool):
    """Inference forward on the split kernels, BatchNorm folded: out = relu?(conv * scale + shift (+ residual)).
    out / residual: fp32 tensors or (unscaled) sp tensors."""
    check(lib().mvg_conv_fprop_split_affine(C.byref(d), _p(x_sp), _sinv(x_sp), _p(w_sp), _sinv(w_sp), _p(out), int(is_sp(out)), _p(scale),
                                            _p(shift), _p(residual), int(is_sp(residual)), int(relu), _s()), "conv_fprop_split_affine")


def conv_dgrad_split(d: ConvDesc, dy_sp: Tensor, wt_sp: Tensor, dx: Tensor, addend: Optional[Tensor] = None,
                     relu_mask_sp: Optional[Tensor] = None):
    assert relu_mask_sp is None or (is_sp(relu_mask_sp) and getattr(relu_mask_sp, "sinv", None) is None)
    check(lib().mvg_conv_dgrad_split(C.byref(d), _p(dy_sp), _sinv(dy_sp), _p(wt_sp), _sinv(wt_sp), _p(dx), _p(addend), _p(relu_mask_sp),
                                     _s()), "conv_dgrad_split")


def stem_rowwindow_split(x_nhwc4: Tensor) -> Tensor:
    """[G, N, H, W, 4] fp32 image -> the stem's row-window operand [G, N, H, W/2, 32] in sp (mvg_stem_fprop_split)."""
    G, N, H, W, C = x_nhwc4.shape
    assert C == 4 and W % 2 == 0 and x_nhwc4.dtype == torch.float32 and x_nhwc4.is_contiguous()
    xw = sp_empty(G, N, H, W // 2, 32, device=x_nhwc4.device)
    check(lib().mvg_stem_rowwindow_split(_p(x_nhwc4), _p(xw), G * N, H, W, _s()), "stem_rowwindow_split")
    xw.sinv = None
    return xw


def stem_rowwindow_split_nchw(x_nchw: Tensor, out: Tensor):
    """[B, 3, H, W] fp32 NCHW -> out (one view of an sp_empty(V, B, H, W/2, 32)): the windows straight from the module's input."""
    B, Cc, H, W = x_nchw.shape
    assert Cc == 3 and W % 2 == 0 and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and is_sp(out) and out.is_contiguous()
    assert sp_shape(out) == (B, H, W // 2, 32)
    check(lib().mvg_stem_rowwindow_split_nchw(_p(x_nchw), _p(out), B, H, W, _s()), "stem_rowwindow_split_nchw")


def stem_fprop_split(d: ConvDesc, xw: Tensor, w_sp: Tensor, y: Tensor, stats: Optional[Tensor]):
    check(lib().mvg_stem_fprop_split(C.byref(d), _p(xw), _p(w_sp), _sinv(w_sp), _p(y), _p(stats), _s()), "stem_fprop_split")


def stem_wgrad_split(d: ConvDesc, xw: Tensor, dy_sp: Tensor, dw_rw: Tensor, accumulate: bool = False):
    """dw_rw [cout, 7, 8, 4] fp32 (+)= the stem's weight gradient in the row-window tap layout."""
    splits = lib().mvg_stem_wgrad_splits_split(C.byref(d))
    if splits < 1:
        check(1, "stem_wgrad_splits_split")
    ws = torch.empty(splits * d.cout * 224, dtype=torch.float32, device=xw.device) if splits > 1 else None
    check(lib().mvg_stem_wgrad_split(C.byref(d), _p(xw), _p(dy_sp), _sinv(dy_sp), _p(dw_rw), _p(ws), splits, int(accumulate), _s()),
          "stem_wgrad_split")


def bn_relu_maxpool_bwd_reduce_split(g_pooled, argmax, y, mean, invstd, scale, shift, groups, n_per_group, h, w, c, ho, wo, s1, s2,
                                     dgamma, dbeta, accumulate, mx, gamma=None, dy_sinv=None):
    """gamma + dy_sinv (a 1-element device tensor): the finalize launch also leaves the 2^-k of the dy the apply pass will write."""
    n = lib().mvg_bn_bwd_workspace_floats(groups, n_per_group * h * w, c)
    ws = torch.empty(n, dtype=torch.float32, device=y.device)
    check(lib().mvg_bn_relu_maxpool_bwd_reduce_split(_p(g_pooled), _p(argmax), _p(y), _p(mean), _p(invstd), _p(scale), _p(shift), groups,
                                                     n_per_group, h, w, c, ho, wo, _p(s1), _p(s2), _p(dgamma), _p(dbeta), int(accumulate),
                                                     _p(ws), _p(mx), _p(gamma), _p(dy_sinv), _s(True)), "bn_relu_maxpool_bwd_reduce_split")


def bn_relu_maxpool_bwd_apply_split(g_pooled, argmax, y, mean, invstd, gamma, scale, shift, s1, s2, groups, n_per_group, h, w, c, ho, wo,
                                    dy_sp, mx, dy_sinv=None):
    """dy_sinv: the slot the reduce pass already filled (else this call computes the scale itself)."""
    ready = dy_sinv is not None
    dy_sp.sinv = dy_sinv if ready else torch.empty(1, dtype=torch.float32, device=y.device)
    check(lib().mvg_bn_relu_maxpool_bwd_apply_split(_p(g_pooled), _p(argmax), _p(y), _p(mean), _p(invstd), _p(gamma), _p(scale), _p(shift),
                                                    _p(s1), _p(s2), groups, n_per_group, h, w, c, ho, wo, _p(dy_sp), _p(mx), _p(dy_sp.sinv),
                                                    int(ready), _s()), "bn_relu_maxpool_bwd_apply_split")


def stem_rowwindow_bf16(x_nchw: Tensor, out: Tensor):
    """[B, 3, H, W] fp32 NCHW -> out [B, H, W/4, 64] bf16: the stem's folded row windows (mvg_stem_fprop_bf16)."""
    B, Cc, H, W = x_nchw.shape
    assert Cc == 3 and W % 4 == 0 and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    assert out.dtype == torch.bfloat16 and out.shape == (B, H, W // 4, 64) and out.is_contiguous()
    check(lib().mvg_stem_rowwindow_bf16(_p(x_nchw), _p(out), B, H, W, _s()), "stem_rowwindow_bf16")


def stem_fprop_bf16(d: ConvDesc, xw: Tensor, w_fold: Tensor, y: Tensor, stats: Optional[Tensor]):
    check(lib().mvg_stem_fprop_bf16(C.byref(d), _p(xw), _p(w_fold), _p(y), _p(stats), _s()), "stem_fprop_bf16")


def stem_wgrad_bf16(d: ConvDesc, xw: Tensor, dy: Tensor, dw_fold: Tensor, accumulate: bool = False):
    """dw_fold [2 cout, 7, 16, 4] fp32 (+)= the stem's weight gradient in the folded-window tap layout."""
    splits = lib().mvg_stem_wgrad_splits_bf16(C.byref(d))
    if splits < 1:
        check(1, "stem_wgrad_splits_bf16")
    ws = torch.empty(splits * 2 * d.cout * 448, dtype=torch.float32, device=xw.device) if splits > 1 else None
    check(lib().mvg_stem_wgrad_bf16(C.byref(d), _p(xw), _p(dy), _p(dw_fold), _p(ws), splits, int(accumulate), _s()), "stem_wgrad_bf16")


def conv_dgrad_bn_partials_split(d: ConvDesc) -> int:
    n = lib().mvg_conv_dgrad_bn_partials_split(C.byref(d))
    if n < 0:
        check(1, "conv_dgrad_bn_partials_split")
    return n


def conv_dgrad_split_bnreduce(d: ConvDesc, dy_sp, wt_sp, dx, addend, bn_y, bn_bits, bn_mean, bn_invstd, relu_affine, s1, s2, dgamma,
                              dbeta, accumulate: bool, mx: Optional[Tensor] = None, bn_gamma: Optional[Tensor] = None,
                              dx_dy_sinv: Optional[Tensor] = None):
    """conv_dgrad_split + the BatchNorm-backward reduce pass of the unit whose output gradient dx is, in one launch.
    mx [groups, cin]: receives max |dx| per (group, channel) (bn_bwd_apply_split's bound)."""
    P = conv_dgrad_bn_partials_split(d)
    part = torch.empty(d.groups * P * 3 * d.cin, dtype=torch.float32, device=dx.device)
    rs, rh = relu_affine if relu_affine is not None else (None, None)
    check(lib().mvg_conv_dgrad_split_bnreduce(C.byref(d), _p(dy_sp), _sinv(dy_sp), _p(wt_sp), _sinv(wt_sp), _p(dx), _p(addend), _p(bn_y),
                                              _p(bn_bits), _p(bn_mean), _p(bn_invstd), _p(rs), _p(rh), _p(part), _p(s1), _p(s2),
                                              _p(dgamma), _p(dbeta), int(accumulate), _p(mx), _p(bn_gamma), _p(dx_dy_sinv), _s(True)),
          "conv_dgrad_split_bnreduce")


def conv_dgrad_bf16_bnreduce(d: ConvDesc, dy, wt, dx, addend, bn_y, bn_bits, bn_mean, bn_invstd, relu_affine, s1, s2, dgamma, dbeta,
                             accumulate: bool):
    """conv_dgrad (bf16 storage) + the BatchNorm-backward reduce pass of the unit whose output gradient dx is, in one launch."""
    P = lib().mvg_conv_dgrad_bn_partials_bf16(C.byref(d))
    if P < 1:
        check(1, "conv_dgrad_bn_partials_bf16")
    part = torch.empty(d.groups * P * 2 * d.cin, dtype=torch.float32, device=dx.device)
    rs, rh = relu_affine if relu_affine is not None else (None, None)
    check(lib().mvg_conv_dgrad_bf16_bnreduce(C.byref(d), _p(dy), _p(wt), _p(dx), _p(addend), _p(bn_y), _p(bn_bits), _p(bn_mean),
                                             _p(bn_invstd), _p(rs), _p(rh), _p(part), _p(s1), _p(s2), _p(dgamma), _p(dbeta),
                                             int(accumulate), _s()), "conv_dgrad_bf16_bnreduce")


def conv_wgrad_split(d: ConvDesc, x_sp: Tensor, dy_sp: Tensor, dw: Tensor, accumulate: bool = False, defer: Optional[list] = None):
    """defer (a list): with more than one pixel split, only the slabs are written and (slabs, dw, splits, accumulate) is appended
    for ONE wgrad_reduce_batch launch over the list (the caller's: at the end of a residual block, on the same stream)."""
    splits = lib().mvg_conv_wgrad_splits_split(C.byref(d))
    if splits < 1:
        check(1, "conv_wgrad_splits_split")
    ws = torch.empty(splits * dw.numel(), dtype=torch.float32, device=dw.device) if splits > 1 else None
    assert getattr(x_sp, "sinv", None) is None, "conv_wgrad_split: the activation operand is stored unscaled"
    if defer is not None and splits > 1:
        check(lib().mvg_conv_wgrad_split_slabs(C.byref(d), _p(x_sp), _p(dy_sp), _sinv(dy_sp), _p(ws), splits, _s()), "conv_wgrad_split_slabs")
        defer.append((ws, dw, splits, bool(accumulate)))
        return
    check(lib().mvg_conv_wgrad_split(C.byref(d), _p(x_sp), _p(dy_sp), _sinv(dy_sp), _p(dw), _p(ws), splits, int(accumulate), _s()),
          "conv_wgrad_split")


def wgrad_reduce_batch(items: list):
    """items: (slabs, dw, splits, accumulate) records left by conv_wgrad_split(defer=...): their slab sums, 8 per launch."""
    for i in range(0, len(items), 8):
        chunk = items[i:i + 8]
        n = len(chunk)
        for ws, dw, splits, acc in chunk:
            assert dw.dtype == torch.float32 and ws.numel() == splits * dw.numel() and dw.numel() % 4 == 0
        check(lib().mvg_wgrad_reduce_batch((C.c_void_p * n)(*[c[0].data_ptr() for c in chunk]), (C.c_void_p * n)(*[c[1].data_ptr() for c in chunk]),
                                           (C.c_int64 * n)(*[c[1].numel() for c in chunk]), (C.c_int32 * n)(*[c[2] for c in chunk]),
                                           (C.c_int32 * n)(*[int(c[3]) for c in chunk]), n, _s()), "wgrad_reduce_batch")
    items.clear()


def bn_apply_split(y, scale, shift, residual, relu, out_sp, groups, rows_per_group, c, residual_affine=None, want_bits=False):
    """y fp32 -> normalised (+ residual) (ReLU) activation in sp.  residual: an sp tensor (identity) or the raw fp32
    downsample output with residual_affine = its (scale, shift).  want_bits: also return the ReLU mask bytes."""
    bits = torch.empty(groups * rows_per_group * c // 4, dtype=torch.uint8, device=y.device) if want_bits else None
    rs, rh = residual_affine if residual_affine is not None else (None, None)
    res_sp = is_sp(residual)
    check(lib().mvg_bn_apply_split(_p(y), _p(scale), _p(shift), _p(residual), int(res_sp), _p(rs), _p(rh), int(relu), _p(out_sp),
                                   _p(bits), groups, rows_per_group, c, _s()), "bn_apply_split")
    return bits


def bn_bwd_reduce_split(g, relu_bits, y, mean, invstd, groups, rows_per_group, c, s1, s2, dgamma, dbeta, accumulate, mx,
                        relu_affine=None, dz_out=None, gamma=None, dy_sinv=None):
    """The backward reduce pass of a unit whose dy goes out in sp: s1, s2 (+ dgamma, dbeta) and mx [groups, c] = max |masked
    gradient| per (group, channel)."""
    rs, rh = relu_affine if relu_affine is not None else (None, None)
    n = lib().mvg_bn_bwd_workspace_floats(groups, rows_per_group, c)
    ws = torch.empty(n, dtype=torch.float32, device=g.device)
    check(lib().mvg_bn_bwd_reduce_split(_p(g), _p(relu_bits), _p(y), _p(mean), _p(invstd), _p(rs), _p(rh), groups, rows_per_group, c,
                                        _p(s1), _p(s2), _p(dgamma), _p(dbeta), int(accumulate), _p(ws), _p(dz_out), _p(mx), _p(gamma),
                                        _p(dy_sinv), _s(True)), "bn_bwd_reduce_split")


def bn_bwd_apply_split(g, y, mean, invstd, gamma, s1, s2, groups, rows_per_group, c, dy_sp, relu_affine=None, mx=None, dy_sinv=None):
    """dy (sp) = BatchNorm backward of the masked gradient g, scaled by the 2^k that its bound - from ``mx`` [groups, c] (max
    |masked gradient| per (group, channel), left by the reduce pass) - allows; dy_sp.sinv receives 2^-k."""
    rs, rh = relu_affine if relu_affine is not None else (None, None)
    assert mx is not None and mx.dtype == torch.float32 and mx.numel() == groups * c
    ready = dy_sinv is not None             # the reduce pass / fused backward-data launch already left 2^-k there
    dy_sp.sinv = dy_sinv if ready else torch.empty(1, dtype=torch.float32, device=g.device)
    check(lib().mvg_bn_bwd_apply_split(_p(g), _p(y), _p(mean), _p(invstd), _p(gamma), _p(s1), _p(s2), _p(rs), _p(rh), groups,
                                       rows_per_group, c, _p(dy_sp), _p(mx), _p(dy_sp.sinv), int(ready), _s()), "bn_bwd_apply_split")


def bn_relu_maxpool_fwd_split(y, scale, shift, pooled_sp, argmax, groups, n_per_group, h, w, c, ho, wo):
    check(lib().mvg_bn_relu_maxpool_fwd_split(_p(y), _p(scale), _p(shift), _p(pooled_sp), _p(argmax), groups, n_per_group, h, w, c,
                                              ho, wo, _s()), "bn_relu_maxpool_fwd_split")


def avgpool_fwd_split(x_sp, y, n, hw, c):
    check(lib().mvg_avgpool_fwd_split(_p(x_sp), _p(y), n, hw, c, _s()), "avgpool_fwd_split")


# ---- Linear layers of the fusion block in the bf16 path: fp32 tensors, bf16 matrix product (weights = bf16 copies)
def linear_fprop_mixed(x, w_bf16, bias, relu, y, rows, fin, fout):
    check(lib().mvg_linear_fprop_mixed(_p(x), _p(w_bf16), _p(bias), int(relu), _p(y), rows, fin, fout, _s()), "linear_fprop_mixed")


def linear_dgrad_mixed(dy, wt_bf16, mask, addend, dx, rows, fin, fout):
    check(lib().mvg_linear_dgrad_mixed(_p(dy), _p(wt_bf16), _p(mask), _p(addend), _p(dx), rows, fin, fout, _s()), "linear_dgrad_mixed")


def linear_wgrad_mixed(x, dy, dw, db, rows, fin, fout, accumulate=False):
    d = ConvDesc.linear(rows, fin, fout)
    splits = lib().mvg_conv_wgrad_splits_bf16(C.byref(d))
    if splits < 1:
        check(1, "conv_wgrad_splits_bf16")
    ws = torch.empty(splits * (fout * fin + fout), dtype=torch.float32, device=x.device) if splits > 1 else None
    check(lib().mvg_linear_wgrad_mixed(_p(x), _p(dy), _p(dw), _p(db), rows, fin, fout, _p(ws), splits, int(accumulate), _s()),
          "linear_wgrad_mixed")


def _linear_ws(x: Tensor, rows: int, fin: int, fout: int):
    n = lib().mvg_linear_workspace_floats(rows, fin, fout)
    return torch.empty(n, dtype=torch.float32, device=x.device), n


def linear_fprop(x: Tensor, w: Tensor, bias: Optional[Tensor], relu: bool, y: Tensor, rows: int, fin: int, fout: int):
    ws, n = _linear_ws(x, rows, fin, fout)
    check(lib().mvg_linear_fprop(_p(x), _p(w), _p(bias), int(relu), _p(y), rows, fin, fout, _p(ws), n, _s(True)),
          "linear_fprop")


def linear_dgrad(dy: Tensor, w: Tensor, mask: Optional[Tensor], addend: Optional[Tensor], dx: Tensor, rows: int,
                 fin: int, fout: int):
    ws, n = _linear_ws(dy, rows, fin, fout)
    check(lib().mvg_linear_dgrad(_p(dy), _p(w), _p(mask), _p(addend), _p(dx), rows, fin, fout, _p(ws), n, _s(True)),
          "linear_dgrad")


# ---------------------------------------------------------------- batch norm
def bn_finalize(stats, groups, partials, rows_per_partial, rows_per_group, c, gamma, beta, running_mean, running_var,
                momentum, eps, mean, invstd, scale, shift):
    check(lib().mvg_bn_finalize(_p(stats), groups, partials, rows_per_partial, rows_per_group, c, _p(gamma), _p(beta),
                                _p(running_mean), _p(running_var), momentum, eps, _p(mean), _p(invstd), _p(scale),
                                _p(shift), _s(True)), "bn_finalize")


def bn_eval_affine(groups, c, gamma, beta, running_mean, running_var, eps, scale, shift):
    check(lib().mvg_bn_eval_affine(groups, c, _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(scale),
                                   _p(shift), _s()), "bn_eval_affine")


def bn_apply(y, scale, shift, residual, relu, out, groups, rows_per_group, c, residual_affine=None):
    """residual_affine = (scale, shift) of the downsample branch's BatchNorm when `residual` is its RAW conv output."""
    rs, rh = residual_affine if residual_affine is not None else (None, None)
    check(_fn("mvg_bn_apply", y)(_p(y), _p(scale), _p(shift), _p(residual), _p(rs), _p(rh), int(relu), _p(out), groups,
                                 rows_per_group, c, _s()), "bn_apply")


def bn_apply_bits(y, scale, shift, residual, out, groups, rows_per_group, c, residual_affine=None):
    """bn_apply (+ residual, ReLU) that also returns the ReLU mask as one byte per 16-byte access of `out`."""
    per = 8 if y.dtype == torch.bfloat16 else 4
    bits = torch.empty(groups * rows_per_group * c // per, dtype=torch.uint8, device=y.device)
    rs, rh = residual_affine if residual_affine is not None else (None, None)
    check(_fn("mvg_bn_apply_bits", y)(_p(y), _p(scale), _p(shift), _p(residual), _p(rs), _p(rh), _p(out), _p(bits), groups,
                                      rows_per_group, c, _s()), "bn_apply_bits")
    return bits


def bn_bwd_reduce_bits(g, bits, y, mean, invstd, groups, rows_per_group, c, s1, s2, dgamma, dbeta, accumulate, dz_out=None):
    n = lib().mvg_bn_bwd_workspace_floats(groups, rows_per_group, c)
    ws = torch.empty(n, dtype=torch.float32, device=g.device)
    check(_fn("mvg_bn_bwd_reduce_bits", g)(_p(g), _p(bits), _p(y), _p(mean), _p(invstd), groups, rows_per_group, c, _p(s1), _p(s2),
                                           _p(dgamma), _p(dbeta), int(accumulate), _p(ws), _p(dz_out), _s()), "bn_bwd_reduce_bits")


def bn_bwd_reduce(g, act, y, mean, invstd, groups, rows_per_group, c, s1, s2, dgamma, dbeta, accumulate, relu_affine=None,
                  dz_out=None):
    """relu_affine = (scale, shift) of the forward bn_apply: ReLU mask rebuilt from y (units without residual).
    dz_out (may be g itself): the masked gradient is written out for the apply pass and the residual branch."""
    n = lib().mvg_bn_bwd_workspace_floats(groups, rows_per_group, c)
    ws = torch.empty(n, dtype=torch.float32, device=g.device)
    rs, rh = relu_affine if relu_affine is not None else (None, None)
    check(_fn("mvg_bn_bwd_reduce", g)(_p(g), _p(act), _p(y), _p(mean), _p(invstd), _p(rs), _p(rh), groups, rows_per_group, c,
                                  _p(s1), _p(s2), _p(dgamma), _p(dbeta), int(accumulate), _p(ws), _p(dz_out), _s()), "bn_bwd_reduce")


def bn_bwd_apply(g, act, y, mean, invstd, gamma, s1, s2, groups, rows_per_group, c, dy, dz_out=None, relu_affine=None):
    rs, rh = relu_affine if relu_affine is not None else (None, None)
    check(_fn("mvg_bn_bwd_apply", g)(_p(g), _p(act), _p(y), _p(mean), _p(invstd), _p(gamma), _p(s1), _p(s2), _p(rs), _p(rh),
                                 groups, rows_per_group, c, _p(dy), _p(dz_out), _s()), "bn_bwd_apply")


# ---------------------------------------------------------------- pooling / layout
def maxpool_fwd(x, y, argmax, n, h, w, c, ho, wo):
    check(lib().mvg_maxpool3x3s2_fwd(_p(x), _p(y), _p(argmax), n, h, w, c, ho, wo, _s()), "maxpool_fwd")


def maxpool_bwd(dy, argmax, dx, n, h, w, c, ho, wo):
    check(lib().mvg_maxpool3x3s2_bwd(_p(dy), _p(argmax), _p(dx), n, h, w, c, ho, wo, _s()), "maxpool_bwd")


def bn_relu_maxpool_fwd(y, scale, shift, pooled, argmax, groups, n_per_group, h, w, c, ho, wo):
    check(_fn("mvg_bn_relu_maxpool_fwd", y)(_p(y), _p(scale), _p(shift), _p(pooled), _p(argmax), groups, n_per_group, h, w, c, ho,
                                        wo, _s()), "bn_relu_maxpool_fwd")


def bn_relu_maxpool_bwd_reduce(g_pooled, argmax, y, mean, invstd, scale, shift, groups, n_per_group, h, w, c, ho, wo, s1, s2,
                               dgamma, dbeta, accumulate):
    n = lib().mvg_bn_bwd_workspace_floats(groups, n_per_group * h * w, c)
    ws = torch.empty(n, dtype=torch.float32, device=y.device)
    check(_fn("mvg_bn_relu_maxpool_bwd_reduce", y)(_p(g_pooled), _p(argmax), _p(y), _p(mean), _p(invstd), _p(scale), _p(shift),
                                               groups, n_per_group, h, w, c, ho, wo, _p(s1), _p(s2), _p(dgamma), _p(dbeta),
                                               int(accumulate), _p(ws), _s()), "bn_relu_maxpool_bwd_reduce")


def bn_relu_maxpool_bwd_apply(g_pooled, argmax, y, mean, invstd, gamma, scale, shift, s1, s2, groups, n_per_group, h, w, c,
                              ho, wo, dy):
    check(_fn("mvg_bn_relu_maxpool_bwd_apply", y)(_p(g_pooled), _p(argmax), _p(y), _p(mean), _p(invstd), _p(gamma), _p(scale),
                                              _p(shift), _p(s1), _p(s2), groups, n_per_group, h, w, c, ho, wo, _p(dy), _s()),
          "bn_relu_maxpool_bwd_apply")


def avgpool_fwd(x, y, n, hw, c):
    check(_fn("mvg_avgpool_fwd", x)(_p(x), _p(y), n, hw, c, _s()), "avgpool_fwd")


def avgpool_bwd(dy, dx, n, hw, c):
    check(_fn("mvg_avgpool_bwd", dx)(_p(dy), _p(dx), n, hw, c, _s()), "avgpool_bwd")


def nchw_to_nhwc4(src, dst, n, c, h, w):
    check(lib().mvg_nchw_to_nhwc4(_p(src), _p(dst), n, c, h, w, _s()), "nchw_to_nhwc4")


def nchw_to_nhwc8_bf16(src, dst, n, c, h, w):
    check(lib().mvg_nchw_to_nhwc8_bf16(_p(src), _p(dst), n, c, h, w, _s()), "nchw_to_nhwc8_bf16")


def nhwc4_to_nchw(src, dst, n, c, h, w):
    check(lib().mvg_nhwc4_to_nchw(_p(src), _p(dst), n, c, h, w, _s()), "nhwc4_to_nchw")


def low_priority_stream(device) -> "torch.cuda.Stream":
    """A lowest-priority HIP stream wrapped for torch (work queued on it yields the CUs to the caller's stream)."""
    with torch.cuda.device(device):
        ptr = lib().mvg_stream_create_low_priority()
    if not ptr:
        check(1, "stream_create_low_priority")
    return torch.cuda.ExternalStream(ptr, device=device)


def set_reserved_cus(n: int):
    check(lib().mvg_set_reserved_cus(int(n)), "set_reserved_cus")


def multi_erase_nchw(img, masks, grid, gmax, n, c, h, w):
    check(lib().mvg_multi_erase_nchw(_p(img), _p(masks), _p(grid), gmax, n, c, h, w, _s()), "multi_erase_nchw")


def preprocess_u8hwc_resize(src, dst, n, h, w, oh, ow, mean, std, swap_rb):
    check(lib().mvg_preprocess_u8hwc_resize(_p(src), _p(dst), n, h, w, oh, ow, mean[0], mean[1], mean[2], std[0], std[1],
                                            std[2], int(swap_rb), _s()), "preprocess_u8hwc_resize")


def preprocess_u8hwc(src, dst, n, h, w, mean, std, swap_rb):
    check(lib().mvg_preprocess_u8hwc(_p(src), _p(dst), n, h, w, mean[0], mean[1], mean[2], std[0], std[1], std[2],
                                     int(swap_rb), _s()), "preprocess_u8hwc")


# ---------------------------------------------------------------- geometry / fusion operands
def rotation_matrix_2d(pitch_yaw: Tensor, rot: Tensor, inverse: bool = False):
    check(lib().mvg_rotation_matrix_2d(_p(pitch_yaw), _p(rot), pitch_yaw.shape[0], int(inverse), _s()), "rotation_matrix_2d")


def relative_rotation(rot, vi, vj, rel, batch, views, dirs):
    check(lib().mvg_relative_rotation(_p(rot), _p(vi), _p(vj), _p(rel), batch, views, dirs, _s()), "relative_rotation")


def rotcat_fwd(img_feat, feat, rel, view_of, src_of, x, batch, dirs, cf, nvec):
    check(lib().mvg_rotcat_fwd(_p(img_feat), _p(feat), _p(rel), _p(view_of), _p(src_of), _p(x), batch, dirs, cf, nvec,
                               _s()), "rotcat_fwd")


def rotcat_bwd(dx, rel, src_of, dfeat, batch, dirs, cf, nvec):
    check(lib().mvg_rotcat_bwd(_p(dx), _p(rel), _p(src_of), _p(dfeat), batch, dirs, cf, nvec, _s()), "rotcat_bwd")


def rotcat_ext_fwd(img_feat, feat, rel_apply, rel_append, view_of, src_of, x, ld, batch, dirs, cf, nvec):
    check(lib().mvg_rotcat_ext_fwd(_p(img_feat), _p(feat), _p(rel_apply), _p(rel_append), _p(view_of), _p(src_of), _p(x), ld,
                                   batch, dirs, cf, nvec, _s()), "rotcat_ext_fwd")


def rotcat_ext_bwd(dx, ld, rel, src_of, dfeat, batch, dirs, cf, nvec):
    check(lib().mvg_rotcat_ext_bwd(_p(dx), ld, _p(rel), _p(src_of), _p(dfeat), batch, dirs, cf, nvec, _s()), "rotcat_ext_bwd")


def ibn_scales(a, feat, view_of, src_of, running_mean, training, momentum, eps, scales, batch, dirs, nvec):
    check(lib().mvg_ibn_scales(_p(a), _p(feat), _p(view_of), _p(src_of), _p(running_mean), int(training), momentum, eps,
                               _p(scales), batch, dirs, nvec, _s()), "ibn_scales")


def paircat_fwd(a, feat, rel, scales, view_of, src_of, x, batch, dirs, nvec):
    check(lib().mvg_paircat_fwd(_p(a), _p(feat), _p(rel), _p(scales), _p(view_of), _p(src_of), _p(x), batch, dirs, nvec, _s()),
          "paircat_fwd")


def paircat_bwd(dx, rel, scales, src_of, da_dir, dfeat, batch, dirs, nvec):
    check(lib().mvg_paircat_bwd(_p(dx), _p(rel), _p(scales), _p(src_of), _p(da_dir), _p(dfeat), batch, dirs, nvec, _s()),
          "paircat_bwd")


def segment_sum(x, row_stride, width, seg_of, out, batch, dirs, segments, accumulate):
    check(lib().mvg_segment_sum(_p(x), row_stride, width, _p(seg_of), _p(out), batch, dirs, segments, int(accumulate),
                                _s()), "segment_sum")


def axpby(x, y, a=1.0, b=1.0):
    check(lib().mvg_axpby(_p(x), _p(y), a, b, x.numel(), _s()), "axpby")


def scale_by(x, scale, out):
    check(lib().mvg_scale_by(_p(x), _p(scale), _p(out), x.numel(), _s()), "scale_by")


def linear_skinny_fwd(x, w, bias, y, rows, k, nout):
    check(lib().mvg_linear_skinny_fwd(_p(x), _p(w), _p(bias), _p(y), rows, k, nout, _s()), "linear_skinny_fwd")


def linear_skinny_bwd(dy, x, w, mask, dx, dw, db, rows, k, nout, accumulate=False, dx_absmax=None):
    check(lib().mvg_linear_skinny_bwd(_p(dy), _p(x), _p(w), _p(mask), _p(dx), _p(dw), _p(db), rows, k, nout,
                                      int(accumulate), _p(dx_absmax), _s()), "linear_skinny_bwd")


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step):
    check(lib().mvg_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, beta1, beta2, eps,
                              weight_decay, step, _s()), "adam_step")
    # the kernel wrote the parameters (and the moments) through raw pointers: tell autograd's version counters, which
    # the views of the arena share (saved-tensor checks, and the inference path's cache of split weights, rely on them)
    for t in (param, exp_avg, exp_avg_sq):
        torch.autograd.graph.increment_version(t)


def gaze_angular_loss(pred, gt, n, row_weight, loss, accumulate=False, dpred=None, theta=None):
    check(lib().mvg_gaze_angular_loss(_p(pred), _p(gt), n, row_weight, _p(loss), int(accumulate), _p(dpred), _p(theta),
                                      _s()), "gaze_angular_loss")


def gaze_lp_loss(pred, label, n, p, loss, dpred=None):
    check(lib().mvg_gaze_lp_loss(_p(pred), _p(label), n, p, _p(loss), _p(dpred), _s()), "gaze_lp_loss")


# ---------------------------------------------------------------- the fusion block on the split kernels
# Device "slots" are 1-element fp32 views into one per-model statistics arena that heads.FusionHead clears once per
# step: abs-max slots hold a float's bits (atomicMax), sinv slots the 2^-k of a scaled sp tensor.
def absmax_multi(tensors, slots):
    """slots[i] <- max |tensors[i]| (bits), one launch for up to 8 small fp32 tensors."""
    n = len(tensors)
    assert 1 <= n <= 8 and n == len(slots)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    cnts = (C.c_int64 * n)(*[t.numel() for t in tensors])
    outs = (C.c_void_p * n)(*[t.data_ptr() for t in slots])
    for t in tensors:
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    check(lib().mvg_absmax_multi(ptrs, cnts, outs, n, _s()), "absmax_multi")


def fuse_build_split(img_feat, feat, rel, row_img, row_src_f, row_src_h, xf_sp, xh_sp, am_img, am_feat, rows, cf, nvec):
    """xf_sp / xh_sp (sp tensors or None) <- the next fuser input / this head input built from `feat`; their .sinv slots are written."""
    check(lib().mvg_fuse_build_split(_p(img_feat), _p(feat), _p(rel), _p(row_img), _p(row_src_f), _p(row_src_h), _p(xf_sp), _p(xh_sp),
                                     _p(am_img), _p(am_feat), _sinv(xf_sp), _sinv(xh_sp), rows, cf, nvec, _s()), "fuse_build_split")


def fuse_unbuild(dxh, dxn, rel, seg, vi, dfeat, da, da_accumulate, segments, views, dirs, batch, cf, nvec, absmax=None):
    check(lib().mvg_fuse_unbuild(_p(dxh), _p(dxn), _p(rel), _p(seg), _p(vi), _p(dfeat), _p(da), int(da_accumulate), segments, views, dirs,
                                 batch, cf, nvec, _p(absmax), _s()), "fuse_unbuild")


def linear_fprop_split(x_sp, w_sp, bias, relu, out, rows, fin, fout, bias_absmax=None, out_absmax=None):
    """out = relu?(x W^T + bias) on the split kernels; out: an fp32 tensor, or an sp tensor whose .sinv slot receives its 2^-k."""
    check(lib().mvg_linear_fprop_split(rows, fin, fout, _p(x_sp), _sinv(x_sp), _p(w_sp), _sinv(w_sp), _p(bias), int(relu), _p(out),
                                       int(is_sp(out)), _sinv(out) if is_sp(out) else None, _p(bias_absmax), _p(out_absmax), _s()),
          "linear_fprop_split")


def linear_dgrad_split(dy_sp, wt_sp, dx, rows, fin, fout, addend=None, relu_mask_sp=None, out_absmax=None):
    check(lib().mvg_linear_dgrad_split(rows, fin, fout, _p(dy_sp), _sinv(dy_sp), _p(wt_sp), _sinv(wt_sp), _p(dx), _p(addend),
                                       _p(relu_mask_sp), _p(out_absmax), _s()), "linear_dgrad_split")


def linear_wgrad_split(x_sp, dy_sp, dw, rows, fin, fout, accumulate=False):
    d = ConvDesc.make(1, rows, 1, 1, fin, fout, 1, 1, 0)
    splits = lib().mvg_conv_wgrad_splits_split(C.byref(d))
    if splits < 1:
        check(1, "conv_wgrad_splits_split")
    ws = torch.empty(splits * dw.numel(), dtype=torch.float32, device=dw.device) if splits > 1 else None
    check(lib().mvg_linear_wgrad_split(rows, fin, fout, _p(x_sp), _sinv(x_sp), _p(dy_sp), _sinv(dy_sp), _p(dw), _p(ws), splits,
                                       int(accumulate), _s()), "linear_wgrad_split")


def split_colsum(g, rows, cols, absmax, out_sp, db=None, accumulate=False):
    """out_sp <- g in sp, scaled from the abs-max slot its producer filled (out_sp.sinv slot written); db (+)= column sums of g."""
    assert g.dtype == torch.float32 and g.is_contiguous() and g.numel() == rows * cols and cols % 32 == 0
    check(lib().mvg_split_colsum(_p(g), rows, cols, _p(absmax), _p(out_sp), _sinv(out_sp), _p(db), int(accumulate), _s()), "split_colsum")


def gaze_angular_loss_multi(pred, gt, iters, dirs, batch, weights, loss, dpred=None):
    w = (C.c_float * (iters * dirs))(*[float(x) for x in weights])
    check(lib().mvg_gaze_angular_loss_multi(_p(pred), _p(gt), iters, dirs, batch, w, _p(loss), _p(dpred), _s()), "gaze_angular_loss_multi")


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, lr_dev, state3, beta1, beta2, eps, weight_decay):
    check(lib().mvg_adam_step_dev(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(lr_dev), _p(state3), beta1, beta2, eps,
                                  weight_decay, _s()), "adam_step_dev")
    for t in (param, exp_avg, exp_avg_sq):
        torch.autograd.graph.increment_version(t)


# ---------------------------------------------------------------- profiling
def prof_enable(on: bool):
    lib().mvg_prof_enable(int(on))


def prof_reset():
    lib().mvg_prof_reset()


def prof_collect():
    arr = (ProfEntry * K_FAMILIES)()
    check(lib().mvg_prof_collect(arr), "prof_collect")
    out = {}
    for i in range(K_FAMILIES):
        e = arr[i]
        if e.launches:
            out[lib().mvg_prof_family_name(i).decode()] = {
                "launches": int(e.launches), "ms": e.ms, "flops": e.flops, "bytes": e.bytes}
    return out
