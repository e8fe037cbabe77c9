"""Loss modules with the reference's names, constructor arguments and dict protocol, computed by
the fused HIP angular-loss kernel (forward value + analytic gradient in one launch).

Mirrors /root/reference/losses/gaze_loss.py:8-52 (``GazeLoss`` / ``gaze_angular_loss``) and
/root/reference/losses/stereo_loss.py:25-84 (``StereoL1Loss``, ``IterationLoss``).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .heads import directed_pairs

Tensor = torch.Tensor


class _AngularLossFn(torch.autograd.Function):
    """mean_b theta(pred_b, gt_b) in degrees; gradient wrt pred comes out of the same launch."""

    @staticmethod
    def forward(ctx, pred: Tensor, gt: Tensor):
        assert pred.shape[-1] == 2 and gt.shape[-1] == 2 and pred.is_cuda
        n = pred.shape[0]
        p = pred.detach().to(torch.float32).contiguous()
        g = gt.detach().to(torch.float32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        dpred = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        ops.gaze_angular_loss(p, g, n, 1.0 / n, loss, False, dpred, None)
        ctx.dpred = dpred
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.dpred is None:
            return None, None
        out = torch.empty_like(ctx.dpred)
        ops.scale_by(ctx.dpred, g.reshape(1).to(torch.float32).contiguous(), out)
        return out, None


def gaze_angular_loss(y_hat: Tensor, y: Tensor) -> Tensor:
    return _AngularLossFn.apply(y_hat, y)


class _LpLossFn(torch.autograd.Function):
    """mean |pred - label|^p over all elements (gaze_loss.py:56-64); the gradient comes out of the same launch."""

    @staticmethod
    def forward(ctx, pred: Tensor, label: Tensor, p: int):
        if not pred.is_cuda:
            raise RuntimeError("GazeLoss: the MI355X path needs device tensors (no CPU fallback)")
        a = pred.detach().to(torch.float32).contiguous()
        b = label.detach().to(torch.float32).expand_as(a).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        dpred = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        ops.gaze_lp_loss(a, b, a.numel(), p, loss, dpred)
        ctx.dpred = dpred
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.dpred is None:
            return None, None, None
        out = torch.empty_like(ctx.dpred)
        ops.scale_by(ctx.dpred, g.reshape(1).to(torch.float32).contiguous(), out)
        return out, None, None


def gaze_l1_loss(y: Tensor, y_hat: Tensor) -> Tensor:
    return _LpLossFn.apply(y, y_hat, 1)


def gaze_l2_loss(y: Tensor, y_hat: Tensor) -> Tensor:
    return _LpLossFn.apply(y, y_hat, 2)


class GazeLoss(nn.Module):
    def __init__(self, gaze_weight, loss_type: str, head_weight=1.0):
        super().__init__()
        self.gaze_weight = gaze_weight
        self.head_weight = head_weight
        assert loss_type in ['l1', 'l2', 'angular']
        self.loss_type = loss_type

    def forward(self, pred, label):
        if self.loss_type == 'angular':                     # the only type StereoL1Loss builds (stereo_loss.py:37-39)
            return gaze_angular_loss(pred, label)
        assert pred.shape[-1] == 2 and label.shape[-1] == 2, \
            f"the prediction should be in pitchyaw [batch, 2], got pred: {pred.shape}, and label should be in pitchyaw, got label: {label.shape}"
        return gaze_l1_loss(pred, label) if self.loss_type == 'l1' else gaze_l2_loss(pred, label)


class AbstractLoss(nn.Module):
    @property
    def name(self) -> str:
        return self._name if getattr(self, "_name", None) is not None else self.__class__.__name__


class StereoL1Loss(AbstractLoss):
    def __init__(self, rel_weight: float = 1, reference_decay: float = 1.0, distance_metric: str = "angular_error",
                 pred_gaze_key: str = "pred_gaze", name: Optional[str] = None):
        super().__init__()
        self._rel_weight = rel_weight
        self._distance_metric = GazeLoss(gaze_weight=1.0, loss_type='angular')
        self._reference_decay = reference_decay
        self._pred_gaze_key = pred_gaze_key
        if name is not None:
            self._name = name

    def forward(self, data: Dict[str, Any]):
        loss = self._distance_metric(data[f"{self._pred_gaze_key}_0"], data["gt_gaze"])
        loss_aux = self._distance_metric(data[f"{self._pred_gaze_key}_1"], data["gt_gaze_1"])
        return (loss + loss_aux * self._reference_decay) * self._rel_weight


class _FusedIterLossFn(torch.autograd.Function):
    """sum_it w_it * sum_d c_d * mean_b theta(pred[it,d,b], gt[d,b]) over the head's stacked
    prediction buffer [I,D,B,2]: one launch, gradient written once for all rows."""

    @staticmethod
    def forward(ctx, preds: Tensor, gt_dir: Tensor, iter_w, dir_w):
        I, D, B, _ = preds.shape
        p = preds.detach().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        dpred = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        # every iteration and direction in ONE launch (row weight = iter_w[it] * dir_w[d] / B)
        ops.gaze_angular_loss_multi(p, gt_dir.contiguous(), I, D, B, [iter_w[it] * dir_w[d] for it in range(I) for d in range(D)], loss, dpred)
        ctx.dpred = dpred
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.dpred is None:
            return None, None, None, None
        out = torch.empty_like(ctx.dpred)
        ops.scale_by(ctx.dpred, g.reshape(1).to(torch.float32).contiguous(), out)
        return out, None, None, None


class IterationLoss(AbstractLoss):
    def __init__(self, loss: AbstractLoss, iter_decay: float = 1.0, additional_decay: Optional[float] = None) -> None:
        super().__init__()
        self._name = "Iter" + loss.name
        self._loss = loss
        self._iter_decay = iter_decay
        self._addtional_decay = additional_decay

    def forward(self, data: Dict[str, Any]) -> Tensor:
        """total = sum_i w_i * loss(iter_i) with Horner weights w_i = iter_decay^(n-1-i) (and the
        optional separately weighted last iteration), stereo_loss.py:65-84.  Like the reference, every
        per-iteration dict is first completed with the shared (non ``iter_*``) entries of ``data``."""
        n = int(data["num_iter"])
        shared = {k: v for k, v in data.items() if not k.startswith("iter_")}
        for i in range(n):
            data[f"iter_{i}"].update(shared)
        fused = self._fused(data, n)
        if fused is not None:
            return fused
        n_horner = n if self._addtional_decay is None else n - 1
        total: Any = 0
        for i in range(n_horner):
            total = total * self._iter_decay + self._loss(data[f"iter_{i}"])
        if self._addtional_decay is not None:
            total = total + self._loss(data[f"iter_{n_horner}"]) * self._addtional_decay
        return total

    def _fused(self, data: Dict[str, Any], num_iter: int) -> Optional[Tensor]:
        """One autograd node over the head's stacked predictions when the dict came from the
        MI355X FeatRotationSymm and the wrapped loss is the standard StereoL1Loss."""
        preds = data.get("_mvg_preds")
        if preds is None or type(self._loss) is not StereoL1Loss or self._loss._pred_gaze_key != "pred_gaze":
            return None
        if preds.shape[0] != num_iter or preds.shape[1] != 2:
            return None
        l = self._loss
        if self._addtional_decay is None:
            iw = [self._iter_decay ** (num_iter - 1 - i) for i in range(num_iter)]
        else:
            iw = [self._iter_decay ** (num_iter - 2 - i) for i in range(num_iter - 1)] + [self._addtional_decay]
        gt = torch.stack([data["gt_gaze"], data["gt_gaze_1"]], 0).to(torch.float32).contiguous()
        return _FusedIterLossFn.apply(preds, gt, iw, [l._rel_weight, l._rel_weight * l._reference_decay])


class MultiViewIterationLoss(AbstractLoss):
    """V-view loss of SURVEY.md §8(a) A9: mean over unordered pairs of the two-view IterationLoss.
    Takes the dict of ``MultiViewGaze.forward_multiview`` plus gt [B,V,2]."""

    def __init__(self, rel_weight: float = 0.01, reference_decay: float = 1.0, iter_decay: float = 0.5):
        super().__init__()
        self._rel_weight, self._reference_decay, self._iter_decay = rel_weight, reference_decay, iter_decay

    def forward(self, out: Dict[str, Any], gt: Tensor) -> Tensor:
        preds = out["_mvg_preds"]
        I, D = preds.shape[0], preds.shape[1]
        V = out["views"]
        vi, _ = directed_pairs(V)
        npairs = D // 2
        # [D,B,2] (index plumbing).  Slices, not gt[...][vi]: indexing with a Python list builds the index
        # tensor on the host and its pageable H2D copy waits for the stream - a host/GPU sync per step
        # that stops the host from queueing the next step's launches under this step's backward
        g32 = gt.to(torch.float32)
        gt_dir = torch.stack([g32[:, v] for v in vi], 0)
        iw = [self._iter_decay ** (I - 1 - i) for i in range(I)]
        dw = [(self._rel_weight if d % 2 == 0 else self._rel_weight * self._reference_decay) / npairs
              for d in range(D)]
        return _FusedIterLossFn.apply(preds, gt_dir, iw, dw)
