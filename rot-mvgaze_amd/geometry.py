"""Geometry helpers with the reference's names and argument meaning, computed by HIP kernels.

Mirrors /root/reference/utils/math.py: ``rotation_matrix_2d`` (:188-219), ``pitchyaw_to_vector``
(:24-60), ``vector_to_pitchyaw`` (:62-94) and ``angular_error`` (:97-136).  The functions on the path take
device tensors only - there is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def _dev_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise AssertionError(f"make sure the {what} here is torch.tensor")          # math.py:193
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the MI355X path needs a device tensor (no CPU fallback)")
    return t.detach().to(torch.float32).contiguous()


def rotation_matrix_2d(pitch_yaw: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """[N,2] (or [2]) head pose (pitch, yaw) -> [N,3,3] R = Ry(yaw) @ Rx(-pitch); inverse -> R^T."""
    hp = _dev_f32(pitch_yaw, "pitchyaw")
    if hp.dim() == 1:
        hp = hp.unsqueeze(0)
    rot = torch.empty(hp.shape[0], 3, 3, dtype=torch.float32, device=hp.device)
    ops.rotation_matrix_2d(hp, rot, inverse)
    return rot


def pitchyaw_to_vector(pitchyaws: torch.Tensor) -> torch.Tensor:
    """[N,2] -> unit gaze vectors [N,3] = R(pitchyaw)[:, :, 2]... computed as the third column of
    Ry(yaw) @ Rx(-pitch'), which equals (cos p sin y, sin p, cos p cos y)."""
    py = _dev_f32(pitchyaws, "pitchyaws")
    rot = torch.empty(py.shape[0], 3, 3, dtype=torch.float32, device=py.device)
    ops.rotation_matrix_2d(py, rot, False)
    return rot[:, :, 2].contiguous()


def vector_to_pitchyaw(vectors):
    """math.py:62-94: gaze vectors [N,3] (any length) -> (pitch, yaw) = (asin(y/|v|), atan2(x, z)), of the
    input's type.  The reference imports it next to the functions above (trainer.py:26,
    losses/gaze_loss.py:6) but never calls it on the training/eval path, so it is plain tensor/array
    arithmetic here (device tensors stay on the device), not a kernel."""
    if isinstance(vectors, np.ndarray):
        v = vectors / np.linalg.norm(vectors, axis=1).reshape(-1, 1)
        out = np.empty((v.shape[0], 2))
        out[:, 0] = np.arcsin(v[:, 1])
        out[:, 1] = np.arctan2(v[:, 0], v[:, 2])
        return out
    if isinstance(vectors, torch.Tensor):
        v = vectors / torch.norm(vectors, dim=1).reshape(-1, 1)
        return torch.stack([torch.asin(v[:, 1]), torch.atan2(v[:, 0], v[:, 2])], dim=1)
    raise ValueError("Unsupported input type. Only numpy arrays and torch tensors are supported.")


def angular_error(a, b):
    """Angular error in degrees.  numpy inputs follow the reference's evaluation metric
    (math.py:105-120, float64 on the host - the reference itself leaves the device here,
    trainer.py:128); device tensors [N,2] use the loss kernel's per-row angle."""
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        def vec(x):
            if x.shape[1] != 2:
                return x
            s, c = np.sin(x), np.cos(x)
            return np.stack([c[:, 0] * s[:, 1], s[:, 0], c[:, 0] * c[:, 1]], axis=1)
        a, b = vec(a), vec(b)
        ab = np.sum(a * b, axis=1)
        na = np.clip(np.linalg.norm(a, axis=1), 1e-7, None)
        nb = np.clip(np.linalg.norm(b, axis=1), 1e-7, None)
        return np.arccos(ab / (na * nb)) * 180.0 / np.pi
    if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
        pa, pb = _dev_f32(a, "a"), _dev_f32(b, "b")
        assert pa.shape[1] == 2 and pb.shape[1] == 2, "device path takes (pitch, yaw) pairs"
        theta = torch.empty(pa.shape[0], dtype=torch.float32, device=pa.device)
        loss = torch.empty(1, dtype=torch.float32, device=pa.device)
        ops.gaze_angular_loss(pa, pb, pa.shape[0], 1.0, loss, False, None, theta)
        return theta
    raise ValueError("Input type mismatch. Both inputs should be either numpy arrays or torch tensors.")
