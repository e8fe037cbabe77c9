"""RandomMultiErasing on a device batch (the last step of the reference's
training transform, /root/reference/main.py:48, /root/reference/utils/augment.py:10-47).

The random draws stay on the host and replay the reference's calls in its order - per image:
``random.random()`` (apply?), ``np.random.uniform(*dot_size)``, ``np.random.uniform(*proportion)``,
``torch.rand(g, g)`` with g = int(1 / dot_size) - so a run seeded like the reference erases the
same cells.  The multiply runs in one HIP launch over the whole batch (mvg_multi_erase_nchw).
"""
from __future__ import annotations

import random
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import ops

Tensor = torch.Tensor


class RandomMultiErasing:
    def __init__(self, proportion: Sequence[float], p: float, dot_size: Sequence[float]):
        self.proportion = proportion
        self.p = p
        self.dot_size = dot_size

    def draw(self, n: int) -> List[Tuple[int, Tensor]]:
        """The reference's per-image draws (augment.py:38-45, :16-20) for n images: [(g, keep-mask g x g)],
        g == 0 for an image that is left alone."""
        out = []
        for _ in range(n):
            if random.random() > self.p:
                out.append((0, torch.zeros(0, 0)))
                continue
            dot_size = np.random.uniform(*self.dot_size)
            proportion = np.random.uniform(*self.proportion)
            g = int(1 / dot_size)
            mask = (torch.rand(g, g) > proportion).to(torch.float32)
            out.append((g, mask))
        return out

    def apply(self, img: Tensor, draws: List[Tuple[int, Tensor]]) -> Tensor:
        """img [B,C,H,W] fp32 on the GPU, erased in place (the reference's ``img *= mask``)."""
        if not img.is_cuda:
            raise RuntimeError("RandomMultiErasing (MI355X build) works on device batches: no CPU fallback")
        assert img.dim() == 4 and img.dtype == torch.float32 and img.is_contiguous() and len(draws) == img.shape[0]
        B, C, H, W = img.shape
        gmax = max(1, max(g for g, _ in draws))
        masks = torch.zeros(B, gmax * gmax, dtype=torch.float32)
        grid = torch.zeros(B, dtype=torch.int32)
        for i, (g, m) in enumerate(draws):
            grid[i] = g
            if g:
                masks[i, : g * g] = m.reshape(-1)
        ops.multi_erase_nchw(img, masks.to(img.device), grid.to(img.device), gmax, B, C, H, W)
        return img

    def __call__(self, img: Tensor) -> Tensor:
        """[B,C,H,W] (or one image [C,H,W]) device tensor -> erased in place, one draw per image."""
        one = img.dim() == 3
        x = img.unsqueeze(0) if one else img
        self.apply(x, self.draw(x.shape[0]))
        return img
