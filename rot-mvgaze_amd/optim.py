"""Adam with the reference trainer's configuration surface, one fused HIP launch per step.

Mirrors ``optim.Adam(self.model.parameters(), lr=0, weight_decay=1e-6)`` of
/root/reference/trainer.py:54 (stepped at :141-143, lr driven by CyclicLR :58-62): a
``torch.optim.Optimizer`` subclass, so ``CyclicLR(optimizer, ..., cycle_momentum=False)`` works on it
unchanged.  The update itself is ``mvg_adam_step`` over the model's parameter / gradient arenas
(identical offsets), i.e. one kernel for all ~90 M parameters instead of ~190 foreach launches.
Parameters without a gradient (the unused ``resnet.fc``) are skipped, as torch does.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise NotImplementedError("one parameter group (the reference uses one)")
        self._flat: Dict[int, dict] = {}
        self._pending = None          # moments restored by load_state_dict, adopted by the next step()
        # capturable (like torch.optim.Adam's flag): the step counter, the bias corrections and the learning rate live on
        # the DEVICE (mvg_adam_step_dev), so that a step captured in a hipGraph (rot_mvgaze_amd.graph.GraphedStep) replays
        # with nothing from the host; the learning rate is pushed to the device by sync_lr() - outside a capture - whenever
        # the scheduler changed it (the reference steps CyclicLR once per epoch, trainer.py:147).
        self.capturable = bool(capturable)

    def sync_lr(self):
        """capturable: copy param_groups[0]['lr'] to the device scalar the captured update reads (no-op when unchanged)."""
        lr = float(self.param_groups[0]["lr"])
        for st in self._flat.values():
            if "lr_dev" in st and st["lr_host"] != lr:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("Adam(capturable=True): the learning rate changed inside a graph capture; call "
                                       "optimizer.sync_lr() before capturing / replaying")
                st["lr_dev"].fill_(lr)
                st["lr_host"] = lr

    def _owners(self):
        owners = {}
        for p in self.param_groups[0]["params"]:
            ref = getattr(p, "_mvg_owner", None)
            model = ref() if ref is not None else None
            if model is None:
                raise RuntimeError("rot_mvgaze_amd.optim.Adam only updates parameters of the MI355X "
                                   "FeatRotationSymm / MultiViewGaze modules")
            owners[id(model)] = model
        return list(owners.values())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        for model in self._owners():
            model.ensure_layout()
            arena_g, entries = model.grad_arena()
            have = [p.grad is not None for (p, _, _) in entries]
            if not any(have):
                continue                              # nothing was back-propagated: torch skips too
            for (p, off, n) in entries:
                if p.grad is None or p.grad.data_ptr() != arena_g.data_ptr() + 4 * off:
                    raise RuntimeError("every trainable parameter must hold the gradient written by the "
                                       "module's backward (a view of its gradient arena)")
            st = self._flat.get(id(model))
            arena_p = model.param_arena()
            if st is None and self._pending:
                saved = self._pending.pop(0)
                if saved["exp_avg"].numel() != arena_p.numel():
                    raise RuntimeError("optimizer state does not match this model's parameter arena "
                                       f"({saved['exp_avg'].numel()} vs {arena_p.numel()} elements)")
                st = {"step": int(saved["step"]), "n": arena_p.numel(),
                      "exp_avg": saved["exp_avg"].to(arena_p.device, torch.float32).contiguous(),
                      "exp_avg_sq": saved["exp_avg_sq"].to(arena_p.device, torch.float32).contiguous()}
                self._flat[id(model)] = st
            if st is None or st["exp_avg"].data_ptr() == 0 or st["n"] != arena_p.numel() or \
                    st["exp_avg"].device != arena_p.device:
                st = {"step": 0, "n": arena_p.numel(), "exp_avg": torch.zeros_like(arena_p),
                      "exp_avg_sq": torch.zeros_like(arena_p)}
                self._flat[id(model)] = st
            st["step"] += 1
            if self.capturable:
                if "lr_dev" not in st:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("Adam(capturable=True): run one eager step before capturing (its device state is created then)")
                    st["lr_dev"] = torch.full((1,), float(g["lr"]), dtype=torch.float32, device=arena_p.device)
                    st["lr_host"] = float(g["lr"])
                    # {step, 1 - beta1^step, sqrt(1 - beta2^step)}: advanced on the device by every (captured or eager) step
                    st["state3"] = torch.tensor([float(st["step"] - 1), 0.0, 0.0], dtype=torch.float32).to(arena_p.device)
                self.sync_lr()
                ops.adam_step_dev(arena_p, arena_g, st["exp_avg"], st["exp_avg_sq"], st["lr_dev"], st["state3"], float(g["betas"][0]),
                                  float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
            else:
                ops.adam_step(arena_p, arena_g, st["exp_avg"], st["exp_avg_sq"], float(g["lr"]), float(g["betas"][0]),
                              float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), st["step"])
            # the parameters alias the arena through `.data` (their own version counters): mark them modified, as an
            # in-place torch op would (the inference path's cache of split weights keys on the version)
            for (p, _, _) in entries:
                torch.autograd.graph.increment_version(p)
        return loss

    # The moments live in flat buffers that mirror the model's parameter arena (one launch updates all of
    # them), not in ``self.state``; checkpoints carry them under one extra key so that a resumed run keeps
    # its bias correction and moments (the reference itself checkpoints the model only, trainer.py:91-96).
    def state_dict(self):
        sd = super().state_dict()
        flat = []
        for model in self._owners():
            st = self._flat.get(id(model))
            if st is not None:
                if "state3" in st:                  # capturable: the device counter is the truth (graph replays advance it)
                    st["step"] = int(round(float(st["state3"][0].item())))
                flat.append({"step": st["step"], "exp_avg": st["exp_avg"].detach().clone(),
                             "exp_avg_sq": st["exp_avg_sq"].detach().clone()})
        sd["mvg_arena_state"] = flat
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        flat = sd.pop("mvg_arena_state", None)
        super().load_state_dict(sd)
        self._flat.clear()
        self._pending = [dict(e) for e in flat] if flat else None
