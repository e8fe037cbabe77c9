"""A whole training step - zero_grad, forward, loss, backward, fused Adam - captured ONCE in a hipGraph and replayed.

Why: a ResNet-50 step is ~600 kernel launches that the Python host queues one by one through ctypes (~35 us each):
21.5 ms of host time against a 25 ms GPU step at BASELINE config C4's per-GPU share (profiles/r03_bench_c4.json), i.e.
the data-parallel configurations sit at the edge of being host-bound.  Nothing on the step path synchronises or
depends on host data (tests/test_model_gpu.py::test_training_step_does_not_synchronise_the_host), the optimizer's step
counter and learning rate live on the device (optim.Adam(capturable=True)), and the side stream of the weight-gradient
kernels forks from / joins the capturing stream through events - so the step is capturable as it stands; a replay costs
the host one hipGraphLaunch.  (The reference's caller: /root/reference/trainer.py:119-147 - model(data), metrics(data),
zero_grad, backward, optimizer.step per iteration.)

Measured on MI355X / ROCm 7.0 (scripts/graph_probe.py, profiles/README.md round 4): a replay costs the host 0.08-0.15 ms
instead of 5-20 ms, and takes exactly the GPU time of the eager step WITHOUT the side stream (C4's share 27.98 vs 27.80 ms,
C2 9.73 vs 9.74): the eager step is GPU-bound, so the graph buys host time, not throughput.  A capture that includes the
weight-gradient side stream (fork / join events -> a branching graph) replays 1.45-1.7x SLOWER than the eager step (38.0 vs
26.2 ms, 15.3 vs 9.0): the capture is therefore taken single-stream (overlap_wgrad off), which gives up the ~6 % the
overlap is worth on one GPU.  Use it where the host is the bottleneck (many ranks per host, small per-GPU batches).

PyTorch here is plumbing only: ``torch.cuda.CUDAGraph`` is hipStreamBeginCapture / hipGraphInstantiate / hipGraphLaunch
plus a private pool of the caching allocator, which is what keeps every buffer the captured launches point at alive
and un-recycled between replays.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

Tensor = torch.Tensor


class GraphedStep:
    """``step_fn()`` -> loss: one training step on STATIC input tensors (the caller refills them in place between
    replays: ``img.copy_(next_batch)``).  ``run()`` replays the captured step and returns the (static) loss tensor."""

    def __init__(self, model, step_fn: Callable[[], Tensor], optimizer=None, warmup: int = 3):
        if optimizer is not None and not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedStep needs rot_mvgaze_amd.optim.Adam(capturable=True): the step counter and the "
                             "learning rate must live on the device")
        self.model, self.optimizer = model, optimizer
        self._step_fn = step_fn
        dev = next(model.parameters()).device
        model.ensure_layout()
        model._backbone.overlap_wgrad = False          # single-stream capture (see the module docstring)
        # eager warm-up on a side stream (allocator pools, weight-copy tables, scratch registration, the optimizer's
        # device state): capture must find nothing left to set up
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                step_fn()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        if optimizer is not None:
            optimizer.sync_lr()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step_fn()
        self.replays = 0

    def run(self) -> Tensor:
        if self.optimizer is not None:
            self.optimizer.sync_lr()                 # a scheduler may have moved the learning rate: one 4-byte fill
        self.graph.replay()
        self.replays += 1
        # the captured kernels rewrote the parameters through raw pointers: drop the inference path's cached weight copies
        self.model.invalidate_weight_cache()
        return self.loss
