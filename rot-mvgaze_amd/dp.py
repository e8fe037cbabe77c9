"""Data parallelism over samples: one process per GPU, RCCL all-reduce of the gradient arena over
xGMI, overlapped with the rest of backward on a side HIP stream.

Nothing in the reference corresponds to this (it is single-device, trainer.py:42,50;
SURVEY.md §8(e)); the contract is "N ranks with averaged gradients == N independent reference
steps with averaged gradients" (BatchNorm statistics stay rank-local: the reference has no SyncBN).

The model's gradient arena is already laid out in grad-ready order (heads/fusers first - final
before backbone backward even starts - then layer4 ... stem), so buckets are contiguous arena
slices: no flatten/unflatten copies, one ``all_reduce`` per bucket issued the moment its last
parameter is published, on a side stream that waits on an event recorded on the compute stream.

Replicas are made identical at construction: rank 0's parameter arena, the parameters outside it
(the unused ``resnet.fc``) and every buffer (BatchNorm / IntensityBatchNorm running statistics) are
broadcast to the other ranks (``sync()``), as torch's DistributedDataParallel does - each process
otherwise draws its own random initial weights (model.py:_init_tensor) and averaged gradients of
different replicas mean nothing.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, model, bucket_mb: float = 64.0, process_group=None, average: bool = True, force: bool = False,
                 broadcast: bool = True, reserved_cus: Optional[int] = None):
        self.model = model
        self.pg = process_group
        self.average = average
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = force            # run the collectives even with one rank (rehearsal of the RCCL path)
        self.reserved_cus = int(os.environ.get("MVG_RESERVED_CUS", "4")) if reserved_cus is None else int(reserved_cus)
        self._built_for = None
        self._side: Optional[torch.cuda.Stream] = None
        self._native_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        model._on_grads_ready = self._on_ready
        model._on_backward_done = self._on_done
        # Everything that changes how a backward runs (the stream the backward-weight kernels use, the
        # CUs the persistent conv grids plan for) is applied HERE, before any backward work exists:
        # flipping them from the first publish callback - in the middle of the first backward - left the
        # head's weight gradients on a stream nobody joined.
        self._configure()
        first = next(iter(model.parameters()), None) if hasattr(model, "parameters") else None
        if first is not None and hasattr(model, "ensure_layout") and first.is_cuda:
            model.ensure_layout()
            self._build()
        if broadcast and hasattr(model, "param_arena"):
            self.sync()

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    def _configure(self):
        if not self.active or not hasattr(self.model, "_wgrad_low_priority"):
            return
        # The conv kernels run as persistent stream-K grids sized to fill every CU; an RCCL kernel that
        # is resident when one starts would push part of that grid into a second round (2x the kernel
        # time).  Plan the grids for a few CUs less and keep RCCL to a few channels (the gradient
        # stream needs a few GB/s of bus bandwidth, a fraction of what 4 channels move; bench.py --help has the
        # one-rank rehearsal figures behind the defaults).
        from . import ops
        ops.set_reserved_cus(self.reserved_cus)
        # backward-weight kernels on a LOWEST-priority stream finish last, which would hold every
        # backbone bucket back until the end of backward; with ranks to feed they run on an ordinary
        # side stream (still off the critical path, +3 % instead of +4 %).  The model applies the
        # setting to every Backbone it (re)builds.
        if os.environ.get("MVG_DP_LOW_PRIORITY_WGRAD", "0") == "0":
            self.model._wgrad_low_priority = False
            bb = getattr(self.model, "_backbone", None)
            if bb is not None:
                bb.wgrad_low_priority = False

    def sync(self, src: int = 0):
        """Make this replica identical to rank ``src``: parameter arena, parameters outside the arena
        (``resnet.fc``: in the state_dict, never trained) and all buffers."""
        if self.world <= 1:
            return
        m = self.model
        if hasattr(m, "ensure_layout"):
            first = next(iter(m.parameters()))
            if first.is_cuda:
                m.ensure_layout()
        arena = m.param_arena() if getattr(m, "_layout_sig", True) is not None else None
        in_arena = getattr(m, "_grad_offsets", {})
        tensors = ([arena] if arena is not None else []) + \
                  [p.data for p in m.parameters() if arena is None or id(p) not in in_arena] + list(m.buffers())
        with torch.no_grad():
            for t in tensors:
                dist.broadcast(t, src, group=self.pg)
            if arena is not None:
                # the parameters alias the arena through `.data` and keep their own version counters: mark them
                # modified (the inference path caches split copies of the conv weights keyed on the version)
                for p_ in m.parameters():
                    torch.autograd.graph.increment_version(p_)

    # bucket = [start, end) element range of the arena + the id of its last parameter
    def _build(self):
        arena, entries = self.model.grad_arena()
        if self._built_for == arena.data_ptr():
            return
        self.arena = arena
        self.buckets: List[List[int]] = []
        self.last_param_bucket = {}
        start, cur = 0, 0
        for (p, off, n) in entries:
            cur = off + n
            if (cur - start) * 4 >= self.bucket_bytes:
                self.buckets.append([start, cur])
                self.last_param_bucket[id(p)] = len(self.buckets) - 1
                start = cur
        if cur > start:
            self.buckets.append([start, cur])
            self.last_param_bucket[id(entries[-1][0])] = len(self.buckets) - 1
        self._entries = entries
        # a bucket is complete when EVERY parameter inside it has been published this backward (not when some later
        # parameter has: a parameter the backward never reaches would otherwise ship last step's gradient)
        self._bucket_params = [[id(p) for (p, off, n) in entries if s <= off < e] for (s, e) in self.buckets]
        self._published = set()
        self._next = 0
        if arena.is_cuda and self._side is None:
            self._side = torch.cuda.Stream(device=arena.device)
        self._built_for = arena.data_ptr()

    def _launch(self, b: int):
        s, e = self.buckets[b]
        buf = self.arena[s:e]
        if not self.active:
            return
        if self.arena.is_cuda:
            # gradients of this bucket are complete once the compute stream AND the model's other gradient
            # streams (backward-weight kernels run on their own) reach this point
            for st in [torch.cuda.current_stream()] + list(getattr(self.model, "_grad_streams", [])):
                ev = torch.cuda.Event()
                ev.record(st)
                self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                if self.average and self._native_avg:
                    dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.pg)      # RCCL averages in the collective
                else:
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
                    if self.average:
                        buf.mul_(1.0 / self.world)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            if self.average:
                buf.mul_(1.0 / self.world)

    def _on_ready(self, params):
        self._build()
        for p in params:
            self._published.add(id(p))
        # buckets ship in arena order (= grad-ready order), each as soon as all of its parameters are final
        while self._next < len(self.buckets) and all(pid in self._published for pid in self._bucket_params[self._next]):
            self._launch(self._next)
            self._next += 1

    def _on_done(self):
        self._build()
        # Buckets that were not complete when backward ended: a backward that never reached part of the model (no
        # gradient flows into the image features, say) leaves those slices of the arena holding the PREVIOUS step's
        # gradients.  A bucket none of whose parameters was published this backward is skipped (every rank skips the
        # same ones: the model is the same everywhere); in a partly published bucket the stale slices are zeroed first,
        # so that nothing stale is averaged into gradients that are still attached from an earlier step.  (Those zeroed
        # slices are the one place where an N-rank run hands the optimizer something else than a 1-rank run of the same
        # code would - there the stale .grad of an unreached parameter stays attached; callers that rely on it zero_grad
        # every step, as the reference trainer does: trainer.py:141.)
        while self._next < len(self.buckets):
            s, e = self.buckets[self._next]
            inside = [(p, off, n) for (p, off, n) in self._entries if s <= off < e]
            if any(id(p) in self._published for (p, _, _) in inside):
                for (p, off, n) in inside:
                    if id(p) not in self._published:
                        self.arena[off:off + n].zero_()
                self._launch(self._next)
            self._next += 1
        if self.arena.is_cuda and self.active:
            torch.cuda.current_stream().wait_stream(self._side)
        self._published = set()
        self._next = 0
