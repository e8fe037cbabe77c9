"""world_size-2 rehearsal of the data-parallel gradient reduction on CPU (gloo): the same
GradAllReducer bucketing / grad-ready ordering that runs over RCCL on the GPUs, driven over a CPU
gradient arena."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rot_mvgaze_amd  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel:
    """Stands in for MultiViewGaze on CPU: a gradient arena with parameters in grad-ready order."""

    def __init__(self, sizes, rank):
        self.params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
        total = sum(sizes)
        g = torch.Generator().manual_seed(100 + rank)
        self.arena = torch.randn(total, generator=g)
        self.entries, off = [], 0
        for p, n in zip(self.params, sizes):
            self.entries.append((p, off, n))
            off += n
        self._on_grads_ready = None
        self._on_backward_done = None

    def grad_arena(self):
        return self.arena, self.entries


class _FakeReplica(_FakeModel):
    """... with a parameter arena, a parameter outside it and a buffer, each seeded per rank: what
    GradAllReducer.sync() must make identical to rank 0's."""

    def __init__(self, sizes, rank):
        super().__init__(sizes, rank)
        g = torch.Generator().manual_seed(500 + rank)
        self.parena = torch.randn(sum(sizes), generator=g)
        for (p, off, n) in self.entries:
            p.data = self.parena[off:off + n]
        self.outside = torch.nn.Parameter(torch.randn(7, generator=g))          # like resnet.fc: no gradient, not in the arena
        self.buf = torch.randn(5, generator=g)
        self.count = torch.tensor(3 + rank, dtype=torch.long)
        self._grad_offsets = {id(p): off for (p, off, n) in self.entries}

    def param_arena(self):
        return self.parena

    def parameters(self):
        return self.params + [self.outside]

    def buffers(self):
        return [self.buf, self.count]


def _sync_worker(rank, world, port, sizes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rot_mvgaze_amd.dp import GradAllReducer
    m = _FakeReplica(sizes, rank)
    GradAllReducer(m, bucket_mb=0.01)                 # broadcast=True is the default
    torch.save({"arena": m.parena.clone(), "p0": m.params[0].data.clone(), "outside": m.outside.data.clone(),
                "buf": m.buf.clone(), "count": m.count.clone()}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_replicas_are_broadcast_from_rank0(tmp_path):
    """Ranks that start from different weights (each process draws its own random init) end up with rank
    0's parameter arena, out-of-arena parameters and buffers after GradAllReducer(model)."""
    sizes = [1000, 37, 4096, 5]
    port = _free_port()
    mp.spawn(_sync_worker, args=(2, port, sizes, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / f"s{r}.pt", weights_only=True) for r in range(2))
    ref = _FakeReplica(sizes, 0)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["arena"], ref.parena) and torch.equal(a["outside"], ref.outside.data)
    assert int(a["count"]) == 3 and torch.equal(a["p0"], ref.parena[:1000])


def _worker(rank, world, port, sizes, bucket_mb, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rot_mvgaze_amd.dp import GradAllReducer
    m = _FakeModel(sizes, rank)
    local = m.arena.clone()
    red = GradAllReducer(m, bucket_mb=bucket_mb)
    launched = []
    orig = red._launch
    red._launch = lambda b: (launched.append(b), orig(b))[1]
    # publish parameters in grad-ready order, a few at a time (as the backward does)
    i = 0
    for step in (3, 1, 4, 2, 100):
        chunk = m.params[i:i + step]
        if chunk:
            m._on_grads_ready(chunk)
        i += step
    m._on_backward_done()
    # second "step": state must have been reset
    m.arena.copy_(local)
    m._on_grads_ready(m.params)
    m._on_backward_done()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    np.save(os.path.join(out_dir, f"r{rank}.npy"),
            np.array([float((m.arena - want).abs().max()), len(red.buckets), len(launched)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb", [0.001, 0.01, 64.0])
def test_bucketed_allreduce_world2(tmp_path, bucket_mb):
    sizes = [1000, 37, 4096, 5, 512, 2048, 64, 64, 3000, 11, 700]
    port = _free_port()
    mp.spawn(_worker, args=(2, port, sizes, bucket_mb, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        err, nb, nl = np.load(tmp_path / f"r{r}.npy")
        assert err < 1e-6, f"rank {r}: averaged gradient wrong by {err}"
        assert nl == 2 * nb                      # every bucket reduced exactly once per step
        if bucket_mb < 0.002:
            assert nb > 4
        if bucket_mb > 1:
            assert nb == 1


def test_bucket_boundaries_follow_grad_ready_order():
    from rot_mvgaze_amd.dp import GradAllReducer
    m = _FakeModel([300, 300, 300, 300, 100], 0)
    red = GradAllReducer(m, bucket_mb=600 * 4 / (1 << 20))
    red._build()
    assert red.buckets == [[0, 600], [600, 1200], [1200, 1300]]
    fired = []
    red._launch = fired.append
    m._on_grads_ready(m.params[:1])
    assert fired == []
    m._on_grads_ready(m.params[1:3])
    assert fired == [0]                           # bucket 0 complete, bucket 1 still waiting for param 3
    m._on_grads_ready(m.params[3:4])
    assert fired == [0, 1]
    m._on_backward_done()
    assert fired == [0, 1]                        # bucket 2 holds only param 4, which this backward never published: skipped
    # a backward that stops after the first two parameters (nothing flows further down: model.py's dfeat-is-None
    # branch): buckets 1 and 2 hold the previous step's gradients and must not be averaged into anything
    arena, entries = m.grad_arena()
    arena.fill_(7.0)
    del fired[:]
    m._on_grads_ready(m.params[:2])
    assert fired == [0]
    m._on_backward_done()
    assert fired == [0]
    # ... and one that stops in the middle of bucket 1: the published half is reduced, the stale half zeroed first
    del fired[:]
    m._on_grads_ready(m.params[:3])
    m._on_backward_done()
    assert fired == [0, 1]
    (p3, off3, n3) = entries[3]
    assert float(arena[off3:off3 + n3].abs().max()) == 0.0 and float(arena[off3 - 1]) == 7.0     # the unpublished half was zeroed


def test_a_parameter_published_out_of_order_does_not_release_the_buckets_before_it():
    """Completeness is per bucket, from the set of published parameters: publishing a LATER parameter must not
    ship an earlier bucket whose own parameters have not been published (it would average last step's gradient)."""
    from rot_mvgaze_amd.dp import GradAllReducer
    m = _FakeModel([300, 300, 300, 300, 100], 0)
    red = GradAllReducer(m, bucket_mb=600 * 4 / (1 << 20))
    red._build()
    fired = []
    red._launch = fired.append
    m._on_grads_ready(m.params[2:4])              # bucket 1's parameters first
    assert fired == []                            # bucket 0 is not complete: nothing ships (buckets go in arena order)
    m._on_grads_ready(m.params[0:1])
    assert fired == []
    m._on_grads_ready(m.params[1:2])
    assert fired == [0, 1]
    # an unpublished parameter in the MIDDLE (bucket 0 never completes): bucket 0 ships at the end with the stale half zeroed
    arena, entries = m.grad_arena()
    m._on_backward_done()
    arena.fill_(3.0)
    del fired[:]
    m._on_grads_ready([m.params[0]] + m.params[2:5])
    assert fired == []
    m._on_backward_done()
    assert fired == [0, 1, 2]
    (p1, off1, n1) = entries[1]
    assert float(arena[off1:off1 + n1].abs().max()) == 0.0 and float(arena[off1 - 1]) == 3.0 and float(arena[off1 + n1]) == 3.0


def test_buckets_follow_the_models_grad_ready_order():
    """The arena order the reducer buckets over is the order backward finishes parameters in: heads + fusers of iteration
    I-1 .. 0, the lifter, then the backbone from layer4 down to the stem (model.py:_ensure_layout builds it from the same
    spec; checked here on the spec, without a GPU) - so bucket k's all-reduce can start while backward is still above it."""
    from rot_mvgaze_amd.arch import backbone_spec, state_dict_shapes, DEFAULT_VARIANT
    depth, I = 50, 3
    spec = backbone_spec(depth)
    names = []
    for it in range(I - 1, -1, -1):
        names += [f"_gaze_estimators.{it}.", f"_img_fusers.{it}._fuser."]
    names += ["_lifter._lifter."]
    for blk in reversed(spec.blocks):
        cs = [blk.convs[-1]] + list(reversed(blk.convs[:-1])) + ([blk.downsample] if blk.downsample else [])
        names += [c.name for c in cs]
    names += [spec.stem.name]
    # groups of parameters as backward publishes them -> one fake parameter per group, sized like the real ones
    shapes = {n: s for n, s, k in state_dict_shapes(depth, I, DEFAULT_VARIANT)}
    sizes = []
    for pre in names:
        sizes.append(sum(int(np.prod(s)) for n, s in shapes.items() if n.startswith(pre) and "running" not in n and "num_batches" not in n))
    assert sizes[0] > 0 and all(sz > 0 for sz in sizes)
    m = _FakeModel(sizes, 0)
    from rot_mvgaze_amd.dp import GradAllReducer
    red = GradAllReducer(m, bucket_mb=25.0)
    red._build()
    order = []
    red._launch = order.append
    for p in m.params:                              # backward publishes group after group
        m._on_grads_ready([p])
    m._on_backward_done()
    assert order == list(range(len(red.buckets))) and len(red.buckets) >= 5
    # the fusion block's three iterations (60 M of the 89.6 M parameters: one (head, fuser) pair of ~80 MB per bucket at 25 MB
    # granularity) ship before any backbone gradient exists
    head_elems = sum(sizes[:2 * I])
    assert [e0 for (s0, e0) in red.buckets[:I]] == [sum(sizes[:2 * (k + 1)]) for k in range(I)]
    first_backbone_bucket = next(b for b, (s0, e0) in enumerate(red.buckets) if e0 > head_elems + sizes[2 * I])
    assert first_backbone_bucket == I
