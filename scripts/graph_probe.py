"""Eager step vs hipGraph replay, with and without the side stream (which makes the captured graph branch).
python scripts/graph_probe.py [workload] -> one line per variant: ms per step (wall), host ms to queue a step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import rot_mvgaze_amd  # noqa: F401
from bench import WORKLOADS
from rot_mvgaze_amd import synth
from rot_mvgaze_amd.geometry import rotation_matrix_2d
from rot_mvgaze_amd.graph import GraphedStep
from rot_mvgaze_amd.losses import MultiViewIterationLoss
from rot_mvgaze_amd.model import MultiViewGaze
from rot_mvgaze_amd.optim import Adam

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
depth, V, B, _ = WORKLOADS[wl]
dev = torch.device("cuda:0")


def build(overlap):
    m = MultiViewGaze(depth, 3)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.make_state_dict(depth, 0, 3).items()}, strict=True)
    m.to(dev).train()
    m.ensure_layout()
    m._backbone.overlap_wgrad = overlap
    inp = synth.make_inputs(B, V, 1234, 224)
    img = [torch.from_numpy(np.ascontiguousarray(inp["img"][:, v])).to(dev) for v in range(V)]
    gt = torch.from_numpy(inp["gt_gaze"]).to(dev)
    rot = rotation_matrix_2d(torch.from_numpy(inp["head_pose"]).reshape(-1, 2).to(dev)).reshape(B, V, 3, 3)
    crit = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)
    opt = Adam(m.parameters(), lr=1e-4, weight_decay=1e-6, capturable=True)

    def step():
        m.zero_grad(set_to_none=True)
        loss = crit(m.forward_multiview(img, rot), gt)
        loss.backward()
        opt.step()
        return loss
    return m, opt, step


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host


for overlap in (True, False):
    m, opt, step = build(overlap)
    ms, host = timeit(step)
    print(f"{wl} eager  overlap={overlap}: {ms:8.3f} ms/step, host {host:8.3f} ms", flush=True)
    gs = GraphedStep(m, step, opt, warmup=2)
    ms, host = timeit(gs.run)
    print(f"{wl} graph  overlap={overlap}: {ms:8.3f} ms/step, host {host:8.3f} ms", flush=True)
    del gs, m, opt, step
    torch.cuda.empty_cache()
