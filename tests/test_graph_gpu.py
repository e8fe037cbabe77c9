"""A training step captured in a hipGraph (rot_mvgaze_amd.graph.GraphedStep) replays to the same bits as the eager step:
same kernels, same order, the optimizer's step counter / learning rate read from the device (optim.Adam(capturable=True)).
Caller contract mirrored: /root/reference/trainer.py:119-147 (model(data), loss, zero_grad, backward, optimizer.step; CyclicLR
moves the learning rate between steps)."""
import numpy as np
import pytest
import torch

from rot_mvgaze_amd import synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _setup(depth, V, B, hw, capturable=True, bf16=False):
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    from rot_mvgaze_amd.losses import MultiViewIterationLoss
    from rot_mvgaze_amd.model import MultiViewGaze
    from rot_mvgaze_amd.optim import Adam
    m = MultiViewGaze(depth, 3)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.make_state_dict(depth, 0, 3).items()}, strict=True)
    m.to(dev()).train()
    if bf16:
        m.compute_dtype = torch.bfloat16
    inp = synth.make_inputs(B, V, 77, hw)
    img = [torch.from_numpy(np.ascontiguousarray(inp["img"][:, v])).to(dev()) for v in range(V)]
    gt = torch.from_numpy(inp["gt_gaze"]).to(dev())
    rot = rotation_matrix_2d(torch.from_numpy(inp["head_pose"]).reshape(-1, 2).to(dev())).reshape(B, V, 3, 3)
    crit = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)
    opt = Adam(m.parameters(), lr=1e-3, weight_decay=1e-6, capturable=capturable)

    def step():
        m.zero_grad(set_to_none=True)
        loss = crit(m.forward_multiview(img, rot), gt)
        loss.backward()
        opt.step()
        return loss
    return m, opt, step


@pytest.mark.parametrize("depth,V,B,hw,bf16", [(18, 2, 4, 64, False), (50, 3, 2, 64, False), (50, 2, 2, 64, True)],
                         ids=["r18_fp32", "r50_fp32", "r50_bf16_storage"])
def test_graphed_step_replays_the_eager_step_bit_for_bit(depth, V, B, hw, bf16):
    from rot_mvgaze_amd.graph import GraphedStep
    warm, n = 2, 3
    lrs = [1e-3, 5e-4, 2e-3]
    m1, o1, s1 = _setup(depth, V, B, hw, bf16=bf16)
    eager_losses = []
    for _ in range(warm):
        s1()
    for k in range(n):
        o1.param_groups[0]["lr"] = lrs[k]            # a scheduler stepping between iterations
        eager_losses.append(float(s1().item()))
    m2, o2, s2 = _setup(depth, V, B, hw, bf16=bf16)
    gs = GraphedStep(m2, s2, o2, warmup=warm)
    graph_losses = []
    for k in range(n):
        o2.param_groups[0]["lr"] = lrs[k]
        graph_losses.append(float(gs.run().item()))
    assert graph_losses == eager_losses, (graph_losses, eager_losses)
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    assert o2.state_dict()["mvg_arena_state"][0]["step"] == warm + n == o1.state_dict()["mvg_arena_state"][0]["step"]


def test_device_side_adam_is_the_host_side_adam():
    """mvg_adam_step_dev (step counter, bias corrections and lr on the device) == mvg_adam_step for the same state."""
    m1, o1, s1 = _setup(18, 2, 4, 64, capturable=True)
    m3, o3, s3 = _setup(18, 2, 4, 64, capturable=False)
    for _ in range(2):
        s1()
        s3()
        for (k1, p1), (_, p3) in zip(m1.named_parameters(), m3.named_parameters()):
            assert torch.allclose(p1, p3, rtol=2e-6, atol=1e-8), k1
        # (same parameters for the next round: the two flavours only differ in where the scalars come from)
        m3.load_state_dict(m1.state_dict())


def test_graphed_step_needs_a_capturable_optimizer():
    from rot_mvgaze_amd.graph import GraphedStep
    m, opt, step = _setup(18, 2, 2, 64, capturable=False)
    with pytest.raises(ValueError, match="capturable"):
        GraphedStep(m, step, opt)
