#!/usr/bin/env python3
"""Random (variant, batch, image size) runs of the two-view model against the CPU oracle: loss and last-iteration
predictions at 1e-4 (5e-4 for batches under 4: tiny BatchNorm populations), gradients in relative L2 (ReLU flips).
model_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rot_mvgaze_amd
from rot_mvgaze_amd import synth
from rot_mvgaze_amd.arch import Variant
from rot_mvgaze_amd.model import FeatRotationSymm
from rot_mvgaze_amd.losses import IterationLoss, StereoL1Loss
from rot_mvgaze_amd.geometry import rotation_matrix_2d
from oracle import restatement as R

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
torch.set_num_threads(min(16, os.cpu_count() or 1))
VARS = [{}, {"share_weights": True}, {"ignore_rotmat": True}, {"encode_rotmat": True}, {"share_feature": True},
        {"share_weights": True, "encode_rotmat": True}]
bad = 0
for it in range(cases):
    kw = VARS[int(rng.integers(0, len(VARS)))]
    B = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64, 96]))
    hw = int(rng.choice([40, 64, 96, 128]))
    depth = 18 if rng.random() < 0.8 or B > 17 else 50
    seed = int(rng.integers(0, 1000))
    v = Variant(**kw)
    sdn = synth.make_state_dict(depth, 0, 3, perturb_bn=True, variant=v)
    m = FeatRotationSymm(backbone_depth=depth, num_iter=3, **kw)
    m.load_state_dict({k: torch.from_numpy(np.array(x)) for k, x in sdn.items()}, strict=True)
    m.to(dev).train()
    inp = synth.make_inputs(B, 2, seed, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    data = {"img_0": img[:, 0].contiguous().to(dev), "img_1": img[:, 1].contiguous().to(dev),
            "rot_0": rotation_matrix_2d(hp[:, 0].contiguous().to(dev)), "rot_1": rotation_matrix_2d(hp[:, 1].contiguous().to(dev)),
            "gt_gaze": gt[:, 0].contiguous().to(dev), "gt_gaze_1": gt[:, 1].contiguous().to(dev)}
    crit = IterationLoss(StereoL1Loss(rel_weight=0.01, reference_decay=1.0), iter_decay=0.5)
    data = m(data)
    loss = crit(data)
    loss.backward()
    sd = {k: torch.from_numpy(np.array(x)) for k, x in sdn.items()}
    leaves = {k: x.requires_grad_(True) for k, x in sd.items() if x.is_floating_point() and "running" not in k}
    od = {"img_0": img[:, 0].contiguous(), "img_1": img[:, 1].contiguous(), "rot_0": R.rotation_matrix_2d(hp[:, 0]),
          "rot_1": R.rotation_matrix_2d(hp[:, 1]), "gt_gaze": gt[:, 0], "gt_gaze_1": gt[:, 1]}
    od = R.model_forward(sd, od, depth, 3, True, variant=v)
    ol = R.iteration_loss(od)
    ol.backward()
    # layer4's BatchNorm population per channel is B * (hw / 32)^2: at 4 samples (B = 1, 64 px) ResNet-50's outputs
    # scatter over 5e-5 .. 9e-4 with BOTH conv kernel families (6 seeds each)
    tol = 1e-4 if B >= 4 else (2e-3 if B * (hw // 32) ** 2 <= 4 else 5e-4)
    e_loss = abs(loss.item() - ol.item()) / abs(ol.item())
    p_dev, p_ref = data["pred_gaze"].detach().cpu().double(), od["pred_gaze"].detach().double()
    e_pred = ((p_dev - p_ref).abs().max() / p_ref.abs().max()).item()
    worst = (0.0, "")
    # share_weights: one device Parameter under several state_dict names - the oracle's leaves are per name,
    # so its gradient is the sum over the names
    groups = {}
    for k, p in m.named_parameters(remove_duplicate=False):
        groups.setdefault(id(p), (p, []))[1].append(k)
    for p, names in groups.values():
        refs = [leaves[k].grad for k in names if leaves[k].grad is not None]
        if not refs or p.grad is None:
            assert not refs and p.grad is None, names
            continue
        b = sum(r.double() for r in refs).reshape(-1)
        a = (p.grad.detach().contiguous() if p.grad.dim() == 4 else p.grad.detach()).cpu().double().reshape(-1)
        e = ((a - b).norm() / (b.norm() + 1e-30)).item()
        if e > worst[0]:
            worst = (e, names[0])
    # inference path (BN from running statistics folded into the conv epilogues) on fresh weights
    me = FeatRotationSymm(backbone_depth=depth, num_iter=3, **kw)
    me.load_state_dict({k: torch.from_numpy(np.array(x)) for k, x in sdn.items()}, strict=True)
    me.to(dev).eval()
    with torch.no_grad():
        de = me({k: data[k] for k in ("img_0", "img_1", "rot_0", "rot_1")})
        sde = {k: torch.from_numpy(np.array(x)) for k, x in sdn.items()}
        oe = R.model_forward(sde, {k: od[k] for k in ("img_0", "img_1", "rot_0", "rot_1")}, depth, 3, False, variant=v)
    q_dev, q_ref = de["pred_gaze"].cpu().double(), oe["pred_gaze"].double()
    e_eval = ((q_dev - q_ref).abs().max() / q_ref.abs().max()).item()
    ok = e_loss <= tol and e_pred <= tol and e_eval <= 1e-4 and worst[0] <= (5e-2 if B >= 4 else 0.3)   # B < 4: 4..12-sample BatchNorm
    bad += not ok
    print(("ok  " if ok else "FAIL"), f"R{depth} B{B} hw{hw} {kw}: loss {e_loss:.1e} pred {e_pred:.1e} eval pred {e_eval:.1e} worst grad L2 {worst[0]:.1e} ({worst[1]})", flush=True)
    del m, me, data, loss
print("failures:", bad)
sys.exit(1 if bad else 0)
