#!/usr/bin/env python3
"""Free-running gradient error (relative L2 against the CPU oracle) of one small configuration over several input seeds,
for both conv kernel families: how much of it is the realisation of ReLU flips.  Usage: flip_spread.py depth batch hw [seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rot_mvgaze_amd
from rot_mvgaze_amd import synth
from rot_mvgaze_amd.model import FeatRotationSymm
from rot_mvgaze_amd.losses import IterationLoss, StereoL1Loss
from rot_mvgaze_amd.geometry import rotation_matrix_2d
from oracle import restatement as R

depth, batch, hw = (int(a) for a in sys.argv[1:4])
seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda:0")
sdn = synth.make_state_dict(depth, 0, 3, perturb_bn=True)
for seed in range(21, 21 + seeds):
    inp = synth.make_inputs(batch, 2, seed, hw)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sdn.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    od = {"img_0": img[:, 0].contiguous(), "img_1": img[:, 1].contiguous(), "rot_0": R.rotation_matrix_2d(hp[:, 0]),
          "rot_1": R.rotation_matrix_2d(hp[:, 1]), "gt_gaze": gt[:, 0], "gt_gaze_1": gt[:, 1]}
    od = R.model_forward(sd, od, depth, 3, True)
    ol = R.iteration_loss(od)
    ol.backward()
    line = [f"seed {seed}:"]
    for split in (True, False):
        m = FeatRotationSymm(backbone_depth=depth, num_iter=3)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdn.items()})
        m.to(dev).train()
        m.ensure_layout()
        m._backbone.split = split
        data = {"img_0": img[:, 0].contiguous().to(dev), "img_1": img[:, 1].contiguous().to(dev),
                "rot_0": rotation_matrix_2d(hp[:, 0].contiguous().to(dev)), "rot_1": rotation_matrix_2d(hp[:, 1].contiguous().to(dev)),
                "gt_gaze": gt[:, 0].contiguous().to(dev), "gt_gaze_1": gt[:, 1].contiguous().to(dev)}
        data = m(data)
        loss = IterationLoss(StereoL1Loss(rel_weight=0.01, reference_decay=1.0), iter_decay=0.5)(data)
        loss.backward()
        worst, wk = 0.0, ""
        for k, p in m.named_parameters():
            if leaves[k].grad is None:
                continue
            g, r = p.grad.detach().cpu().double().numpy(), leaves[k].grad.double().numpy()
            e = np.linalg.norm((g - r).ravel()) / (np.linalg.norm(r.ravel()) + 1e-30)
            if e > worst:
                worst, wk = e, k
        pe = float((data["iter_2"]["pred_gaze_0"].detach().cpu() - od["iter_2"]["pred_gaze_0"].detach()).abs().max() / od["iter_2"]["pred_gaze_0"].abs().max())
        line.append(f"{'split' if split else 'fp32mfma'}: loss rel {abs(loss.item() - ol.item()) / abs(ol.item()):.1e} pred {pe:.1e} worst grad L2 {worst:.2e} ({wk.replace('_feat_extractor.0.', '')})")
    print("  ".join(line), flush=True)
