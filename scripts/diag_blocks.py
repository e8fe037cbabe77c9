import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rot_mvgaze_amd
from rot_mvgaze_amd import synth, backbone as BB
from rot_mvgaze_amd.arch import backbone_spec
from oracle import restatement as R
import test_model_gpu as T

depth, batch, hw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = T.build(depth)
# capture per-block input grads on the HIP side
orig_fwd = BB.Backbone.forward
def fwd(self, imgs, training, keep):
    feat, tape = orig_fwd(self, imgs, training, keep)
    if tape is not None:
        tape["debug"] = []
        m._dbg = tape["debug"]
        m._outs = [tape["units"][idx[-1]].out.clone() for (idx, ds) in tape["blocks"]]
    return feat, tape
BB.Backbone.forward = fwd
data = m(T.inputs(batch, hw))
loss = T.metrics()(data); loss.backward()
sd = {k: torch.from_numpy(np.array(v)).double() if v.dtype == np.float32 else torch.from_numpy(np.array(v))
      for k, v in synth.make_state_dict(depth, 0, 3, perturb_bn=True).items()}
for k, v in sd.items():
    if v.is_floating_point() and "running" not in k: v.requires_grad_(True)
inp = synth.make_inputs(batch, 2, 1234, hw)
img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
spec = backbone_spec(depth)
traces = []
feats = []
for v in range(2):
    tr = []
    feats.append(R.backbone_forward(sd, img[:, v].double(), spec, True, tr))
    for t in tr: t.retain_grad()
    traces.append(tr)
f0, f1 = R.lift(sd, feats[0]), R.lift(sd, feats[1])
od = R.fuse_pair(sd, 3, feats[0], feats[1], f0, f1, R.rotation_matrix_2d(hp[:, 0]).double(), R.rotation_matrix_2d(hp[:, 1]).double())
od["gt_gaze"], od["gt_gaze_1"] = gt[:, 0], gt[:, 1]
ol = R.iteration_loss(od); ol.backward()
nblk = len(spec.blocks)
# m._dbg[i] = grad wrt input of block (nblk-1-i)  == oracle trace[nblk-1-i]
for i, g in enumerate(m._dbg):
    bi = nblk - 1 - i
    ref = torch.stack([traces[v][bi].grad for v in range(2)]).permute(0, 1, 3, 4, 2)
    err = (g.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"grad wrt input of block {bi} ({spec.blocks[bi].prefix}) shape {tuple(g.shape)}: rel err {err:.2e}")

for bi in range(nblk):
    ref = torch.stack([traces[v][bi + 1] for v in range(2)]).permute(0, 1, 3, 4, 2).detach()
    mine = m._outs[bi].cpu().double()
    flips = ((mine > 0) != (ref > 0)).sum().item()
    fl_idx = ((mine > 0) != (ref > 0)).nonzero()[:3].tolist()
    vals = [(mine[tuple(i)].item(), ref[tuple(i)].item()) for i in fl_idx]
    print(f"block {bi} out: max abs diff {(mine-ref).abs().max().item():.2e} (scale {ref.abs().max().item():.2e}); relu flips {flips} {vals}")
