import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import rot_mvgaze_amd
from rot_mvgaze_amd import synth
from rot_mvgaze_amd.losses import MultiViewIterationLoss
from rot_mvgaze_amd.model import MultiViewGaze
from rot_mvgaze_amd.optim import Adam
from rot_mvgaze_amd.geometry import rotation_matrix_2d
dev = torch.device("cuda:0")
B, V = 64, 2
model = MultiViewGaze(18, 3)
sd = synth.make_state_dict(18, 0, 3)
model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
model.to(dev).train()
inp = synth.make_inputs(B, V, 1234, 224)
img = [torch.from_numpy(np.ascontiguousarray(inp["img"][:, v])).to(dev) for v in range(V)]
gt = torch.from_numpy(inp["gt_gaze"]).to(dev)
rot = rotation_matrix_2d(torch.from_numpy(inp["head_pose"]).reshape(-1, 2).to(dev)).reshape(B, V, 3, 3)
crit = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)
opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
def step():
    model.zero_grad(set_to_none=True)
    out = model.forward_multiview(img, rot)
    loss = crit(out, gt)
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step", flush=True)
# host-only: time with GPU far behind? measure CPU time of one step via process_time
c0 = time.process_time(); 
for _ in range(20): step()
c1 = time.process_time(); torch.cuda.synchronize()
print(f"cpu time {1e3*(c1-c0)/20:.2f} ms/step")
torch.cuda.synchronize()
ts = []
for _ in range(6):
    t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("per-step enqueue after a sync (ms):", " ".join(f"{1e3*t:.2f}" for t in ts))
torch.cuda.synchronize()
for it in range(4):
    t = [time.perf_counter()]
    model.zero_grad(set_to_none=True); t.append(time.perf_counter())
    out = model.forward_multiview(img, rot); t.append(time.perf_counter())
    loss = crit(out, gt); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    print(f"step {it}: zero {1e3*(t[1]-t[0]):.2f} fwd {1e3*(t[2]-t[1]):.2f} loss {1e3*(t[3]-t[2]):.2f} bwd {1e3*(t[4]-t[3]):.2f} opt {1e3*(t[5]-t[4]):.2f}", flush=True)
torch.cuda.synchronize()

# the reference's own API (FeatRotationSymm(dict) + IterationLoss(StereoL1Loss), main.py:231-240)
from rot_mvgaze_amd.model import FeatRotationSymm
from rot_mvgaze_amd.losses import StereoL1Loss, IterationLoss
m2 = FeatRotationSymm(18, 3)
m2.load_state_dict(model.state_dict(), strict=True)
m2.to(dev).train()
crit2 = IterationLoss(StereoL1Loss(rel_weight=0.01, reference_decay=1.0), iter_decay=0.5)
opt2 = Adam(m2.parameters(), lr=1e-4, weight_decay=1e-6)
img_nchw = [i for i in img]
def data():
    return {"img_0": img_nchw[0], "img_1": img_nchw[1], "rot_0": rot[:, 0].contiguous(), "rot_1": rot[:, 1].contiguous(),
            "gt_gaze": gt[:, 0].contiguous(), "gt_gaze_1": gt[:, 1].contiguous()}
for _ in range(3):
    opt2.zero_grad(); d = m2(data()); crit2(d).backward(); opt2.step()
torch.cuda.synchronize()
for it in range(4):
    t = [time.perf_counter()]
    opt2.zero_grad(); t.append(time.perf_counter())
    d = m2(data()); t.append(time.perf_counter())
    loss = crit2(d); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt2.step(); t.append(time.perf_counter())
    print(f"dict API step {it}: zero {1e3*(t[1]-t[0]):.2f} fwd {1e3*(t[2]-t[1]):.2f} loss {1e3*(t[3]-t[2]):.2f} bwd {1e3*(t[4]-t[3]):.2f} opt {1e3*(t[5]-t[4]):.2f}", flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    opt2.zero_grad(); d = m2(data()); crit2(d).backward(); opt2.step()
torch.cuda.synchronize()
print(f"dict API: {1e3*(time.perf_counter()-t0)/20:.2f} ms/step")

# is the host ahead of the GPU with slack?  add host-only delay per step and watch the step time
for delay in (0.0, 0.002, 0.005, 0.008):
    torch.cuda.synchronize()
    for _ in range(3): step()
    t0 = time.perf_counter()
    for _ in range(20):
        time.sleep(delay) if delay else None
        step()
    torch.cuda.synchronize()
    print(f"host delay {1e3*delay:.0f} ms/step -> {1e3*(time.perf_counter()-t0)/20:.2f} ms/step", flush=True)
