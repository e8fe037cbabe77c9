"""Two RCCL ranks on two GPUs (skipped on the one-GPU boxes of this pool; runs wherever the driver has a multi-GPU node):
the branch the data-parallel reducer takes on hardware - dist.ReduceOp.AVG inside the collective (dp.py:_launch) - against
the SUM-then-scale branch the gloo tests cover, and the self-launching bench with N = 2.  Nothing in the reference
corresponds (single device: /root/reference/trainer.py:42)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL between ranks)")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
g = torch.Generator(device="cuda").manual_seed(1234 + rank)
arena = torch.randn(6_000_000, device="cuda", generator=g) * torch.logspace(-6, 2, 6_000_000, device="cuda")
a, b = arena[1000:5_001_000].clone(), arena[1000:5_001_000].clone()       # a 20 MB bucket: a contiguous slice of the arena
dist.all_reduce(a, op=dist.ReduceOp.AVG)
dist.all_reduce(b, op=dist.ReduceOp.SUM)
b.mul_(0.5)
ok = bool(torch.equal(a, b))
print("AVG_EQUALS_HALF_SUM", ok, flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


@needs_two
def test_rccl_avg_is_half_the_sum_bit_for_bit_on_two_ranks(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("AVG_EQUALS_HALF_SUM True" in o for o in outs), outs


@needs_two
def test_bench_two_gpus_runs_over_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c2", "--batch", "8", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["dp"]["backend"] == "nccl"
    assert line["config"]["global_batch"] == 16 and line["value"] > 0
