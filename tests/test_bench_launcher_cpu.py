"""bench.py's self-launcher (`python bench.py --gpus N` with no RANK in the environment): the parent starts N rank
processes, relays rank 0's one JSON line and exits with the worst child's code.  Exercised here with stub rank
scripts (no GPU, no torch in the children); the real N = 2 path runs in tests/test_model_gpu.py on the GPU box."""
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stub(tmp_path, body):
    p = tmp_path / "rank_stub.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launcher_sets_rank_environment_and_relays_rank0(tmp_path):
    script = _stub(tmp_path, """
        import json, os, sys
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"rank{r}.json"), "w").write(json.dumps({k: os.environ.get(k) for k in
            ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}))
        print(json.dumps({"rank": r, "argv": sys.argv[2:]}))       # only rank 0's line may reach the caller's stdout
    """)
    rc, out = bench.launch_ranks(3, [str(tmp_path), "--steps", "2"], script=script)
    assert rc == 0
    assert [json.loads(l) for l in out.splitlines()] == [{"rank": 0, "argv": ["--steps", "2"]}]
    envs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_launcher_propagates_failure_and_stops_the_other_ranks(tmp_path):
    script = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)            # a rank waiting in a collective for the dead one
    """)
    t0 = time.time()
    rc, out = bench.launch_ranks(2, [], script=script)
    assert rc == 7 and out == ""
    assert time.time() - t0 < 60


def test_launcher_timeout(tmp_path):
    script = _stub(tmp_path, "import time; time.sleep(600)")
    t0 = time.time()
    rc, _ = bench.launch_ranks(2, [], script=script, timeout=1.0)
    assert rc == 124 and time.time() - t0 < 60


def test_bare_bench_with_gpus_2_takes_the_launcher_branch(tmp_path):
    """`python bench.py --gpus 2` without RANK must not die on the WORLD_SIZE assert: the parent goes to launch_ranks
    before importing torch.  Without a GPU the real rank processes fail fast (no HIP device), which is enough to
    show the branch: the parent exits non-zero, prints nothing on stdout and no WORLD_SIZE assertion on stderr.
    (With a GPU, tests/test_model_gpu.py runs the same command to completion.)"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env["MVG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "c2", "--batch", "2", "--no-roofline", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert "launch with torch.distributed.run" not in r.stderr and "WORLD_SIZE=" not in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and r.stdout.strip() == ""
