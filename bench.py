#!/usr/bin/env python3
"""Benchmark of the Rot-MVGaze hot path on MI355X: multi-view samples/s, forward + loss + backward
(+ gradient all-reduce when --gpus > 1), synthetic B x V x 3 x 224 x 224 batches.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W        # self-launching: starts N rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no RANK in the environment, this process never touches the GPU: it starts one
child process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set),
relays rank 0's single JSON line and exits with the worst child's return code (launch_ranks below).

Prints ONE JSON line (rank 0).  `value` = samples all ranks processed / max-over-ranks time of
exactly K steps bracketed by barrier + synchronize; inputs are resident in HBM before the timed
region.  `roofline` is measured live with HIP events around every launch of the dominant kernel
family (the fp32-MFMA implicit-GEMM convolutions) in extra, separately run profiled steps;
`cpu_baseline` times the CPU oracle (a port of the reference path) on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (depth, views, per-GPU batch, description)  - BASELINE.json configs
    "c2": (18, 2, 64, "C2: ResNet-18, V=2, B=64 per GPU, 3x224x224, fwd+loss+bwd"),
    "c3": (50, 4, 128, "C3: ResNet-50, V=4, B=128 per GPU, 3x224x224, fwd+loss+bwd"),
    "c4": (50, 4, 32, "C4: ResNet-50, V=4, B=32 per GPU (256 on 8 GPUs), 3x224x224, fwd+loss+bwd"),
    "r50v2": (50, 2, 64, "ResNet-50, V=2, B=64 per GPU, 3x224x224, fwd+loss+bwd"),
    "c5": (50, 8, 64, "C5: ResNet-50, V=8, B=64 per GPU (512 on 8 GPUs), 3x224x224, fwd+loss+bwd, bf16 MFMA path "
                      "(bf16 activations + weight copies in the backbone; fp32 statistics, fusion block, loss, master weights)"),
    "c5fp32": (50, 8, 64, "C5 shapes in fp32: ResNet-50, V=8, B=64 per GPU (512 on 8 GPUs), 3x224x224, fwd+loss+bwd"),
}
BF16_WORKLOADS = {"c5"}
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA (the 5 PF headline includes 2:1 sparsity)
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def sample_flops(depth, views):
    """Algorithmic fwd+bwd FLOPs per multi-view sample (BASELINE.md §3 / SURVEY.md §8(d))."""
    bb = {18: 10.65e9, 50: 24.29e9}[depth]
    lifter = {18: 6.29e6, 50: 11.01e6}[depth]
    pair = {18: 100.7e6, 50: 242.2e6}[depth]
    npairs = views * (views - 1) // 2
    return views * bb + 3 * (views * lifter + npairs * pair)


def _oracle_time(depth, views, batch, threads, fwd_only, max_steps, seconds_budget):
    """Median ms/step of the CPU oracle (checker / baseline only - never the product path)."""
    import numpy as np
    import torch
    from rot_mvgaze_amd import synth
    from oracle import restatement as R
    torch.set_num_threads(threads)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_state_dict(depth, 0, 3).items()}
    if not fwd_only:
        for k, v in sd.items():
            if v.dtype == torch.float32 and "running" not in k and ".fc." not in k:
                v.requires_grad_(True)
    inp = synth.make_inputs(batch, views, 1234, 224)
    img, hp, gt = (torch.from_numpy(inp[k]) for k in ("img", "head_pose", "gt_gaze"))
    rot = R.rotation_matrix_2d(hp.reshape(-1, 2)).reshape(batch, views, 3, 3)

    def step():
        if fwd_only:
            with torch.no_grad():
                R.multiview_forward(sd, img, rot, depth, 3, False)       # eval forward ("eval forward")
            return
        for v in sd.values():
            if v.requires_grad:
                v.grad = None
        out = R.multiview_forward(sd, img, rot, depth, 3, True)
        R.multiview_loss(out, gt).backward()

    t0 = time.time()
    step()                                            # warm-up (allocator, oneDNN primitive cache)
    warm = time.time() - t0
    if warm > seconds_budget:                         # a step costs more than the whole budget (oversubscribed threads):
        return warm * 1e3, 1                          # report the one step there is, do not run more
    times = []
    t_end = time.time() + seconds_budget
    while len(times) < max_steps and (time.time() < t_end or len(times) < 2):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    times.sort()
    return times[len(times) // 2] * 1e3, len(times)


def cpu_baseline(depth, views):
    """The CPU oracle (restatement of the reference path, validated against the reference's own
    outputs in tests/) on a bounded sample of the workload - B = 8 samples, fwd+loss+bwd, 16 host threads
    (a one-GPU box's CPU share) - plus the shapes BASELINE.md 4 names: C1 (ResNet-18, V=2, B=8,
    forward only; also on ALL of the box's threads, os.cpu_count()) and fwd+bwd at B=8 for ResNet-18 /
    ResNet-50 (V=2), each on 16 threads and on 8 (the survey container's figure)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))         # a one-GPU box's CPU share; oversubscribing oneDNN is far slower
    B = 8
    ms, n = _oracle_time(depth, views, B, cores, False, 5, 12.0)
    out = {"value": B / (ms * 1e-3), "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"B=8 of the workload's samples (V={views}, ResNet-{depth}, 224x224), fwd+loss+bwd, "
                     f"{n} steps, median {ms:.0f} ms/step, torch CPU {cores} threads",
           "host_cpus": avail, "shapes": {}}
    plan = [("c1_r18_v2_b8_eval_fwd", 18, 2, True, 5, 2.0), ("r18_v2_b8_fwd_bwd", 18, 2, False, 5, 3.0),
            ("r50_v2_b8_fwd_bwd", 50, 2, False, 3, 5.0)]
    for name, d, v, fwd_only, steps, budget in plan:
        # os.cpu_count() threads AND 8 threads.  16 = a one-GPU box's CPU share is timed as well.
        for thr in sorted({cores, min(8, cores)}, reverse=True):
            m, k = _oracle_time(d, v, B, thr, fwd_only, steps, budget)
            out["shapes"][f"{name}_{thr}thr"] = {"ms_per_step": round(m, 1), "samples_per_s": round(B / (m * 1e-3), 2),
                                                 "threads": thr, "steps": k}
    # the all-threads leg (256 on this pool's hosts: oneDNN oversubscribes the box's 16-CPU share and one B = 8 forward took 36 s
    # in round 3 = 57 % of the whole bench run): ONE B = 2 forward of the C1 shape in a child process with a hard 10 s limit -
    # enough to state the oversubscription, bounded whatever the host does
    if avail > cores:
        import subprocess
        code = ("import sys, json; sys.path.insert(0, %r); import bench; "
                "m, k = bench._oracle_time(18, 2, 2, %d, True, 1, 0.0); print(json.dumps([m, k]))" % (ROOT, avail))
        key = f"c1_r18_v2_b2_eval_fwd_{avail}thr"
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=10)
            m, k = json.loads(r.stdout.strip().splitlines()[-1])
            out["shapes"][key] = {"ms_per_step": round(m, 1), "samples_per_s": round(2 / (m * 1e-3), 2), "threads": avail, "steps": k,
                                  "batch": 2}
        except subprocess.TimeoutExpired:
            out["shapes"][key] = {"ms_per_step": None, "threads": avail, "batch": 2,
                                  "note": "one B = 2 forward did not finish within the 10 s limit (thread oversubscription)"}
        except Exception as e:                                   # the leg is informative: never fail the bench on it
            out["shapes"][key] = {"ms_per_step": None, "threads": avail, "note": f"not measured: {type(e).__name__}"}
    return out


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(world, argv, script=None, extra_env=None, timeout=None, python=None):
    """Start `world` rank processes of `script` (default: this file) with `argv`, one per GPU, and wait.

    The caller must not have initialised the GPU (nothing here imports torch): the children are fresh
    processes, never an exec of this one.  Rank r gets RANK = LOCAL_RANK = r, WORLD_SIZE = world,
    MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT (what torch.distributed.run would set).  Rank 0's stdout is
    captured and returned - the contract is ONE JSON line - every other rank's stdout joins stderr.  When a
    rank fails, the remaining ones (who would wait in a collective forever) are terminated.  Returns
    (worst return code, rank 0's stdout text)."""
    import subprocess
    script = script or os.path.abspath(__file__)
    env0 = dict(os.environ)
    env0.update({"WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                 "LOCAL_WORLD_SIZE": str(world)})
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this pool
    env0.update(extra_env or {})
    procs = []
    for r in range(world):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([python or sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(),
                                      stderr=None, text=(r == 0) or None))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = None if timeout is None else time.time() + timeout
    worst, failed = 0, False
    alive = set(range(world))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0:
                worst = max(worst, rc if rc > 0 else 128 - rc)      # killed by signal s: 128 + s, like a shell
                failed = True
        if (failed or (deadline is not None and time.time() > deadline)) and alive:
            if not failed:
                worst = 124
            for r in alive:                   # exactly the processes started above
                procs[r].terminate()
            t_kill = time.time() + 10
            for r in sorted(alive):
                try:
                    procs[r].wait(max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            alive.clear()
            break
        if alive:
            time.sleep(0.05)
    reader.join(10)
    return worst, (out0[0] if out0 else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS),
                    help="default: the largest single-GPU configuration of BASELINE.json (C3)")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=25.0, help="data-parallel gradient bucket size")
    # one-rank RCCL rehearsal at C3 (MVG_FORCE_DIST=1, profiles/README.md): plain 159.2 ms; 0 reserved CUs / 4 channels
    # 163.3; 4 / 4 165.0; 8 / 8 165.3; 12 / 8 166.1.  The gradient stream is 358 MB per ~160 ms step: a few GB/s, far
    # below what 4 channels move over xGMI, so the default keeps RCCL small (one workgroup per channel).
    ap.add_argument("--reserved-cus", type=int, default=int(os.environ.get("MVG_RESERVED_CUS", "4")),
                    help="N > 1: CUs the persistent conv grids leave to the RCCL kernels")
    ap.add_argument("--nccl-channels", type=int, default=int(os.environ.get("NCCL_MAX_NCHANNELS", "4")),
                    help="N > 1: NCCL_MAX_NCHANNELS for the gradient all-reduce")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward+loss+backward only (no Adam step)")
    ap.add_argument("--graph", action="store_true",
                    help="N = 1 training: capture zero_grad + forward + loss + backward + Adam once in a hipGraph and replay it "
                         "(rot_mvgaze_amd/graph.py).  Host time per step drops from ~20 ms to 0.15 ms; the GPU time does not (the eager "
                         "step is GPU-bound) and the capture is single-stream - a captured side stream replays 1.4-1.7x slower on "
                         "ROCm 7.0 - so the weight-gradient overlap (+6 %) is lost: off by default (profiles/README.md, round 4)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the backward-weight kernels on the compute stream (per-kernel profiling: rocprofv3 --stats)")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"],
                    help="backbone storage / matrix-core type (default: f32, bf16 for --workload c5)")
    ap.add_argument("--mode", default="train", choices=["train", "eval"],
                    help="eval = inference forward only (model.eval(), BN folded into the convs; SURVEY §8(f) rank 2)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher (no torch / HIP in this process)
        rc, out = launch_ranks(args.gpus, sys.argv[1:])
        sys.stdout.write(out)
        sys.stdout.flush()
        sys.exit(rc)

    import numpy as np
    import torch
    import torch.distributed as dist
    import rot_mvgaze_amd  # noqa: F401
    from rot_mvgaze_amd import ops, synth
    from rot_mvgaze_amd.dp import GradAllReducer
    from rot_mvgaze_amd.losses import MultiViewIterationLoss
    from rot_mvgaze_amd.model import MultiViewGaze

    depth, V, B, desc = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
        desc += f" [batch overridden to {B}]"
    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("MVG_FORCE_DIST") == "1"      # rehearse the RCCL path with one rank
    saved_stdout = None
    if world > 1 or force_dist:
        # RCCL prints a version banner on stdout when the communicator comes up; the contract is ONE JSON line on
        # stdout, so everything until the final print goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if world > 1:
            assert int(os.environ.get("WORLD_SIZE", "1")) == world, \
                f"--gpus {world} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: start N ranks (torch.distributed.run) or none (self-launch)"
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # MVG_DIST_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (the
        # ranks share the devices, gradients travel through the host); never a measurement
        backend = os.environ.get("MVG_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        # the gradient stream is ~160-360 MB per step: 8 RCCL channels move it inside backward and leave
        # the CUs to the persistent conv kernels (see rot_mvgaze_amd/dp.py)
        os.environ["NCCL_MAX_NCHANNELS"] = str(args.nccl_channels)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    # ---- model (same seed on all ranks) and per-rank synthetic data, resident in HBM
    model = MultiViewGaze(depth, 3)
    sd = synth.make_state_dict(depth, 0, 3)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    del sd
    model.to(dev).train()
    bf16 = (args.dtype == "bf16") if args.dtype else (args.workload in BF16_WORKLOADS)
    if bf16:
        model.compute_dtype = torch.bfloat16
    inp = synth.make_inputs(B, V, 1234 + rank, 224)
    img = [torch.from_numpy(np.ascontiguousarray(inp["img"][:, v])).to(dev) for v in range(V)]   # one tensor per view
    gt = torch.from_numpy(inp["gt_gaze"]).to(dev)
    from rot_mvgaze_amd.geometry import rotation_matrix_2d
    rot = rotation_matrix_2d(torch.from_numpy(inp["head_pose"]).reshape(-1, 2).to(dev)).reshape(B, V, 3, 3)
    del inp
    criterion = MultiViewIterationLoss(rel_weight=0.01, reference_decay=1.0, iter_decay=0.5)
    from rot_mvgaze_amd.optim import Adam
    # N = 1 training: the whole step is captured in a hipGraph (the optimizer's step counter / lr then live on the device)
    use_graph = world == 1 and not force_dist and args.mode == "train" and args.graph and not args.no_optimizer
    optimizer = None if args.no_optimizer else Adam(model.parameters(), lr=1e-4, weight_decay=1e-6, capturable=use_graph)   # trainer.py:54
    reducer = GradAllReducer(model, bucket_mb=args.bucket_mb, force=force_dist, reserved_cus=args.reserved_cus) \
        if (world > 1 or force_dist) else None

    if args.mode == "eval":
        model.eval()
        optimizer = None
    if args.no_overlap:
        model.ensure_layout()
        model._backbone.overlap_wgrad = False

    def step():
        if args.mode == "eval":
            with torch.no_grad():
                out = model.forward_multiview(img, rot)
            return out["pred_gaze"].sum()
        model.zero_grad(set_to_none=True)
        out = model.forward_multiview(img, rot)
        loss = criterion(out, gt)
        loss.backward()
        if optimizer is not None:
            optimizer.step()
        return loss

    def fence():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run = step
    graphed = None
    if use_graph:
        from rot_mvgaze_amd.graph import GraphedStep
        graphed = GraphedStep(model, step, optimizer, warmup=2)       # 2 eager steps, then the capture (which executes nothing)
        run = graphed.run
    for _ in range(args.warmup):
        run()
    fence()
    # hipEvents on the compute stream at every step boundary (recording an event costs no synchronisation):
    # the per-step durations give the median SURVEY 8(d) asks for next to the wall-clock mean
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        loss = run()
    marks[args.steps].record()
    host_ms = (time.perf_counter() - t0) / max(args.steps, 1) * 1e3      # host time to QUEUE a step (no synchronisation inside)
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if step_ms else None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    loss_val = float(loss.item())

    # ---- roofline of the dominant kernel family (separate, event-instrumented steps)
    roofline, families = None, None
    if not args.no_roofline:
        nprof = min(args.steps, 3)
        # per-kernel durations are taken with the kernels alone on the GPU: the timed region overlaps the
        # backward-weight kernels (side stream) with the rest of backward, which stretches every
        # overlapped kernel's own start-to-end time without saying anything about the kernel
        bb = getattr(model, "_backbone", None)
        was = bb.overlap_wgrad if bb is not None else None
        if bb is not None:
            bb.overlap_wgrad = False
        ops.prof_reset()
        ops.prof_enable(True)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        ops.prof_enable(False)
        if bb is not None:
            bb.overlap_wgrad = was
        prof = ops.prof_collect()
        conv = [prof[k] for k in ("conv_fprop", "conv_dgrad", "conv_wgrad") if k in prof]
        flops = sum(e["flops"] for e in conv)
        ms = sum(e["ms"] for e in conv)
        launches = sum(e["launches"] for e in conv)
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic, traffic_src = None, None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_traffic_{args.workload}.json")))
        if cands and not args.batch and not args.dtype:
            with open(cands[-1]) as f:
                tj = json.load(f)
            traffic, traffic_src = tj["traffic_bytes_per_launch"], os.path.relpath(cands[-1], ROOT)
        # fp32 training steps run the split-operand kernels (conv_split.hip: three fp16 MFMAs per fp32-accurate product),
        # so their matrix-pipe ceiling is the fp16 MFMA peak / 3 (fp32-equivalent FLOP/s); MVG_SPLIT=0: the fp32 MFMA
        split = (not bf16) and bb is not None and getattr(bb, "split", False) and \
            (args.mode == "train" or getattr(bb, "split_eval", False))
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else (PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_FP32_MFMA_TFLOPS)
        kernels = "bf16" if bf16 else ("split" if split else "fp32mfma")
        if traffic is not None and tj.get("kernels", "fp32mfma") != kernels:
            traffic, traffic_src = None, None           # the committed counters belong to another kernel family
        roofline = {"bound": "mfma",
                    "kernel": ("igemm_bf16_kernel/wgrad_bf16_kernel (bf16 MFMA implicit-GEMM conv: fprop+dgrad+wgrad)" if bf16 else
                               "igemm_split16_kernel/wgrad_split_kernel (fp32-accurate implicit-GEMM conv on the fp16 MFMA: operands as two fp16 "
                               "pieces, three MFMAs per product; the 3-channel stem as row windows on the same kernels: fprop+dgrad+wgrad)" if split else
                               "igemm_kernel/wgrad_kernel (fp32 MFMA implicit-GEMM conv: fprop+dgrad+wgrad)"),
                    "peak_basis": ("dense bf16 MFMA 2500 TFLOP/s" if bf16 else
                                   "dense fp16 MFMA 2500 TFLOP/s / 3 MFMAs per fp32-accurate product = 833.3 fp32-equivalent TFLOP/s (round 2's six-bf16-MFMA "
                                   "kernels: 416.7; the MFMA-dense loops of this chip hold 1.5-1.7 of 2.4 GHz under load: profiles/README.md)"
                                   if split else "dense fp32 MFMA (v_mfma_f32_32x32x2_f32) 157.3 TFLOP/s"),
                    "frac_of_fp32_mfma_peak": None if bf16 else round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                    "frac_of_round2_six_product_ceiling": round(achieved / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4) if split else None,
                    "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE*2 + WRITE_SIZE, separate passes)",
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": sum(e["bytes"] for e in conv) / max(launches, 1),
                    "flop_per_launch": flops / max(launches, 1), "avg_launch_ms": ms / max(launches, 1),
                    "launches_per_step": launches / nprof, "kernel_ms_per_step": ms / nprof}
        families = {k: {"ms_per_step": round(e["ms"] / nprof, 4), "launches_per_step": e["launches"] / nprof,
                        "tflops": round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 2) if e["ms"] > 0 and e["flops"] > 0 else None,
                        "gbs": round(e["bytes"] / (e["ms"] * 1e-3) / 1e9, 1) if e["ms"] > 0 and e["bytes"] > 0 else None}
                    for k, e in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}

    # ---- the fusion block's GEMM family (lifter + fusers + heads: fprop / dgrad / wgrad launches), priced against
    # BOTH roofs (SURVEY 8(d) "Bounding roofline"): fp32 MFMA and the HBM stream of weights + activations
    roofline_fusion = None
    if families is not None:
        lin = [prof[k] for k in ("linear_fprop", "linear_dgrad", "linear_wgrad") if k in prof]
        lms = sum(e["ms"] for e in lin)
        if lms > 0:
            lfl, lby = sum(e["flops"] for e in lin), sum(e["bytes"] for e in lin)
            tf, gbs = lfl / (lms * 1e-3) / 1e12, lby / (lms * 1e-3) / 1e9
            D = V * (V - 1)
            # the ceiling follows the kernels that ran: split-operand Linears (D * B >= 1024 rows: three fp16 MFMAs per product
            # -> 833.3 fp32-equivalent TFLOP/s), the bf16 path's mixed Linears (bf16 MFMA), else the fp32 MFMA
            from rot_mvgaze_amd.heads import SPLIT_MIN_ROWS
            lin_split = (not bf16) and split and D * B >= SPLIT_MIN_ROWS
            lin_peak = PEAK_BF16_MFMA_TFLOPS if bf16 else (PEAK_BF16_MFMA_TFLOPS / 3.0 if lin_split else PEAK_FP32_MFMA_TFLOPS)
            roofline_fusion = {
                "kernel": "fusion-block GEMMs (lifter, fusers, gaze heads: linear_fprop + linear_dgrad + linear_wgrad)",
                "kernels": "bf16-mixed" if bf16 else ("split (fusers / heads; the lifter's V*B rows on the fp32 MFMA)" if lin_split else "fp32mfma"),
                "rows_per_gemm": D * B, "achieved_tflops": round(tf, 2), "peak_tflops": round(lin_peak, 1),
                "peak_basis": ("dense bf16 MFMA" if bf16 else "dense fp16 MFMA / 3 products per fp32-accurate product" if lin_split
                               else "dense fp32 MFMA"),
                "frac_of_peak": round(tf / lin_peak, 4),
                "achieved_gbs": round(gbs, 1), "frac_hbm": round(gbs / PEAK_HBM_GBS, 4),
                "bound": "mfma" if tf / lin_peak >= gbs / PEAK_HBM_GBS else "hbm",
                "intensity_flop_per_byte": round(lfl / lby, 1), "machine_balance_flop_per_byte": round(lin_peak * 1e3 / PEAK_HBM_GBS, 1),
                "ms_per_step": round(lms / nprof, 4),
                "launches_per_step": sum(e["launches"] for e in lin) / nprof,
                "block_ms_per_step": round(sum(prof[k]["ms"] for k in ("linear_fprop", "linear_dgrad", "linear_wgrad",
                                                                        "rotcat", "colsum", "loss", "geometry") if k in prof) / nprof, 4),
                "block_launches_per_step": sum(prof[k]["launches"] for k in ("linear_fprop", "linear_dgrad", "linear_wgrad",
                                                                             "rotcat", "colsum", "loss", "geometry") if k in prof) / nprof,
                "note": "algorithmic bytes = operands + result of every GEMM once (weights dominate); block_* = the Linears + the operand "
                        "builders (rotcat) + the gradient splits (colsum) + loss + geometry launches"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(depth, V)

    if rank == 0:
        line = {
            "metric": "multi-view samples/sec (fwd+bwd), BxVx3x224x224" if args.mode == "train" else
                      "multi-view samples/sec (inference forward), BxVx3x224x224",
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_step_median_events": round(median_ms, 3) if median_ms else None,
            "ms_per_step_events_min_max": [round(step_ms[0], 3), round(step_ms[-1], 3)] if step_ms else None,
            "host_ms_per_step_queueing": round(host_ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": desc, "backbone": f"ResNet-{depth}", "views": V, "batch_per_gpu": B,
                       "global_batch": B * world, "image": "3x224x224", "parallelism": f"dp{world}" + ("" if os.environ.get("MVG_DIST_BACKEND", "nccl") == "nccl" else " (REHEARSAL: gloo, ranks share devices - not a measurement)"),
                       "timed_region": ("inference forward under torch.no_grad() (BN folded into the conv epilogues)" if args.mode == "eval" else
                                        "forward + loss + backward" + (" + RCCL gradient all-reduce" if world > 1 else "") +
                                        ("" if args.no_optimizer else " + fused Adam step")),
                       "weights": "random init, seed 0 (kaiming-normal convs, default Linear)",
                       "rccl_ranks": (dist.get_world_size() if (world > 1 or force_dist) else 1),
                       "step_launch": ("one hipGraphLaunch per step (zero_grad + forward + loss + backward + Adam captured once; "
                                       "rot_mvgaze_amd/graph.py)" if graphed is not None else "eager: every kernel queued from Python"),
                       "dp": ({"bucket_mb": args.bucket_mb, "buckets": len(reducer.buckets) if hasattr(reducer, "buckets") else None,
                               "reserved_cus": reducer.reserved_cus, "nccl_max_nchannels": int(os.environ.get("NCCL_MAX_NCHANNELS", "0")),
                               "backend": os.environ.get("MVG_DIST_BACKEND", "nccl"), "params_broadcast_from_rank0": True}
                              if reducer is not None else None),
                       "images_per_s": round(value * V, 1), "loss": loss_val,
                       "model_tflops": round(value * sample_flops(depth, V) / 1e12, 2)},
            "roofline": roofline, "roofline_fusion": roofline_fusion, "cpu_baseline": cpu, "kernel_families": families,
        }
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
