#!/usr/bin/env python3
"""Headline benchmark: DDPM sampling throughput of the Moving-MNIST-shape model on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                        # starts the N ranks itself (torch.distributed.run, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      # or under a launcher

    python bench.py --gpus N --global-batch 256         # BASELINE configs[2]

A plain `--gpus N` is the weak-scaling run (BASELINE configs[1] on every GPU, 16 videos each); the strong-scaling
run of configs[2] is asked for with `--global-batch 256`.

One "step" = one denoise step (Unet3D forward + posterior update) over the local batch, replayed
as one captured HIP graph.  Workload at every N: BASELINE.json configs[1] per GPU — Unet3D(dim=64,
dim_mults=(1,2,4,8), channels=1), x = (16, 1, 10, 64, 64), bf16 kernels / fp32 sampler state,
GaussianDiffusion(timesteps=1000), random-init weights, synthetic noise input.  The batch is
sharded by sample (weak scaling: 16 videos per GPU); the only collective is the all-gather of the
finished videos, timed separately.  With --global-batch G the same model samples G videos in total,
G / N per GPU (strong scaling at a fixed global batch: BASELINE.json configs[2] for G = 256).

After the timed steps every rank runs one real p_sample_loop of all 1000 steps; its wall time is
reported beside the extrapolated figure (full_loop_s, videos_per_s_measured).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak of MI355X
PEAK_HBM_GBS = 8000.0

CFG1 = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=1)
FRAMES, SIZE, TIMESTEPS = 10, 64, 1000


def physical_cores():
    """Physical cores of the host (sockets x cores per socket, from lscpu); None when lscpu is not there."""
    import subprocess
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = dict((k.strip(), v.strip()) for k, v in (ln.split(":", 1) for ln in txt.splitlines() if ":" in ln))
        return int(kv["Socket(s)"]) * int(kv["Core(s) per socket"])
    except Exception:
        return None


def cpu_baseline(batch, steps):
    """The CPU oracle (fp32 restatement of the reference path) on the host cores, bounded sample.

    Threads: one per physical core of the host, but no more than the share of the machine this process is given
    (the affinity mask, and OMP_NUM_THREADS when the launcher sets it): more threads than that only contend."""
    from oracle import vdm_oracle as O
    from video_diffusion_amd import Unet3D
    phys = physical_cores()
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        share = min(share, int(os.environ["OMP_NUM_THREADS"]))
    threads = max(1, min(phys or share, share))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = Unet3D(**CFG1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = O.UnetCfg(dim=64, dim_mults=(1, 2, 4, 8), channels=1)
    bufs = O.diffusion_buffers(TIMESTEPS)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch, 1, FRAMES, SIZE, SIZE, generator=g)
    with torch.no_grad():
        def one(i, x):
            t = torch.full((batch,), TIMESTEPS - 1 - i, dtype=torch.long)
            eps = O.unet3d_forward(sd, cfg, x, t)
            return O.p_sample_step(bufs, x, t, eps, torch.randn(x.shape, generator=g))
        x = one(0, x)                      # warm-up
        t0 = time.perf_counter()
        for i in range(steps):
            x = one(i + 1, x)
        dt = time.perf_counter() - t0
    steps_per_s = steps / dt
    return {"value": batch * steps_per_s / TIMESTEPS, "unit": "videos/s", "cores": threads, "kind": "port",
            "physical_cores_of_host": phys, "cpu_share_of_process": share,
            "sample": f"{steps} denoise steps at batch {batch} of the same model/shape, fp32, torch CPU on {threads} "
                      f"threads ({dt:.1f} s); videos/s = batch*steps/s/{TIMESTEPS}",
            "denoise_steps_per_s": steps_per_s}


def train_step_line(device, steps=5, torch_leg=False):
    """BASELINE configs[4]: loss forward + backward of Unet3D(dim=128, channels=3) on (1,3,16,128,128)
    (reference vdp.py:1146-1161, 1451) on the HIP training kernels, under bf16 autocast (torch_leg: also the same step
    as torch ops, which run the convolutions on MIOpen).  Reported beside the sampling metric, outside its timed region."""
    import torch.nn.functional as F
    from video_diffusion_amd import Unet3D
    torch.manual_seed(0)
    unet = Unet3D(dim=128, dim_mults=(1, 2, 4, 8), channels=3).to(device)
    x = torch.randn(1, 3, 16, 128, 128, device=device)
    noise = torch.randn_like(x)
    t = torch.randint(0, TIMESTEPS, (1,), device=device)

    from video_diffusion_amd import train_hip
    store = train_hip.PackStore()

    def step(backend):
        unet.train_backend = backend
        for p in unet.parameters():
            p.grad = None
        # as in the Trainer's loop: packed weight sets live in a store; every timed step counts as following an optimizer
        # step (every set stale), so it packs each weight of the model once, in the store's one batched launch
        store.invalidate()
        with train_hip.fixed_weights(store):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.l1_loss(unet(x, t).float(), noise)
            loss.backward()
        return loss

    out = {"workload": "BASELINE configs[4]: Unet3D(dim=128,dim_mults=(1,2,4,8),channels=3) l1 loss forward + backward on "
                       "x=(1,3,16,128,128), bf16 autocast, random-init weights, no optimizer step", "steps": steps}
    legs = (("hip_ms", "hip"),) + ((("torch_ops_ms", "torch-autograd"),) if torch_leg else ())
    for key, backend in legs:
        for _ in range(2):
            step(backend)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(backend)
        torch.cuda.synchronize()
        out[key] = round((time.perf_counter() - t0) / steps * 1e3, 2)
    out["samples_per_s"] = round(1e3 / out["hip_ms"], 2)
    # Algorithmic work of the step: the forward of this model on this clip is 3 324.7 GFLOP (counted on the reference
    # with torch's FlopCounterMode), the backward twice that (input and weight gradients).  Against the dense bf16 peak.
    gflop = 3.0 * 3324.7
    tf = gflop / out["hip_ms"]
    out["roofline"] = {"bound": "mfma", "algorithmic_gflop_per_step": round(gflop, 1), "achieved": round(tf, 1),
                       "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4)}
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: start N ranks (one per GPU) through
    torch.distributed.run on a free local port and hand back its exit status.  The parent makes no GPU call, before
    or after; rank 0 of the children prints the JSON line, which passes through on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # RCCL needs dmabuf IPC on this driver
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank):
    """VDMI_BENCH_DRY_RUN=1: the ranks only rendezvous (gloo, CPU) and rank 0 prints what the real run would be
    launched as.  Exercises argument and launcher handling on a machine without a GPU."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        tt = torch.tensor([float(rank)])
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        ok = int(tt.item()) == world * (world - 1) // 2
    else:
        ok = True
    if args.global_batch is not None and args.global_batch % world:
        raise SystemExit(f"--global-batch {args.global_batch} does not divide over {world} ranks")
    batch = args.global_batch // world if args.global_batch is not None else args.batch
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "world_size": world, "ranks_met": ok,
                          "batch_per_gpu": batch, "global_batch": batch * world,
                          "scaling": "strong" if args.global_batch is not None else "weak",
                          "steps": args.steps, "warmup": args.warmup}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="videos per GPU")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="total videos over all GPUs (strong scaling; 256 = BASELINE configs[2]); overrides --batch")
    ap.add_argument("--no-full-loop", action="store_true", help="skip the real 1000-step loop after the timed steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the configs[4] training-step line")
    ap.add_argument("--train-torch-ops", action="store_true", help="also time the training step as torch ops (MIOpen convolutions)")
    ap.add_argument("--parts", type=int, default=None,
                    help="concurrent sub-batches one forward is run as (default: the library's choice)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called plainly (`python bench.py --gpus N`): this process only starts the N ranks and waits for them
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}")
    import torch.distributed as dist
    if os.environ.get("VDMI_BENCH_DRY_RUN") == "1":
        raise SystemExit(dry_run(args, world, rank))
    # one rank per GPU; VDMI_BENCH_BACKEND=gloo lets several ranks rehearse the multi-rank path on one card
    backend = os.environ.get("VDMI_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from video_diffusion_amd import GaussianDiffusion, Unet3D
    from video_diffusion_amd.parallel import gather_videos

    if args.global_batch is not None:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} does not divide over {world} ranks")
        args.batch = args.global_batch // world

    from video_diffusion_amd import engine as vd_engine
    from video_diffusion_amd import _lib as vd_lib      # VDMI_PRECISION=f16 benches the IEEE-half build; default bf16
    if args.parts is not None:
        vd_engine.BATCH_PARTS = args.parts
    parts = vd_engine.batch_parts(args.batch)

    torch.manual_seed(0)
    unet = Unet3D(**CFG1).to(device).eval()
    gd = GaussianDiffusion(unet, torch.nn.Identity(), image_size=SIZE, num_frames=FRAMES, channels=1,
                           timesteps=TIMESTEPS).to(device)
    gd.noise_source = "philox"
    B = args.batch
    shape = (B, 1, FRAMES, SIZE, SIZE)
    torch.manual_seed(1234 + rank)

    graph, plan = gd._graph_runner(shape, device)
    plan.x_in.copy_(torch.randn(shape, device=device))
    gd.reseed_noise(device)                       # Philox key in device memory (ranks are seeded differently above)
    rng = gd._rng_state(device)

    def eager_step():
        plan.run()
        gd.ddpm_step(plan.x_in, plan.eps_out, plan.t_in, noise=None, clip_denoised=True, out=plan.x_in, rng_state=rng)

    plan.t_in.fill_(TIMESTEPS - 1)

    def step(i):
        if args.no_graph:
            plan.t_in.fill_(TIMESTEPS - 1 - (i % TIMESTEPS))
            eager_step()
        else:
            if not getattr(plan, "self_advancing", False):
                plan.t_in.fill_(TIMESTEPS - 1 - (i % TIMESTEPS))
            graph.replay()          # the captured posterior kernel counts t down itself (wrapping after 0)

    for i in range(args.warmup):
        step(i)

    def barrier():
        if world > 1:
            dist.barrier()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(plan.x_in).all(), "sampler state went non-finite"

    # ---- one real loop: x_T drawn, all 1000 steps, fresh Philox key (the product entry point, nothing extrapolated) ----
    full_loop_s = None
    if not args.no_full_loop:
        gd.use_hip_graph = not args.no_graph
        barrier()
        torch.cuda.synchronize()
        f0 = time.perf_counter()
        videos = gd.p_sample_loop(shape)
        torch.cuda.synchronize()
        barrier()
        full_loop_s = time.perf_counter() - f0
        if world > 1:
            tt = torch.tensor([full_loop_s], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            full_loop_s = float(tt.item())
        assert videos.shape == shape and torch.isfinite(videos).all()
        del videos

    # the one collective of the sharded sampler: all-gather of the finished videos
    gather_ms = None
    if world > 1:
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        full = gather_videos(plan.x_in, B * world)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert full.shape[0] == B * world

    # ---- per-kernel timing with HIP events on the launch stream (eager, same kernels, same data) ----
    # Every launch of the plan is timed on its own, back to back on one stream.  (In the timed loop above the forward
    # runs as concurrent sub-batch plans, so launches of the two streams overlap there; a launch's own duration is what
    # a kernel trace reports for it.)
    def measure(pl, reps):
        stream = torch.cuda.current_stream()
        evs = []
        for r in range(reps):
            pl.t_in.fill_(TIMESTEPS - 1 - r)
            s = stream.cuda_stream
            for ci, call in enumerate(pl.calls):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                st = call.fn(s)
                e1.record(stream)
                assert st == 0, call.name
                evs.append((ci, e0, e1))
        torch.cuda.synchronize()
        per_call = {}
        for ci, e0, e1 in evs:
            per_call[ci] = per_call.get(ci, 0.0) + e0.elapsed_time(e1) / reps
        classes = {}
        for ci, ms in per_call.items():
            name, meta = pl.calls[ci].name, pl.calls[ci].args or {}
            key = meta.get("kernel", name.split(".")[-1] if not name.startswith(("time", "block", "init", "final")) else name)
            c = classes.setdefault(key, {"ms_per_step": 0.0, "launches_per_step": 0, "flops_per_step": 0.0,
                                         "bytes_per_step": 0.0})
            c["ms_per_step"] += ms
            c["launches_per_step"] += 1
            c["flops_per_step"] += meta.get("flops", 0.0)
            c["bytes_per_step"] += meta.get("bytes", 0.0)
        return classes

    if rank == 0:
        classes = measure(plan, max(3, min(args.steps, 10)))
        # dominant kernel: the matrix-core kernel class with the largest share of the step.  Two classes of this step lie
        # within a percent of each other (the streaming convolution and the fused temporal attention, ~1.0 ms each), and
        # which of them is ahead changes from device to device: among the classes within 2 % of the longest the one with
        # the most matrix work is named, the other is reported beside it (`runner_up`).
        mfma_classes = [k for k in classes if classes[k]["flops_per_step"] > 0]
        top_ms = max(classes[k]["ms_per_step"] for k in mfma_classes)
        near = [k for k in mfma_classes if classes[k]["ms_per_step"] >= 0.98 * top_ms]
        dom_key = max(near, key=lambda k: classes[k]["flops_per_step"])
        dom = classes[dom_key]
        avg_ms = dom["ms_per_step"] / dom["launches_per_step"]
        achieved = dom["flops_per_step"] / (dom["ms_per_step"] * 1e-3) / 1e12
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dom_key)
                traffic_source = ("profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench, "
                                  "not collected in this run): " + str(tj.get("_source", "source not recorded")))
            except Exception:
                traffic = None
        roofline = {"bound": "mfma", "kernel": dom_key, "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": traffic_source,
                    "avg_launch_ms": round(avg_ms, 4), "launches_per_step": dom["launches_per_step"],
                    "videos_per_launch": B // parts,
                    "algorithmic_gflop_per_launch": round(dom["flops_per_step"] / dom["launches_per_step"] / 1e9, 2)}
        others = sorted((k for k in mfma_classes if k != dom_key), key=lambda k: -classes[k]["ms_per_step"])
        if others:
            ru = classes[others[0]]
            ru_tf = ru["flops_per_step"] / (ru["ms_per_step"] * 1e-3) / 1e12
            roofline["runner_up"] = {"kernel": others[0], "ms_per_step": round(ru["ms_per_step"], 4),
                                     "achieved": round(ru_tf, 2), "frac": round(ru_tf / PEAK_BF16_TFLOPS, 4)}
            roofline["ms_per_step"] = round(dom["ms_per_step"], 4)
        # beside the nominal peak: what an operand-free MFMA loop reaches on this device (tools/peak_probe.hip)
        ppath = os.path.join(ROOT, "profiles", "peak_probe.json")
        if os.path.exists(ppath):
            try:
                measured = max(json.load(open(ppath))["mfma_bf16_tflops"].values())
                roofline["peak_measured"] = measured
                roofline["frac_of_measured"] = round(achieved / measured, 4)
            except Exception:
                pass
        # the whole step on the same scale: all matrix-core work of a step over the step time of the timed loop
        # (the sub-batch streams overlap there, so this can exceed what any one launch reaches on its own)
        step_tflops = sum(c["flops_per_step"] for c in classes.values()) / (elapsed / args.steps) / 1e12
        roofline["step_achieved"] = round(step_tflops, 2)
        roofline["step_frac"] = round(step_tflops / PEAK_BF16_TFLOPS, 4)
        if parts > 1:
            # for reference: the same kernel class launched once over the whole batch (the step as ONE plan)
            eng = unet.engine()
            whole = eng.plan_class(eng, B, FRAMES, SIZE, SIZE, device)
            whole.x_in.copy_(plan.x_in)
            wc = measure(whole, 3)[dom_key]
            wa = wc["flops_per_step"] / (wc["ms_per_step"] * 1e-3) / 1e12
            roofline["whole_batch_launch"] = {"achieved": round(wa, 2), "frac": round(wa / PEAK_BF16_TFLOPS, 4),
                                              "avg_launch_ms": round(wc["ms_per_step"] / wc["launches_per_step"], 4),
                                              "launches_per_step": wc["launches_per_step"]}
            del whole
        # the HBM-bound classes on their own roofline: algorithmic bytes over the launch time, against 8 TB/s
        hbm = {}
        if "rmsnorm_silu" in classes:
            c = classes["rmsnorm_silu"]
            gbs = c["bytes_per_step"] / (c["ms_per_step"] * 1e-3) / 1e9
            hbm["rmsnorm_silu"] = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": round(gbs / PEAK_HBM_GBS, 4), "launches_per_step": c["launches_per_step"],
                                   "avg_launch_ms": round(c["ms_per_step"] / c["launches_per_step"], 4)}
        reps = 20
        scratch_x = plan.x_in.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gd.ddpm_step(scratch_x, plan.eps_out, plan.t_in, noise=None, out=scratch_x, rng_state=rng)
        e0.record()
        for _ in range(reps):
            gd.ddpm_step(scratch_x, plan.eps_out, plan.t_in, noise=None, out=scratch_x, rng_state=rng)
        e1.record()
        torch.cuda.synchronize()
        d_ms = e0.elapsed_time(e1) / reps
        d_bytes = scratch_x.numel() * (2 * scratch_x.element_size() + plan.eps_out.element_size())
        hbm["ddpm_step"] = {"achieved": round(d_bytes / (d_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(d_bytes / (d_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "launches_per_step": 1,
                            "avg_launch_ms": round(d_ms, 4),
                            "algorithmic_bytes_per_launch": d_bytes}
        roofline["hbm_classes"] = hbm
        # traffic_gbs: HBM-side bytes of the class from the committed counter passes (sla_fused = its three kernels)
        # over the launch time measured here: how close a class runs to the memory side of its roofline
        tj = {}
        try:
            tj = json.load(open(tpath))
        except Exception:
            pass
        per_launch = dict(tj)
        if all(k in tj for k in ("sla_pass1", "sla_fold", "sla_pass2")):
            per_launch["sla_fused"] = tj["sla_pass1"] + tj["sla_fold"] + tj["sla_pass2"]
        breakdown = {k: {"ms_per_step": round(v["ms_per_step"], 4), "launches": v["launches_per_step"],
                         "tflops": round(v["flops_per_step"] / (v["ms_per_step"] * 1e-3) / 1e12, 1) if v["flops_per_step"] else None,
                         "alg_gbs": round(v["bytes_per_step"] / (v["ms_per_step"] * 1e-3) / 1e9, 1) if v["bytes_per_step"] else None,
                         "traffic_gbs": round(per_launch[k] * v["launches_per_step"] / (v["ms_per_step"] * 1e-3) / 1e9, 1)
                         if isinstance(per_launch.get(k), (int, float)) else None}
                     for k, v in sorted(classes.items(), key=lambda kv: -kv[1]["ms_per_step"])}

    if rank == 0:
        steps_per_s = args.steps / elapsed
        videos_per_s = B * world * steps_per_s / TIMESTEPS
        out = {
            "metric": "sampled videos/sec + denoise steps/sec, 10x64x64 MovingMNIST, 1/2/4/8 MI355X",
            "value": round(videos_per_s, 4), "unit": "videos/s (T=1000 DDPM steps per video)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if args.global_batch is not None else "weak",
            "vs_baseline": None, "dtype": vd_lib.PRECISION, "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[2]: Unet3D(dim=64,dim_mults=(1,2,4,8),channels=1), global batch "
                                    f"{args.global_batch} sharded over {world} GPU(s) = x=({B},1,10,64,64) per GPU, "
                                    "T=1000, random-init weights") if args.global_batch is not None else
                                   ("BASELINE configs[1]: Unet3D(dim=64,dim_mults=(1,2,4,8),channels=1), "
                                    f"p_sample over x=({B},1,10,64,64) per GPU, T=1000, random-init weights"),
                       "batch_per_gpu": B, "global_batch": B * world, "timesteps": TIMESTEPS,
                       "launch": "eager" if args.no_graph else "hip-graph replay",
                       "concurrency": f"{parts} sub-batch(es) of {B // parts} videos on {parts} HIP stream(s)",
                       "parallelism": f"batch-shard x{world}, all-gather of finished videos"},
            "denoise_steps_per_s": round(steps_per_s, 3),
            "sample_steps_per_s": round(B * world * steps_per_s, 2),
            "model_tflops": round(154.5e9 * B * world * steps_per_s / 1e12, 1),
            "allgather_ms": None if gather_ms is None else round(gather_ms, 3),
            "world_size": world, "collective_backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
            "full_loop_s": None if full_loop_s is None else round(full_loop_s, 3),
            "videos_per_s_measured": None if full_loop_s is None else round(B * world / full_loop_s, 4),
            "roofline": roofline, "breakdown_ms_per_step": breakdown,
        }
        if world == 1 and not args.no_train_step:
            try:
                out["train_step"] = train_step_line(device, torch_leg=args.train_torch_ops)
            except Exception as e:      # a side measurement: report the failure, keep the sampling line
                out["train_step"] = {"failed": f"{type(e).__name__}: {str(e)[:300]}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch=16, steps=3)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
