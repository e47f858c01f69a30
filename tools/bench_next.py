#!/usr/bin/env python3
"""Timing of the paths next to the denoising loop: frame-conditioned (guided) denoise step, VAE decode / encode,
frame encoder.  One JSON line per workload; per-kernel-class times come from HIP events on the launch stream.

    python tools/bench_next.py [--batch 16] [--frames 10] [--latent 64] [--prev 4] [--steps 10]

Workloads (random-init weights, synthetic inputs):
  framecond  Unet3D(dim=64, dim_mults=(1,2,4,8), channels=4) on latents (B,4,F,S,S), conditioning on `prev`
             previous frames (context = prev * (S/4)^2 tokens of 512 features), frame_cond_scale=1.5: one step =
             one pass over the doubled batch + guidance + posterior step, replayed as a HIP graph.
  vae        VAE(channels=3, latent_dim=4, base_dim=64): decode (B,4,F,S,S) -> (B,3,F,8S,8S), encode back.
  textcfg    Unet3D(dim=128, dim_mults=(1,2,4,8), channels=3, cond_dim=768) on (--cfg-batch, 3, 16, 64, 64) with
             classifier-free guidance cond_scale=2: one step = one pass over the doubled batch (rows B.. take the
             null embedding) + guidance + posterior step, replayed as a HIP graph; with its roofline block.
  guidance   (on request) version1.Unet3D(dim=64, channels=4) on latents (--guid-batch, 4, F, --guid-latent, ..): the
             reconstruction-guidance gradient and the guided step, denoiser gradient on the HIP kernels vs torch ops.
  train      Unet3D(dim=128, dim_mults=(1,2,4,8), channels=3) on (--train-batch, 3, 16, 128, 128): one p_losses
             forward + backward (L1 loss on the predicted noise), with the all-torch backend ("torch-autograd", fp32 and
             under bf16 autocast) and with the ResnetBlocks forward + backward on the HIP kernels ("hip-resnet").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def per_call(calls, reps=5):
    """calls: iterable of (fn(stream), name).  Returns {name: ms per run}."""
    stream = torch.cuda.current_stream()
    evs = []
    for _ in range(reps):
        for fn, name in calls:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            assert fn(stream.cuda_stream) == 0, name
            e1.record(stream)
            evs.append((name, e0, e1))
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1 in evs:
        out[name] = out.get(name, 0.0) + e0.elapsed_time(e1) / reps
    return out


def text_cfg(args, dev):
    """Text-conditioned model with classifier-free guidance (the guided forward of reference vdp.py:731-748 inside
    the posterior-step loop), one JSON line with the roofline of its dominant matrix-core kernel class."""
    from video_diffusion_amd import GaussianDiffusion, Unet3D
    B, F, S = args.cfg_batch, 16, 64
    torch.manual_seed(0)
    unet = Unet3D(dim=128, dim_mults=(1, 2, 4, 8), channels=3, cond_dim=768).to(dev).eval()
    gd = GaussianDiffusion(unet, torch.nn.Identity(), image_size=S, num_frames=F, channels=3, timesteps=1000).to(dev)
    gd.noise_source = "philox"
    shape = (B, 3, F, S, S)
    cond = torch.randn(B, 768, device=dev)
    t0 = time.perf_counter()
    gd.cond_sample_loop(shape, cond, 2.0, steps=3)                  # builds plan + graph
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gd.cond_sample_loop(shape, cond, 2.0, steps=3)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    gd.cond_sample_loop(shape, cond, 2.0, steps=3 + args.steps)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    ms = ((t3 - t2) - (t2 - t1)) / args.steps * 1e3
    plan = unet.engine().plan(2 * B, F, S, S, dev)
    stream = torch.cuda.current_stream()
    evs, reps = [], 3
    for _ in range(reps):
        for c in plan.calls:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            assert c.fn(stream.cuda_stream) == 0
            e1.record(stream)
            evs.append((c, e0, e1))
    torch.cuda.synchronize()
    cl = {}
    for c, e0, e1 in evs:
        meta = c.args or {}
        d = cl.setdefault(meta.get("kernel", c.name.split(".")[-1]), {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
        d["ms"] += e0.elapsed_time(e1) / reps
        d["flops"] += meta.get("flops", 0.0) / reps
        d["bytes"] += meta.get("bytes", 0.0) / reps
        d["n"] += 1
    dom_key = max((k for k in cl if cl[k]["flops"] > 0), key=lambda k: cl[k]["ms"])
    dom = cl[dom_key]
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    total_flops = sum(v["flops"] for v in cl.values())
    print(json.dumps({
        "workload": f"text-conditioned CFG x2.0 (doubled batch): Unet3D(dim=128, cond_dim=768) on {shape}, one guided denoise step",
        "ms_per_step": round(ms, 3), "guided_samples_per_s_at_T1000": round(B / (ms * 1e-3) / 1000, 4),
        "first_call_s": round(t1 - t0, 2), "launch": "hip-graph replay", "dtype": "bf16",
        "step_tflops": round(total_flops / (ms * 1e-3) / 1e12, 1),
        "roofline": {"bound": "mfma", "kernel": dom_key, "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": round(ach / 2500.0, 4), "launches_per_step": dom["n"] // reps,
                     "avg_launch_ms": round(dom["ms"] / (dom["n"] // reps), 4), "traffic": None},
        "classes_ms": {k: {"ms": round(v["ms"], 4), "launches": v["n"] // reps,
                           "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None}
                       for k, v in sorted(cl.items(), key=lambda kv: -kv[1]["ms"])},
    }), flush=True)


def train_step(args, dev):
    """configs[4] shape: loss forward + backward of the denoiser (reference vdp.py:1146-1161, 1451), per backend."""
    import torch.nn.functional as F
    from video_diffusion_amd import Unet3D
    torch.manual_seed(0)
    unet = Unet3D(dim=args.train_dim, dim_mults=(1, 2, 4, 8), channels=args.train_channels).to(dev)
    B = args.train_batch
    x = torch.randn(B, args.train_channels, args.train_frames, args.train_size, args.train_size, device=dev)
    noise = torch.randn_like(x)
    t = torch.randint(0, 1000, (B,), device=dev)
    # --train-context L: previous-frame conditioning features (B, L, 512), i.e. the eight CrossAttention layers in the step
    feats = torch.randn(B, args.train_context, 512, device=dev, requires_grad=True) if args.train_context else None

    from video_diffusion_amd import train_hip
    store = train_hip.PackStore()

    def step(backend, amp):
        unet.train_backend = backend
        for p in unet.parameters():
            p.grad = None
        if feats is not None:
            feats.grad = None
        # as in the Trainer: the packed weight sets live in a store; every step counts as following an optimizer step
        # (all sets stale), so each step packs every weight once, in the store's one batched launch
        store.invalidate()
        with train_hip.fixed_weights(store):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                eps = unet(x, t) if feats is None else unet(x, t, frame_cond_features=feats)
                loss = F.l1_loss(eps.float(), noise)
            loss.backward()
        return loss

    out = {"workload": f"training step: Unet3D(dim={args.train_dim}, channels={args.train_channels}) p_losses forward + backward on {tuple(x.shape)}, "
                       "random-init weights, no optimizer step" + (f", frame conditioning on {args.train_context} context tokens" if args.train_context else ""),
           "ms_per_step": {}, "loss": {}, "peak_mem_gib": {}}
    runs = (("hip + bf16 autocast of the torch layers", "hip", True), ("hip", "hip", False),
            ("hip-resnet", "hip-resnet", False), ("hip-resnet + bf16 autocast of the torch layers", "hip-resnet", True),
            ("torch-autograd bf16 autocast", "torch-autograd", True),
            ("torch-autograd fp32", "torch-autograd", False))
    if args.train_backends:
        runs = tuple(r for r in runs if r[0].split()[0] in args.train_backends.split(","))
    if args.train_amp_only:
        runs = tuple(r for r in runs if r[2])
    opt = torch.optim.Adam(unet.parameters(), lr=1e-5, fused=True)

    def full_step(backend, amp):
        # what a Trainer iteration runs: forward + backward, then the (fused) optimizer step; the next forward then finds
        # every packed weight set stale (step() invalidates the store itself, as the Trainer does after opt.step())
        step(backend, amp)
        opt.step()

    for name, backend, amp in runs:
        torch.cuda.reset_peak_memory_stats()
        try:
            ms = timed(lambda: step(backend, amp), args.train_steps, warmup=1)
            out["ms_per_step"][name] = round(ms, 2)
            if args.train_with_adam:
                out.setdefault("ms_per_step_with_fused_adam", {})[name] = round(
                    timed(lambda: full_step(backend, amp), args.train_steps, warmup=1), 2)
            out["loss"][name] = round(float(step(backend, amp)), 5)
            out["peak_mem_gib"][name] = round(torch.cuda.max_memory_allocated() / 2**30, 2)
        except Exception as e:     # report, do not hide: a backend that cannot run this shape says why
            out["ms_per_step"][name] = f"failed: {type(e).__name__}: {str(e)[:200]}"
    unet.train_backend = None
    if args.train_profile:
        # kernel time of ONE more hip-resnet step (after warm-up), by kernel name, from the torch profiler
        from torch.profiler import ProfilerActivity, profile
        amp = args.train_profile == "amp"
        step(args.train_profile_backend, amp)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step(args.train_profile_backend, amp)
            torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:40]
        out["hip_resnet_step_kernels_ms"] = {e.key[:90]: [round(e.device_time_total / 1e3, 3), e.count] for e in rows}
    print(json.dumps(out), flush=True)


def guidance_step(args, dev):
    """One guided sampling step of the version1 variant (version1diffusion.py:572-592, 707-747): the reconstruction-guidance
    gradient (forward + input gradient through the denoiser and the VAE decoder) and the denoising step after it, with the
    denoiser's gradient on the HIP training kernels and on torch ops."""
    from video_diffusion_amd import VAE, train_hip, version1 as V1
    torch.manual_seed(0)
    B, F, S, ov = args.guid_batch, args.frames, args.guid_latent, 2
    if args.guid_variant:
        # the video_diffusion_guidance.py denoiser (TemporalSpatialBlocks, GroupNorm, per-frame spatial attention)
        from video_diffusion_amd import guidance_variant as GV, train_hip_variant
        train_hip = train_hip_variant           # its own shape test decides between the HIP layers and torch ops
        unet = GV.Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    else:
        unet = V1.Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    vae = VAE(channels=3, latent_dim=4, base_dim=64)
    gd = V1.GaussianDiffusion(unet.eval(), vae.eval(), image_size=8 * S, num_frames=F, channels=3, timesteps=1000).to(dev)
    z = torch.randn(B, 4, F, S, S, device=dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.long)
    prev = torch.rand(B, 3, ov, 8 * S, 8 * S, device=dev)
    fn = V1.temporal_consistency_guidance(gd, prev, ov, w_r=1.0)
    out = {"workload": f"guided sampling step: {'guidance_variant' if args.guid_variant else 'version1'}.Unet3D(dim=64, channels=4) on latents {tuple(z.shape)}, reconstruction "
                       f"guidance over {ov} overlap frames decoded to {8 * S}x{8 * S}, random-init weights",
           "supports_hip": bool(train_hip.supports(gd.denoise_fn)), "ms": {}}
    orig = train_hip.supports
    for name, forced in (("guidance gradient, denoiser on the HIP training kernels", None),
                         ("guidance gradient, denoiser as torch ops", lambda m: False)):
        train_hip.supports = forced or orig
        try:
            out["ms"][name] = round(timed(lambda: fn(z, t), args.steps, warmup=2), 2)
            out["ms"][name.replace("guidance gradient", "guided p_sample")] = round(
                timed(lambda: gd.p_sample(z, t, guidance_fn=fn, guidance_scale=1.0), args.steps, warmup=2), 2)
        finally:
            train_hip.supports = orig
    with torch.no_grad():
        out["ms"]["p_sample without guidance"] = round(timed(lambda: gd.p_sample(z, t), args.steps, warmup=2), 2)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--guid-variant", action="store_true", help="time the video_diffusion_guidance.py denoiser instead of version1's")
    ap.add_argument("--guid-batch", type=int, default=2)
    ap.add_argument("--guid-latent", type=int, default=32)
    ap.add_argument("--train-batch", type=int, default=1)
    ap.add_argument("--train-size", type=int, default=128)
    ap.add_argument("--train-dim", type=int, default=128)
    ap.add_argument("--train-frames", type=int, default=16)
    ap.add_argument("--train-amp-only", action="store_true", help="skip the runs whose torch layers compute in fp32")
    ap.add_argument("--train-with-adam", action="store_true", help="also time the step followed by a fused Adam update")
    ap.add_argument("--train-context", type=int, default=0, help="context tokens of previous-frame conditioning (0: none)")
    ap.add_argument("--train-channels", type=int, default=3)
    ap.add_argument("--train-steps", type=int, default=3)
    ap.add_argument("--train-profile-backend", default="hip")
    ap.add_argument("--train-profile", default="", help="fp32 | amp: add per-kernel times of one hip-resnet step")
    ap.add_argument("--train-backends", default="", help="comma list out of hip-resnet,torch-autograd (default: all)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--prev", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cfg-batch", type=int, default=8)
    ap.add_argument("--only", default="", help="comma list of workloads to run (default: all)")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    want = lambda name: not only or name in only
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from video_diffusion_amd import GaussianDiffusion, Unet3D, VAE
    from video_diffusion_amd.vae import FrameConditioningEncoder
    B, F, S = args.batch, args.frames, args.latent
    torch.manual_seed(0)

    if want("textcfg"):
        text_cfg(args, dev)
    if "train" in only:        # only on request: the all-torch backends take seconds per step
        train_step(args, dev)
    if "guidance" in only:
        guidance_step(args, dev)
    if not (want("framecond") or want("vae")):
        return

    # ---- frame-conditioned guided step ------------------------------------------------------------------
    unet = Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=4).to(dev).eval()
    enc = FrameConditioningEncoder(4, 512, args.prev).to(dev).eval()
    with torch.no_grad():
        feats = enc(torch.randn(B, 4, args.prev, S, S, device=dev))
    L = feats.shape[1]
    gd = GaussianDiffusion(unet, torch.nn.Identity(), image_size=8 * S, num_frames=F, channels=3, timesteps=1000).to(dev)
    gd.noise_source = "philox"
    shape = (B, 4, F, S, S)
    for guided, scale in ((True, 1.5), (False, 1.0)):
        t0 = time.perf_counter()
        gd.frame_cond_sample_loop(shape, feats, scale, steps=3)          # builds plan + graph
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gd.frame_cond_sample_loop(shape, feats, scale, steps=3)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        gd.frame_cond_sample_loop(shape, feats, scale, steps=3 + args.steps)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ms = ((t3 - t2) - (t2 - t1)) / args.steps * 1e3               # marginal cost of a step (set-up cancels)
        rep = 2 if guided else 1
        plan = unet.engine().plan(rep * B, F, S, S, dev, L)
        # per kernel class (names repeat per layer)
        pc2 = {}
        stream = torch.cuda.current_stream()
        evs = []
        for _ in range(5):
            for c in plan.calls:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                assert c.fn(stream.cuda_stream) == 0
                e1.record(stream)
                evs.append((c, e0, e1))
        torch.cuda.synchronize()
        for c, e0, e1 in evs:
            k = (c.args or {}).get("kernel", c.name.split(".")[-1])
            d = pc2.setdefault(k, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
            d["ms"] += e0.elapsed_time(e1) / 5
            d["flops"] += (c.args or {}).get("flops", 0.0) / 5
            d["bytes"] += (c.args or {}).get("bytes", 0.0) / 5
            d["n"] += 1
        ctx = per_call([(c.fn, c.name) for c in plan.ctx_calls])
        xa = pc2.get("xattn_core", {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 5})
        print(json.dumps({
            "workload": f"framecond {'guided x1.5 (doubled batch)' if guided else 'scale 1'}: latents {shape}, context {L} tokens",
            "ms_per_step": round(ms, 3), "first_call_s": round(t1 - t0, 2),
            "context_kv_ms_once_per_loop": {k: round(v, 4) for k, v in ctx.items()},
            "xattn_core": {"ms_per_step": round(xa["ms"], 4), "launches": xa["n"] // 5,
                           "tflops": round(xa["flops"] / (xa["ms"] * 1e-3) / 1e12, 1) if xa["ms"] else None,
                           "alg_gbs": round(xa["bytes"] / (xa["ms"] * 1e-3) / 1e9, 1) if xa["ms"] else None},
            "classes_ms": {k: round(v["ms"], 4) for k, v in sorted(pc2.items(), key=lambda kv: -kv[1]["ms"])},
        }), flush=True)

    # ---- VAE ---------------------------------------------------------------------------------------------
    vae = VAE(channels=3, latent_dim=4, base_dim=64).to(dev).eval()
    z = torch.randn(B, 4, F, S, S, device=dev)
    with torch.no_grad():
        x = vae.decode(z)
        dec_ms = timed(lambda: vae.decode(z), args.steps)
        enc_ms = timed(lambda: vae.encode(x), args.steps)
        fe_ms = timed(lambda: enc(torch.randn(B, 4, args.prev, S, S, device=dev)), args.steps)
    dplan = next(iter(vae.decoder._hip_engine._plans.values()))
    eplan = next(iter(vae.encoder._hip_engine._plans.values()))
    out_bytes = x.numel() * 4
    print(json.dumps({
        "workload": f"vae: decode {tuple(z.shape)} -> {tuple(x.shape)}, encode back; frame encoder on {args.prev} frames",
        "decode_ms": round(dec_ms, 3), "encode_ms": round(enc_ms, 3), "frame_encoder_ms": round(fe_ms, 3),
        "decode_output_gbs": round(out_bytes / (dec_ms * 1e-3) / 1e9, 1),
        "decode_calls_ms": {k: round(v, 4) for k, v in per_call(dplan.calls).items()},
        "encode_calls_ms": {k: round(v, 4) for k, v in per_call(eplan.calls).items()},
    }), flush=True)


if __name__ == "__main__":
    main()
