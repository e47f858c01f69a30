"""GPU parity at the shapes of BASELINE.json's configs that the layer tests do not reach, and of the sampling loop as
the benchmark runs it (HIP-graph replay, in-kernel Philox noise).

Reference arithmetic: video_diffusion_pytorch/video_diffusion_pytorch.py:731-754 (classifier-free guidance branches),
:756-831 (Unet3D.forward), :973-1014 (p_sample / p_sample_loop) of the reference.  Tolerance: BASELINE.json's bf16 bar,
2e-2 relative L2 error per epsilon prediction; the full-loop bound is stated at its test.
"""
import os

import pytest
import torch

from tests.helpers import CFG0, CFG1, CFG3, TINY, TINY_COND, golden, inp, make_unet, record, rel_err

from video_diffusion_amd._lib import PRECISION

pytestmark = pytest.mark.gpu

MODEL_TOL = 2e-2
CFG4 = dict(dim=128, dim_mults=(1, 2, 4, 8), channels=3)


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("B", [32, 64, 128])
def test_shard_sizes_of_global_batch_256(B):
    """configs[2]: global batch 256 over 8 / 4 / 2 GPUs = shards of 32 / 64 / 128 videos.  Samples never interact, so
    the epsilon of a shard equals the epsilon of its 16-video slices bit for bit (the 16-video batch is the one
    pinned to the reference), and the reference's B = 2 golden embedded in the shard holds at 2e-2."""
    m = make_unet(**CFG1).to(dev())
    g = golden("cfg1_eps")
    x = inp(f"cfg1.shard{B}", (B, 1, 10, 64, 64)).to(dev())
    t = (torch.arange(B, device=dev()) * 37) % 1000
    x[5:7] = inp("cfg1.x", (2, 1, 10, 64, 64)).to(dev())
    t[5:7] = g["t"].to(dev())
    with torch.no_grad():
        full = m(x, t)
        for s in range(0, B, 16):
            part = m(x[s:s + 16], t[s:s + 16])
            assert torch.equal(full[s:s + 16], part), (B, s)
    assert record(f"cfg1_eps_inside_shard_of_{B}", rel_err(full[5:7].cpu(), g["eps"]), MODEL_TOL) < MODEL_TOL


def test_cfg3_guidance_branches_vs_reference():
    """configs[3]: the conditional and the null-conditioned evaluation of forward_with_cond_scale, each on its own."""
    m = make_unet(**CFG3).to(dev())
    g = golden("cfg3_branches")
    x, c = inp("cfg3.x", (1, 3, 16, 64, 64)).to(dev()), inp("cfg3.cond", (1, 768)).to(dev())
    with torch.no_grad():
        e_c = m(x, g["t"].to(dev()), cond=c, null_cond_prob=0.)
        e_n = m(x, g["t"].to(dev()), cond=c, null_cond_prob=1.)
    assert record("cfg3_eps_cond", rel_err(e_c.cpu(), g["eps_cond"]), MODEL_TOL) < MODEL_TOL
    assert record("cfg3_eps_null", rel_err(e_n.cpu(), g["eps_null"]), MODEL_TOL) < MODEL_TOL


def test_cfg4_shape_eps_vs_reference():
    """configs[4] shape on the forward path: Unet3D(dim=128, channels=3) on one 16 x 128 x 128 RGB clip."""
    m = make_unet(**CFG4).to(dev())
    g = golden("cfg4_eps")
    x = inp("cfg4.x", (1, 3, 16, 128, 128)).to(dev())
    with torch.no_grad():
        eps = m(x, g["t"].to(dev()))
    assert record("cfg4_eps", rel_err(eps.cpu(), g["eps"].float()), MODEL_TOL) < MODEL_TOL


def test_cfg0_full_loop_T250_vs_reference():
    """configs[0], the whole free-running loop: 250 steps from the reference's x_T with the reference's noise
    draws, every step feeding the next (no teacher forcing).

    What the fixtures say about this loop (random closed-form weights, x_0 clipped to [-1, 1] every step):
      * the fp32 reference with nothing but its epsilon rounded to bf16 ends 7e-4 from itself: the loop is stable;
      * a bf16 implementation's per-step error is mostly a FIXED perturbation (rounded weights, the same rounding
        pattern on nearly the same activations step after step), so it adds up coherently over the 250 steps;
      * the reference itself under torch.autocast(bfloat16) is the yardstick: its states and final sample are in
        cfg0_full_T250_autocast.  Up to 200 steps the HIP loop may not drift further from the fp32 reference than
        1.25 x that (+1e-3); the last 50 steps, where x_0 is no longer clipped, amplify any difference about 4.5 x
        (autocast reference: 0.034 -> 0.120), and last-bit changes in a kernel move the HIP loop's final value
        between 0.13 and 0.15, so the final sample is held to 2 x the yardstick.
    Every measured number goes to the parity log."""
    from video_diffusion_amd import GaussianDiffusion
    m = make_unet(**CFG0).to(dev())
    g, ga = golden("cfg0_full_T250"), golden("cfg0_full_T250_autocast")
    shape = (1, 3, 5, 32, 32)
    gd = GaussianDiffusion(m, torch.nn.Identity(), image_size=32, num_frames=5, channels=3, timesteps=250).to(dev())
    noises = [inp(f"cfg0.full.z{k}", shape) for k in range(250)]
    x_T = inp("cfg0.full.xT", shape)
    out = gd.p_sample_loop(shape, x_T=x_T, noises=noises)
    assert torch.isfinite(out).all()
    # the same loop step by step, to read the state at the steps the fixture holds
    marks = [int(k) for k in g["marks"]]
    x = x_T.to(dev())
    for k, i in enumerate(reversed(range(250))):
        if k in marks:
            j = marks.index(k)
            e_hip = rel_err(x.cpu(), g["states"][j].float())
            e_auto = rel_err(ga["states"][j].float(), g["states"][j].float())
            record(f"cfg0_full_loop_state_after_{k}_steps", e_hip, 1.25 * e_auto + 1e-3)
            record(f"cfg0_full_loop_state_after_{k}_steps_reference_autocast_bf16", e_auto, None)
            assert e_hip < 1.25 * e_auto + 1e-3, (k, e_hip, e_auto)
        x = gd.p_sample(x, torch.full((1,), i, device=dev(), dtype=torch.long), noise=noises[k].to(dev()))
    assert torch.equal(x, out)                                   # the loop entry point is these 250 steps
    e_auto = record("cfg0_full_loop_T250_final_sample_reference_autocast_bf16", rel_err(ga["final"], g["final"]), None)
    e_round = record("cfg0_full_loop_T250_final_sample_reference_eps_rounded_to_bf16",
                     rel_err(g["final_bf16eps"], g["final"]), None)
    # the IEEE-half build (VDMI_PRECISION=f16) is held to the stated tolerance itself: 2e-2 on the final sample
    bar = 2e-2 if PRECISION == "f16" else 2.0 * e_auto
    assert e_round < 2e-3
    assert record("cfg0_full_loop_T250_final_sample", rel_err(out.cpu(), g["final"]), bar) < bar


def _tiny_gd(T=8):
    from video_diffusion_amd import GaussianDiffusion
    m = make_unet(**TINY).to(dev())
    gd = GaussianDiffusion(m, torch.nn.Identity(), image_size=12, num_frames=4, channels=3, timesteps=T).to(dev())
    gd.noise_source = "philox"
    return gd


def test_graph_replay_equals_eager_loop_with_philox():
    """The benchmarked path (HIP-graph replay, timestep tables, sub-batch streams, in-kernel Philox) gives the same
    samples, bit for bit, as the eager loop of separate launches keyed with the same seed."""
    gd = _tiny_gd()
    shape = (6, 3, 4, 12, 12)
    x_T = inp("graph.xT", shape)
    torch.manual_seed(77)
    gd.use_hip_graph = True
    a = gd.p_sample_loop(shape, x_T=x_T)
    assert gd._graphs, "the loop did not go through a captured graph"
    torch.manual_seed(77)
    gd.use_hip_graph = False
    b = gd.p_sample_loop(shape, x_T=x_T)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    # the parts of the batch as independent loops (one graph and stream per part, no join between steps, the second
    # started later): the same noise field, the same bits
    torch.manual_seed(77)
    gd.use_hip_graph, gd.pipeline_parts, gd.pipeline_stagger_s = True, True, 2e-3
    c = gd.p_sample_loop(shape, x_T=x_T)
    assert any(k[-1] is True for k in gd._graphs if k[0] != "cond")
    assert torch.equal(a, c)


def test_dynamic_threshold_loop_graph_equals_eager_and_torch_expressions():
    """use_dynamic_thres=True: the captured loop (selection kernel + thresholded step inside the graph) equals the eager
    loop bit for bit, and, with injected noise, the loop whose posterior is the reference's torch expressions
    (torch.quantile, clamp, divide: vdp.py:957-968) on the same epsilons to float32 rounding."""
    gd = _tiny_gd()
    gd.use_dynamic_thres, gd.dynamic_thres_percentile = True, 0.9
    shape = (2, 3, 4, 12, 12)
    x_T = inp("dynloop.xT", shape) * 1.5
    torch.manual_seed(9)
    gd.use_hip_graph = True
    a = gd.p_sample_loop(shape, x_T=x_T)
    assert gd._graphs, "the loop did not go through a captured graph"
    torch.manual_seed(9)
    gd.use_hip_graph = False
    b = gd.p_sample_loop(shape, x_T=x_T)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    # injected noise: fused step against p_mean_variance's torch expressions, step by step on the same state
    T = gd.num_timesteps
    img = x_T.to(a.device)
    worst = 0.0
    for k, i in enumerate(reversed(range(T - 3, T))):
        t = torch.full((shape[0],), i, device=a.device, dtype=torch.long)
        z = inp(f"dynloop.z{k}", shape).to(a.device)
        with torch.no_grad():
            mean, _, logvar = gd.p_mean_variance(img, t, clip_denoised=True)
            want = mean + (0.5 * logvar).exp() * z
        got = gd.p_sample(img, t, noise=z)
        worst = max(worst, float((got - want).abs().max() / want.abs().max()))
        img = got
    assert record("dyn_thres_loop_vs_torch", worst, 1e-5) < 1e-5


def test_philox_noise_differs_between_calls():
    """Two loops of the same shape on one object (the second replays the cached graph) draw different noise; the same
    torch seed reproduces a loop."""
    gd = _tiny_gd()
    shape = (2, 3, 4, 12, 12)
    x_T = inp("philox.xT", shape)
    torch.manual_seed(5)
    a = gd.p_sample_loop(shape, x_T=x_T)
    b = gd.p_sample_loop(shape, x_T=x_T)
    torch.manual_seed(5)
    c = gd.p_sample_loop(shape, x_T=x_T)
    assert not torch.equal(a, b)
    assert float((a - b).abs().mean()) > 1e-2          # a different noise field, not a rounding difference
    assert torch.equal(a, c)


def test_weights_rewritten_through_data_are_repacked():
    """In-place writes through `.data` (the reference's EMA, `p.data.copy_`) change neither the pointer nor autograd's
    version counter.  The sampling loop notices them through its per-loop value fingerprint, a bare forward after
    Unet3D.weights_changed()."""
    gd = _tiny_gd()
    m = gd.denoise_fn
    shape = (2, 3, 4, 12, 12)
    x_T = inp("stale.xT", shape)
    torch.manual_seed(3)
    a = gd.p_sample_loop(shape, x_T=x_T)
    x, t = inp("stale.x", shape).to(dev()), torch.tensor([3, 5], device=dev())
    saved = [p.detach().clone() for p in m.parameters()]
    with torch.no_grad():
        e0 = m(x, t)
        for p in m.parameters():
            p.data.mul_(1.25)
    torch.manual_seed(3)
    b = gd.p_sample_loop(shape, x_T=x_T)                       # same seed, same x_T: only the weights differ
    assert not torch.equal(a, b)
    with torch.no_grad():
        for p, q in zip(m.parameters(), saved):
            p.data.copy_(q)
        e_stale = m(x, t)                                       # a bare forward does not look at values ...
        m.weights_changed()
        e1 = m(x, t)                                            # ... until it is told
    assert not torch.equal(e_stale, e0) and torch.equal(e1, e0)
    torch.manual_seed(3)
    c = gd.p_sample_loop(shape, x_T=x_T)
    assert torch.equal(c, a)


def test_cond_sample_loop_is_the_two_pass_guided_loop():
    """configs[3] harness: the guided loop over a doubled batch (rows B.. take the null embedding) equals the loop
    that evaluates the conditional and the null-conditioned epsilon in two passes (forward_with_cond_scale of the
    reference, vdp.py:731-748, then the posterior step :973-1002), and its graph replay equals its eager run."""
    from video_diffusion_amd import GaussianDiffusion
    m = make_unet(**TINY_COND).to(dev())
    T = 6
    gd = GaussianDiffusion(m, torch.nn.Identity(), image_size=12, num_frames=4, channels=3, timesteps=T).to(dev())
    gd.noise_source = "philox"
    shape = (2, 3, 4, 12, 12)
    x_T, cond = inp("cl.xT", shape).to(dev()), inp("cl.cond", (2, 16)).to(dev())
    noises = [inp(f"cl.z{k}", shape).to(dev()) for k in range(T)]
    a = gd.cond_sample_loop(shape, cond, 2.0, x_T=x_T, noises=noises)
    x = x_T.clone()
    with torch.no_grad():
        for k, i in enumerate(reversed(range(T))):
            t = torch.full((2,), i, device=dev(), dtype=torch.long)
            e_c = m(x, t, cond=cond, null_cond_prob=0.)
            e_n = m(x, t, cond=cond, null_cond_prob=1.)
            x = gd.ddpm_step(x, e_n + (e_c - e_n) * 2.0, t, noise=noises[k])
            g = m.forward_with_cond_scale(x, t, cond=cond, cond_scale=2.0)       # one doubled pass
            e_c2, e_n2 = m(x, t, cond=cond, null_cond_prob=0.), m(x, t, cond=cond, null_cond_prob=1.)
            assert rel_err(g.cpu(), (e_n2 + (e_c2 - e_n2) * 2.0).cpu()) < 1e-6
    assert torch.isfinite(a).all() and rel_err(a.cpu(), x.cpu()) < 1e-5
    assert torch.equal(a, gd.p_sample_loop(shape, x_T=x_T, noises=noises, cond=cond, cond_scale=2.0))
    outs = []
    for use_graph in (True, False):
        torch.manual_seed(9)
        gd.use_hip_graph = use_graph
        outs.append(gd.cond_sample_loop(shape, cond, 2.0, x_T=x_T))
    assert torch.equal(outs[0], outs[1])
    u = gd.cond_sample_loop(shape, cond, 1.0, x_T=x_T, noises=noises)                # unguided: one pass per step
    assert not torch.equal(u, a)


def test_sla_pass1_beside_fold_kernels_on_another_stream():
    """The fused linear attention's pass 1 on one stream while the fold / context kernels of another plan run on a
    second stream: its workspace (partial sums, folded weights) must equal the serial run bit for bit.  (An earlier
    version of the fold kernels, with three MFMAs chained into one accumulator, made this fail.)"""
    from video_diffusion_amd.engine import Plan
    m = make_unet(**CFG1).to(dev())
    eng = m.engine()
    eng.ensure_packed(dev())
    p0, p1 = Plan(eng, 4, 10, 64, 64, dev()), Plan(eng, 4, 10, 64, 64, dev())
    main = torch.cuda.current_stream()
    for p in (p0, p1):
        p.x_in.copy_(inp("conc.x", tuple(p.x_in.shape)).to(dev()))
        p.t_in.fill_(500)
        for c in p.calls:
            assert c.fn(main.cuda_stream) == 0
    torch.cuda.synchronize()
    name = "ups.3.2.fused"
    i0 = [i for i, c in enumerate(p0.calls) if c.name == name][0]
    ws = [k for k in p0.keep if isinstance(k, torch.Tensor) and k.dtype == torch.uint8]
    p0.calls[i0].fn(main.cuda_stream)
    torch.cuda.synchronize()
    ref = [w.clone() for w in ws]
    others = [c for c in p1.calls if c.name.endswith(".fused") or c.name.endswith(".ctx")]
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    for it in range(10):
        s0.wait_stream(main)
        s1.wait_stream(main)
        for k, c in enumerate(others * 2):
            c.fn(s1.cuda_stream)
            if k == 0:
                p0.calls[i0].fn(s0.cuda_stream)
        main.wait_stream(s0)
        main.wait_stream(s1)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ws, ref)), it


def test_cfg4_loss_and_gradients_vs_reference():
    """configs[4] at full size on the HIP training kernels: one L1-loss forward + backward of Unet3D(dim=128,
    channels=3) on a (1, 3, 16, 128, 128) clip (the denoiser call and loss of p_losses, reference vdp.py:1146-1161, and
    its backward, :1451) against the reference's own fp32 loss and gradients (tests/golden/cfg4_grads.npz, made by
    oracle/make_golden_cfg4_grads.py from the imported reference).  Compared: the loss, the L2 norm of every
    parameter's gradient, and the leading 256 entries of the gradient of a few layers of every kind; the set of
    parameters the step does not reach must be the reference's (the cross-attention layers)."""
    import torch.nn.functional as F
    if PRECISION != "bf16":
        pytest.skip("the training kernels are built for the bf16 library")
    g = golden("cfg4_grads")
    names = open(os.path.join(os.path.dirname(__file__), "golden", "cfg4_grads_names.txt")).read().split()
    m = make_unet(**CFG4).to(dev())
    m.train()
    m.train_backend = "hip"
    x = inp("cfg4.x", (1, 3, 16, 128, 128)).to(dev())
    target = inp("cfg4.noise", (1, 3, 16, 128, 128)).to(dev())
    loss = F.l1_loss(m(x, g["t"].to(dev())).float(), target)
    loss.backward()
    e_loss = abs(float(loss.detach()) - float(g["loss"][0])) / float(g["loss"][0])
    assert record("cfg4_train_loss", e_loss, 1e-2) < 1e-2
    params = dict(m.named_parameters())
    ref_norms = dict(zip(names, g["grad_norms"].tolist()))
    worst, worst_name = 0.0, None
    for n, rn in ref_norms.items():
        gr = params[n].grad
        if rn < 0:
            assert gr is None or float(gr.abs().max()) == 0.0, n          # the reference's step does not reach it either
            continue
        assert gr is not None, n
        e = abs(float(gr.double().norm()) - rn) / max(rn, 1e-12)
        if e > worst:
            worst, worst_name = e, n
    record("cfg4_train_grad_norm_worst_parameter", worst, 2e-2)
    assert worst < 2e-2, (worst_name, worst)
    errs = []
    for k, v in g.items():
        if k.startswith("slice."):
            got = params[k[len("slice."):]].grad.reshape(-1)[:256].float().cpu()
            errs.append((rel_err(got, v.float()), k))
    record("cfg4_train_grad_slices_worst", max(errs)[0], 1.5e-1)
    record("cfg4_train_grad_slices_median", sorted(errs)[len(errs) // 2][0], 3e-2)
    assert max(errs)[0] < 1.5e-1, max(errs)
    assert sorted(errs)[len(errs) // 2][0] < 3e-2
