"""CPU tests of the training path: the differentiable torch forward (autograd.py) against the reference's golden
outputs, the Trainer loop (GIF dataset, EMA, checkpoint format), and data-parallel training over gloo (world 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from PIL import Image

from oracle import weightgen
from tests.helpers import TINY, TINY_COND, golden, inp, make_unet, rel_err


def test_autograd_forward_matches_reference_golden():
    from video_diffusion_amd import autograd as A, version1 as V1
    g = golden("tiny_framecond")
    m = make_unet(**TINY)
    x = inp("fc.x", tuple(int(v) for v in g["x_shape"]))
    with torch.no_grad():
        assert rel_err(A.unet_forward(m, x, g["t"]), g["eps_plain"]) < 1e-5
        assert rel_err(A.unet_forward(m, x, g["t"], frame_cond_features=g["features"]), g["eps_cond"]) < 1e-5
        assert rel_err(A.unet_forward(m, x, g["t"], frame_cond_features=g["features"], null_frame_cond_prob=1.),
                       g["eps_null"]) < 1e-5
        assert rel_err(A.unet_forward(m, x, g["t"], focus_present_mask=g["focus_mixed"]), g["eps_focus_mixed"]) < 1e-5
        assert rel_err(A.unet_forward(m, x, g["t"], prob_focus_present=1.), g["eps_focus_all"]) < 1e-5
    gc = golden("tiny_cond")
    mc = make_unet(salt=1, **TINY_COND)
    xc = inp("tiny.x", (2, 3, 4, 12, 12))
    with torch.no_grad():
        assert rel_err(A.unet_forward(mc, xc, gc["t"], cond=gc["cond"]), gc["eps_cond"]) < 1e-5
        assert rel_err(A.unet_forward(mc, xc, gc["t"], cond=gc["cond"], null_cond_prob=1.), gc["eps_null"]) < 1e-5
    g1 = golden("v1_tiny")
    m1 = V1.Unet3D(dim=32, dim_mults=(1, 2), channels=3)
    m1.load_state_dict(weightgen.fill_state_dict(m1.state_dict(), 3))
    with torch.no_grad():
        assert rel_err(A.unet_forward(m1, inp("v1.x", tuple(int(v) for v in g1["x_shape"])), g1["t"]), g1["eps"]) < 1e-5


def test_forward_needs_opt_in_for_gradients():
    m = make_unet(**TINY)
    x = inp("fc.x", (1, 3, 2, 8, 8)).requires_grad_(True)
    with pytest.raises(NotImplementedError):
        m(x, torch.tensor([3]))
    m.train_backend = "torch-autograd"
    out = m(x, torch.tensor([3]))
    out.square().mean().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    assert m.init_conv.weight.grad is not None


def _guidance_setup(device):
    from video_diffusion_amd import VAE, version1 as V1
    g = golden("v1_tiny")
    m4 = V1.Unet3D(dim=32, dim_mults=(1, 2), channels=4)
    m4.load_state_dict(weightgen.fill_state_dict(m4.state_dict(), 3))
    vae = VAE(channels=3, latent_dim=4, base_dim=64)
    vae.load_state_dict(weightgen.fill_state_dict(vae.state_dict(), 7))
    gd = V1.GaussianDiffusion(m4.eval(), vae.eval(), image_size=32, num_frames=3, channels=3, timesteps=250).to(device)
    z = inp("v1.zt", tuple(int(v) for v in g["g_shape"])).to(device)
    prev = inp("v1.prevpix", (2, 3, 2, 32, 32)).mul(0.25).add(0.5).clamp(0, 1).to(device)
    return V1, g, gd, z, prev


def test_reconstruction_guidance_matches_reference_golden():
    """temporal_consistency_guidance (version1diffusion.py:707-747) against the gradient the reference computes."""
    V1, g, gd, z, prev = _guidance_setup("cpu")
    out = V1.temporal_consistency_guidance(gd, prev, 2, w_r=3.0)(z, g["g_t"])
    assert rel_err(out, g["guidance"]) < 1e-4
    assert gd.denoise_fn.train_backend is None          # restored: sampling stays on the HIP path


@pytest.mark.gpu
def test_reconstruction_guidance_sampling_gpu():
    """The same gradient on the GPU (torch autograd ops), then a guided p_sample step whose denoiser call runs on
    the HIP kernels (reference version1diffusion.py:572-592)."""
    V1, g, gd, z, prev = _guidance_setup("cuda:0")
    fn = V1.temporal_consistency_guidance(gd, prev, 2, w_r=3.0)
    t = g["g_t"].to("cuda:0")
    out = fn(z, t)
    assert rel_err(out.cpu(), g["guidance"]) < 2e-3
    noise = inp("v1.gz", tuple(z.shape)).to("cuda:0")
    stepped = gd.p_sample(z, t, guidance_fn=fn, guidance_scale=2.0, noise=noise)
    with torch.no_grad():
        from video_diffusion_amd.diffusion import extract
        x_g = z + 2.0 * extract(gd.betas, t, z.shape) * out
        plain = gd.p_sample(x_g, t, noise=noise)
    assert torch.isfinite(stepped).all() and rel_err(stepped.cpu(), plain.cpu()) < 1e-3
    assert gd.denoise_fn._engine is not None


def _write_gifs(folder, n=4, frames=6, size=20):
    os.makedirs(folder, exist_ok=True)
    rng = np.random.RandomState(0)
    for i in range(n):
        imgs = [Image.fromarray(rng.randint(0, 255, (size, size + 4, 3), dtype=np.uint8)) for _ in range(frames)]
        imgs[0].save(os.path.join(folder, f"v{i}.gif"), save_all=True, append_images=imgs[1:], duration=50, loop=0)


def _build(folder, results, steps, seed=0):
    from video_diffusion_amd import GaussianDiffusion, Trainer, Unet3D, VAE
    from video_diffusion_amd.vae import FrameConditioningEncoder
    torch.manual_seed(seed)
    unet = Unet3D(dim=32, dim_mults=(1, 2), channels=4)
    vae = VAE(channels=3, latent_dim=4, base_dim=32)
    enc = FrameConditioningEncoder(4, 512, 2)
    gd = GaussianDiffusion(unet, vae, image_size=16, num_frames=3, channels=3, timesteps=10, frame_cond_encoder=enc)
    return Trainer(gd, folder, num_prev_frames=2, train_batch_size=2, train_num_steps=steps, gradient_accumulate_every=2,
                   step_start_ema=1, update_ema_every=1, save_and_sample_every=10 ** 6, results_folder=results)


def test_dataset_and_trainer_steps(tmp_path):
    from video_diffusion_amd.trainer import Dataset
    folder, results = str(tmp_path / "gifs"), str(tmp_path / "results")
    _write_gifs(folder)
    ds = Dataset(folder, 16, channels=3, num_frames=3, num_prev_frames=2)
    item = ds[0]
    assert item["current"].shape == (3, 3, 16, 16) and item["prev"].shape == (3, 2, 16, 16)
    assert 0.0 <= float(item["current"].min()) and float(item["current"].max()) <= 1.0
    tr = _build(folder, results, steps=2)
    before = {k: v.clone() for k, v in tr.model.denoise_fn.state_dict().items()}
    logs = []
    tr.train(log_fn=logs.append)
    assert tr.step == 2 and len(logs) == 2 and all(np.isfinite(l["loss"]) and np.isfinite(l["vae_loss"]) for l in logs)
    after = tr.model.denoise_fn.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before)            # the optimizer moved the denoiser
    assert tr.model.denoise_fn.frame_cross_attns_down[0].fn.fn.to_q.weight.grad is None   # zero_grad ran
    # EMA follows the model with decay 0.995 after step_start_ema
    k = "init_conv.weight"
    ema, cur = tr.ema_model.denoise_fn.state_dict()[k], after[k]
    assert not torch.equal(ema, cur) and rel_err(ema, cur) < 0.1
    # checkpoint format of the reference: model-{milestone}.pt with step / model / ema / scaler / opt
    tr.save(7)
    data = torch.load(os.path.join(results, "model-7.pt"), weights_only=False)
    assert set(data) == {"step", "model", "ema", "scaler", "opt"} and data["step"] == 2
    tr2 = _build(folder, results, steps=2, seed=1)
    tr2.load(-1)
    assert tr2.step == 2 and torch.equal(tr2.model.denoise_fn.state_dict()[k], cur)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, folder, results, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tr = _build(folder, results + str(rank), steps=2)       # same seed: identical initial weights on both ranks
    assert tr.distributed
    tr.train()
    flat = torch.cat([p.detach().flatten() for p in tr.model.denoise_fn.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        q.put((bool(torch.equal(gathered[0], gathered[1])), float(flat.abs().sum())))
    dist.destroy_process_group()


def test_data_parallel_training_keeps_replicas_identical(tmp_path):
    folder, results = str(tmp_path / "gifs"), str(tmp_path / "results")
    _write_gifs(folder)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, folder, results, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    same, norm = q.get(timeout=10)
    assert same and np.isfinite(norm)       # different data shards, all-reduced gradients: replicas stay bit-identical


@pytest.mark.gpu
def test_trainer_on_gpu_trains_samples_and_saves(tmp_path):
    """Two optimizer steps on the GPU (torch-autograd backward, HIP VAE inside the loss) followed by the
    milestone: frame-conditioned and unconditional sampling of the EMA model on the HIP path, GIFs, checkpoint."""
    from video_diffusion_amd import GaussianDiffusion, Trainer, Unet3D, VAE
    from video_diffusion_amd.vae import FrameConditioningEncoder
    folder, results = str(tmp_path / "gifs"), str(tmp_path / "results")
    _write_gifs(folder, n=4, frames=6, size=36)
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    unet = Unet3D(dim=32, dim_mults=(1, 2), channels=4)
    vae = VAE(channels=3, latent_dim=4, base_dim=64)
    enc = FrameConditioningEncoder(4, 512, 2)
    gd = GaussianDiffusion(unet, vae, image_size=32, num_frames=3, channels=3, timesteps=4, frame_cond_encoder=enc).to(dev)
    gd.noise_source = "philox"
    tr = Trainer(gd, folder, num_prev_frames=2, train_batch_size=2, train_num_steps=2, gradient_accumulate_every=1,
                 step_start_ema=0, update_ema_every=1, save_and_sample_every=2, results_folder=results, num_sample_rows=2)
    logs = []
    tr.train(log_fn=logs.append)
    assert len(logs) == 2 and all(np.isfinite(l["loss"]) for l in logs)
    assert os.path.exists(os.path.join(results, "model-0.pt"))
    assert os.path.exists(os.path.join(results, "0_cond_with_context.gif"))
    assert os.path.exists(os.path.join(results, "0_unconditioned.gif"))
    assert tr.ema_model.denoise_fn._engine is not None          # the milestone sampled through the HIP engine


@pytest.mark.gpu
def test_trainer_hip_resnet_backend_follows_torch_backend(tmp_path):
    """The Trainer loop (reference vdp.py:1414-1453) with the ResnetBlocks' forward + backward on the HIP kernels
    (`trainer.train_backend = "hip-resnet"`): same seed, same data, three optimizer steps - the loss of every step
    stays within 3e-2 of the all-torch backend's, and the milestone still samples through the HIP engine."""
    from video_diffusion_amd import GaussianDiffusion, Trainer, Unet3D, VAE
    from tests.helpers import record
    folder = str(tmp_path / "gifs")
    _write_gifs(folder, n=4, frames=6, size=36)
    dev = torch.device("cuda:0")

    def run(backend, results, amp=False):
        torch.manual_seed(0)
        unet = Unet3D(dim=64, dim_mults=(1, 2), channels=4)
        vae = VAE(channels=3, latent_dim=4, base_dim=64)
        gd = GaussianDiffusion(unet, vae, image_size=32, num_frames=4, channels=3, timesteps=50).to(dev)
        tr = Trainer(gd, folder, num_prev_frames=2, train_batch_size=2, train_num_steps=3, gradient_accumulate_every=1,
                     step_start_ema=0, update_ema_every=1, save_and_sample_every=10 ** 6, results_folder=results, amp=amp)
        tr.train_backend = backend
        torch.manual_seed(1)                      # the data order and the (t, noise) draws of the loss
        logs = []
        tr.train(log_fn=logs.append)
        return [l["loss"] for l in logs], tr

    ref, _ = run("torch-autograd", str(tmp_path / "r0"))
    for k, backend in enumerate(("hip-resnet", "hip")):      # "hip": temporal attention and resampling convs too
        hip, tr = run(backend, str(tmp_path / f"r{k + 1}"))
        assert len(ref) == len(hip) == 3
        for i, (a, b) in enumerate(zip(hip, ref)):
            assert record(f"trainer_{backend}.loss_step{i}", abs(a - b) / abs(b), 3e-2) < 3e-2
        assert tr.model.denoise_fn.train_backend == backend
    # "auto" resolves to the HIP layers for this model (every width a power of two from 64) ...
    auto, tr = run("auto", str(tmp_path / "r_auto"))
    assert tr.model.denoise_fn.train_backend == "hip" and tr.opt.defaults.get("fused")
    for i, (a, b) in enumerate(zip(auto, ref)):
        assert record(f"trainer_auto.loss_step{i}", abs(a - b) / abs(b), 3e-2) < 3e-2
    # Trainer(amp=True): the torch layers run under autocast with a gradient scaler, the HIP layers take the scaled
    # gradients in bf16 (fp32's exponent range); the loss follows the full-precision run
    amp_losses, _ = run("hip", str(tmp_path / "r_amp"), amp=True)
    for i, (a, b) in enumerate(zip(amp_losses, ref)):
        assert record(f"trainer_hip_amp.loss_step{i}", abs(a - b) / abs(b), 5e-2) < 5e-2


@pytest.mark.gpu
def test_reconstruction_guidance_gradient_on_hip_kernels_matches_torch_ops():
    """temporal_consistency_guidance (version1diffusion.py:707-747) on a denoiser whose shapes the training kernels take:
    the gradient through the HIP Blocks / temporal attention / resampling convolutions against the all-torch gradient."""
    from video_diffusion_amd import VAE, train_hip, version1 as V1
    from tests.helpers import record
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    unet = V1.Unet3D(dim=64, dim_mults=(1, 2), channels=4)
    vae = VAE(channels=3, latent_dim=4, base_dim=64)
    gd = V1.GaussianDiffusion(unet.eval(), vae.eval(), image_size=32, num_frames=3, channels=3, timesteps=250).to(dev)
    assert train_hip.supports(gd.denoise_fn)
    small = V1.Unet3D(dim=32, dim_mults=(1, 2), channels=4)
    assert not train_hip.supports(small)                    # 32-channel blocks: the guidance falls back to torch ops
    z = torch.randn(2, 4, 3, 4, 4, device=dev)
    t = torch.tensor([40, 200], device=dev)
    prev = torch.rand(2, 3, 2, 32, 32, device=dev)
    fn = V1.temporal_consistency_guidance(gd, prev, 2, w_r=3.0)
    g_hip = fn(z, t)
    orig = train_hip.supports
    train_hip.supports = lambda m: False                    # the all-torch path
    try:
        g_ref = fn(z, t)
    finally:
        train_hip.supports = orig
    assert gd.denoise_fn.train_backend is None
    # measured 4.4e-2: the gradient runs backwards through a bf16 denoiser whose epsilon error the x0 prediction amplifies
    assert record("v1_guidance.hip_vs_torch", rel_err(g_hip, g_ref), 8e-2) < 8e-2


def test_guidance_variant_gradient_path_matches_reference_and_oracle():
    """guidance_variant.Unet3D with a gradient asked for (video_diffusion_guidance.py:725-761 differentiates the
    reconstruction loss through the denoiser): the torch-op forward reproduces that file's own output (golden made by
    oracle/make_golden_guidance.py) and its input gradient is the oracle's."""
    from oracle import vdm_oracle as O, weightgen
    from tests.helpers import golden, inp
    from video_diffusion_amd import guidance_variant as GV
    g = golden("guidance_tiny")
    m = GV.Unet3D(dim=64, dim_mults=(1, 2), channels=4)
    m.load_state_dict(weightgen.fill_state_dict(m.state_dict(), 4))
    x = inp("g.x", tuple(int(v) for v in g["x_shape"])).requires_grad_(True)
    t = torch.as_tensor(g["t"])
    y = m(x, t)
    assert rel_err(y.detach(), g["eps"]) < 1e-5
    gx, = torch.autograd.grad(y.square().mean(), x)
    cfg = O.UnetCfg(dim=64, dim_mults=(1, 2), channels=4, max_distance=300)
    x2 = x.detach().clone().requires_grad_(True)
    g2, = torch.autograd.grad(O.unet3d_guidance_forward(m.state_dict(), cfg, x2, t).square().mean(), x2)
    assert rel_err(gx, g2) < 1e-4
    # training: parameter gradients with the backend the Trainer sets; anything else is refused, not ignored
    m.train_backend = "torch-autograd"
    m(x.detach(), t).square().mean().backward()
    missing = [n for n, p in m.named_parameters() if p.grad is None]
    assert all(n.startswith("final_conv.0.mlp") for n in missing), missing      # that block runs without the embedding
    m.train_backend = "hip"                                  # the HIP layers have no CPU path
    with pytest.raises(Exception, match="needs CUDA"):
        m(x.detach(), t)
    m.train_backend = "something-else"
    with pytest.raises(Exception, match="differentiates through"):
        m(x.detach(), t)


@pytest.mark.gpu
def test_guidance_variant_reconstruction_guidance_on_the_gpu():
    """The guidance sampler of video_diffusion_guidance.py on the GPU: the gradient step differentiates through torch ops
    over the variant's parameters, the denoising step that follows runs on the HIP plan; the two forwards agree to the bf16
    bar and the gradient equals the one computed on the CPU."""
    from video_diffusion_amd import VAE, guidance_variant as GV
    from tests.helpers import record
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    unet = GV.Unet3D(dim=64, dim_mults=(1, 2), channels=4)
    vae = VAE(channels=3, latent_dim=4, base_dim=64)
    gd = GV.GaussianDiffusion(unet.eval(), vae.eval(), image_size=32, num_frames=3, channels=3, timesteps=250)
    z = torch.randn(2, 4, 3, 4, 4)
    t = torch.tensor([40, 200])
    prev = torch.rand(2, 3, 2, 32, 32)
    g_cpu = GV.temporal_consistency_guidance(gd, prev, 2, w_r=3.0)(z, t)
    gd = gd.to(dev)
    zd, td = z.to(dev), t.to(dev)
    from video_diffusion_amd import train_hip_variant as HV
    orig = HV.supports
    HV.supports = lambda m: False               # float torch ops on the GPU (the HIP layers: tests/test_variant_backward_gpu.py)
    try:
        g_gpu = GV.temporal_consistency_guidance(gd, prev.to(dev), 2, w_r=3.0)(zd, td)
    finally:
        HV.supports = orig
    assert bool(torch.isfinite(g_gpu).all()) and float(g_gpu.abs().max()) > 0
    assert record("gv_guidance.gpu_vs_cpu", rel_err(g_gpu.cpu(), g_cpu), 1e-3) < 1e-3
    with torch.no_grad():
        e_hip = gd.denoise_fn(zd, td)
    e_ops = gd.denoise_fn(zd.clone().requires_grad_(True), td).detach()
    assert record("gv_guidance.hip_forward_vs_torch_ops", rel_err(e_hip, e_ops), 2e-2) < 2e-2
    x = gd.p_sample(zd, td, guidance_fn=GV.temporal_consistency_guidance(gd, prev.to(dev), 2, w_r=3.0), guidance_scale=1.0,
                    noise=torch.zeros_like(zd))
    assert x.shape == zd.shape and bool(torch.isfinite(x).all())


def test_resample_gather_tables_reproduce_the_rearranged_kernels():
    """train_hip._resample_gathers: the index tables the weight packer gathers through give, applied to the flattened
    parameter, exactly the re-arranged kernels written out with torch ops - the 2x2 / stride-1 kernel over the
    space-to-depth view for Downsample's Conv3d(1,4,4) (reference vdp.py:377-378) and the four per-parity tap sets of
    Upsample's ConvTranspose3d (vdp.py:374-375)."""
    from video_diffusion_amd import train_hip as T
    g = torch.Generator().manual_seed(5)
    n, cin = 6, 4
    w = torch.randn(n, cin, 1, 4, 4, generator=g)                      # Conv3d weight (out, in, 1, 4, 4)
    (tab,) = T._resample_gathers("down", w)
    w4 = w[:, :, 0]
    ref = torch.cat([torch.stack([w4[:, :, 2 * dy + a, 2 * dx + b] for dy in (0, 1) for dx in (0, 1)], dim=-1)
                     for a in (0, 1) for b in (0, 1)], dim=1)
    assert tab.dtype == torch.int32 and tab.shape == (n, 4 * cin, 4)
    assert torch.equal(w.flatten()[tab.long()], ref)
    wt = torch.randn(cin, n, 1, 4, 4, generator=g)                     # ConvTranspose3d weight (in, out, 1, 4, 4)
    tabs = T._resample_gathers("up", wt)
    w4 = wt[:, :, 0]
    assert len(tabs) == 4
    for tab, (pp, q) in zip(tabs, [(pp, q) for pp in (0, 1) for q in (0, 1)]):
        ref = torch.stack([w4[:, :, ky, kx].t() for (ky, _) in T._UP_PAIRS[pp] for (kx, _) in T._UP_PAIRS[q]], dim=-1)
        assert tab.shape == (n, cin, 4) and torch.equal(wt.flatten()[tab.long()], ref)


@pytest.mark.gpu
def test_trainer_hip_backend_with_gradient_accumulation_keeps_weight_sets_fresh(tmp_path):
    """Trainer(gradient_accumulate_every=2) on the "hip" backend: the packed weight sets are kept over the two micro-steps of
    an iteration and repacked after every (fused, version-counter-silent) optimizer step - the losses of three iterations
    follow the all-torch backend's within 3e-2 (a stale set shows up as a few percent from the second iteration on)."""
    from video_diffusion_amd import GaussianDiffusion, Trainer, Unet3D, VAE
    from tests.helpers import record
    folder = str(tmp_path / "gifs")
    _write_gifs(folder, n=4, frames=6, size=36)
    dev = torch.device("cuda:0")

    def run(backend, results):
        torch.manual_seed(0)
        unet = Unet3D(dim=64, dim_mults=(1, 2), channels=4)
        vae = VAE(channels=3, latent_dim=4, base_dim=64)
        gd = GaussianDiffusion(unet, vae, image_size=32, num_frames=4, channels=3, timesteps=50).to(dev)
        tr = Trainer(gd, folder, num_prev_frames=2, train_batch_size=2, train_num_steps=3, gradient_accumulate_every=2,
                     train_lr=1e-3, step_start_ema=0, update_ema_every=1, save_and_sample_every=10 ** 6, results_folder=results)
        tr.train_backend = backend
        torch.manual_seed(1)
        logs = []
        tr.train(log_fn=logs.append)
        return [l["loss"] for l in logs]

    ref = run("torch-autograd", str(tmp_path / "r0"))
    hip = run("hip", str(tmp_path / "r1"))
    assert len(ref) == len(hip) == 3
    for i, (a, b) in enumerate(zip(hip, ref)):
        assert record(f"trainer_hip_accum2.loss_step{i}", abs(a - b) / abs(b), 3e-2) < 3e-2


def test_time_tables_equal_the_blocks_own_time_mlps():
    """train_hip.time_tables: the ResnetBlocks' time MLPs (vdp.py:432-436, 445-447) as one Linear over the concatenated
    weights give every block the values and the parameter / embedding gradients of its own MLP (CPU: plain torch ops)."""
    import torch.nn.functional as F
    from video_diffusion_amd import Unet3D, train_hip
    torch.manual_seed(0)
    unet = Unet3D(dim=32, dim_mults=(1, 2), channels=1)
    blocks = [m for m in unet.modules() if getattr(m, "mlp", None) is not None and hasattr(m, "block1")]
    temb = torch.randn(3, blocks[0].mlp[1].in_features, requires_grad=True)
    tables = train_hip.time_tables(unet, temb).tables
    assert tables is not None and len(tables) == len(blocks) >= 8
    own = [m.mlp[1](F.silu(temb)) for m in blocks]
    for m, e in zip(blocks, own):
        assert torch.allclose(tables[id(m)], e, atol=1e-6)
    weights = [m.mlp[1].weight for m in blocks] + [m.mlp[1].bias for m in blocks] + [temb]
    probe = [torch.randn_like(e) for e in own]
    g_tab = torch.autograd.grad(sum((tables[id(m)] * p).sum() for m, p in zip(blocks, probe)), weights)
    g_own = torch.autograd.grad(sum((e * p).sum() for e, p in zip(own, probe)), weights)
    for a, b in zip(g_tab, g_own):
        assert torch.allclose(a, b, atol=1e-5)
    # a model without time-conditioned blocks, or a foreign embedding width: no tables, the blocks run their own MLPs
    assert train_hip.time_tables(unet, torch.randn(3, 7)).tables is None
    assert train_hip.time_tables(unet, None).tables is None
