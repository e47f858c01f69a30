"""GaussianDiffusion with the reference's surface (vdp.py:835-1182 of the reference =
video_diffusion_pytorch/video_diffusion_pytorch.py).  The sampling step (predict x0, clip,
posterior mean, add noise — vdp.py:985-1002) is one HIP kernel (vdmi_ddpm_step); the sequential
loop replays one captured HIP graph per step when the denoiser is the HIP Unet3D.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from ._lib import DdpmCoefs, check


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def extract(a, t, x_shape):
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


def cosine_beta_schedule(timesteps, s=0.008):
    # float64 throughout, as the reference (vdp.py:866-872)
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    alphas_cumprod = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0, 0.9999)


class _StaggeredSteps:
    """replay(): one denoising step of every part, each from its own captured graph on its own stream.  begin() orders
    the part streams behind the work already queued on the current stream (the copy of x_T), finish() orders the
    current stream behind the parts.  The first replay after begin() launches the parts `stagger_s` apart (the host
    sleeps in between; every later launch just queues behind the part's previous step)."""

    def __init__(self, graphs, streams, counters, device, stagger_s):
        self.graphs, self.streams, self.counters, self.device, self.stagger_s = graphs, streams, counters, device, stagger_s
        self._begun = self._first = False

    def begin(self):
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(main)
        self._begun = self._first = True

    def replay(self):
        if not self._begun:
            self.begin()
        for k, (g, s) in enumerate(zip(self.graphs, self.streams)):
            if k and self._first and self.stagger_s > 0:
                time.sleep(self.stagger_s)
            with torch.cuda.stream(s):
                g.replay()
        self._first = False

    def finish(self):
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            main.wait_stream(s)
        self._begun = False


class GaussianDiffusion(nn.Module):
    def __init__(
        self,
        denoise_fn,
        vae,
        image_size,
        num_frames,
        channels=3,
        timesteps=1000,
        loss_type='l1',
        use_dynamic_thres=False,
        dynamic_thres_percentile=0.9,
        latent_scale=1.0,
        frame_cond_encoder=None,
    ):
        super().__init__()
        self.channels = channels
        self.image_size = image_size
        self.num_frames = num_frames
        self.denoise_fn = denoise_fn
        self.vae = vae
        self.frame_cond_encoder = frame_cond_encoder
        self.latent_scale = latent_scale

        for param in self.vae.parameters():
            param.requires_grad = False

        betas = cosine_beta_schedule(timesteps)
        alphas = 1. - betas
        alphas_cumprod = torch.cumprod(alphas, axis=0)
        alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.)

        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
        self.loss_type = loss_type

        register_buffer = lambda name, val: self.register_buffer(name, val.to(torch.float32))
        register_buffer('betas', betas)
        register_buffer('alphas_cumprod', alphas_cumprod)
        register_buffer('alphas_cumprod_prev', alphas_cumprod_prev)
        register_buffer('sqrt_alphas_cumprod', torch.sqrt(alphas_cumprod))
        register_buffer('sqrt_one_minus_alphas_cumprod', torch.sqrt(1. - alphas_cumprod))
        register_buffer('log_one_minus_alphas_cumprod', torch.log(1. - alphas_cumprod))
        register_buffer('sqrt_recip_alphas_cumprod', torch.sqrt(1. / alphas_cumprod))
        register_buffer('sqrt_recipm1_alphas_cumprod', torch.sqrt(1. / alphas_cumprod - 1))
        posterior_variance = betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        register_buffer('posterior_variance', posterior_variance)
        register_buffer('posterior_log_variance_clipped', torch.log(posterior_variance.clamp(min=1e-20)))
        register_buffer('posterior_mean_coef1', betas * torch.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
        register_buffer('posterior_mean_coef2', (1. - alphas_cumprod_prev) * torch.sqrt(alphas) / (1. - alphas_cumprod))

        self.use_dynamic_thres = use_dynamic_thres
        self.dynamic_thres_percentile = dynamic_thres_percentile

        # "torch": per-step noise from torch.randn_like (the reference's RNG consumption order);
        # "philox": drawn inside the step kernel (no extra pass over HBM).
        self.noise_source = "torch"
        self.use_hip_graph = True
        # True: a batch that the engine runs as sub-batch plans loops part by part, each part on its own stream with
        # nothing joining them between steps (samples never interact, vdp.py:1004-1014), the second part started
        # pipeline_stagger_s after the first.  False (default): one graph per step, the parts forked and joined inside it.
        # Same bits either way; measured on configs[1] the two take the same time whatever the stagger (3.63-3.65 ms per
        # step: the compute units are busy throughout, so moving one part's small-image layers beside the other's
        # large-image layers gains nothing).
        self.pipeline_parts = os.environ.get("VDMI_PIPELINE", "0") != "0"
        self.pipeline_stagger_s = float(os.environ.get("VDMI_PIPELINE_STAGGER_US", "0")) * 1e-6
        self._graphs = {}
        self._rng = {}

    # -- closed forms kept for API compatibility (elementwise torch expressions) ------------
    def q_mean_variance(self, x_start, t):
        mean = extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
        variance = extract(1. - self.alphas_cumprod, t, x_start.shape)
        log_variance = extract(self.log_one_minus_alphas_cumprod, t, x_start.shape)
        return mean, variance, log_variance

    def predict_start_from_noise(self, x_t, t, noise):
        return (extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t -
                extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def q_posterior(self, x_start, x_t, t):
        posterior_mean = (extract(self.posterior_mean_coef1, t, x_t.shape) * x_start +
                          extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        posterior_variance = extract(self.posterior_variance, t, x_t.shape)
        posterior_log_variance_clipped = extract(self.posterior_log_variance_clipped, t, x_t.shape)
        return posterior_mean, posterior_variance, posterior_log_variance_clipped

    def _dynamic_clip(self, x_recon):
        s = torch.quantile(x_recon.reshape(x_recon.shape[0], -1).abs(), self.dynamic_thres_percentile, dim=-1)
        s.clamp_(min=1.)
        s = s.view(-1, *((1,) * (x_recon.ndim - 1)))
        return x_recon.clamp(-s, s) / s

    def p_mean_variance(self, x, t, clip_denoised=True):
        x_recon = self.predict_start_from_noise(x, t=t, noise=self.denoise_fn(x, t))
        if clip_denoised:
            x_recon = self._dynamic_clip(x_recon) if self.use_dynamic_thres else x_recon.clamp(-1., 1.)
        return self.q_posterior(x_start=x_recon, x_t=x, t=t)

    # -- the fused step ---------------------------------------------------------------------
    def _coefs(self) -> DdpmCoefs:
        c = DdpmCoefs()
        c.sqrt_recip_alphas_cumprod = self.sqrt_recip_alphas_cumprod.data_ptr()
        c.sqrt_recipm1_alphas_cumprod = self.sqrt_recipm1_alphas_cumprod.data_ptr()
        c.posterior_mean_coef1 = self.posterior_mean_coef1.data_ptr()
        c.posterior_mean_coef2 = self.posterior_mean_coef2.data_ptr()
        c.posterior_log_variance_clipped = self.posterior_log_variance_clipped.data_ptr()
        return c

    def _rng_state(self, device) -> torch.Tensor:
        """Device words {seed, offset} of the in-kernel noise: int64 [2], one buffer per device."""
        key = str(device)
        if key not in self._rng:
            self._rng[key] = torch.zeros(2, device=device, dtype=torch.int64)
        return self._rng[key]

    def reseed_noise(self, device, seed=None) -> int:
        """New Philox key for the next sampling loop on `device` (drawn from torch's default CPU generator when not
        given, so torch.manual_seed makes loops reproducible and differently seeded ranks draw different noise).
        Captured graphs read the key from device memory: every loop, replayed or eager, sees the new one."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._rng_state(device).copy_(torch.tensor([seed, 0], dtype=torch.int64))
        return seed

    def ddpm_step(self, x, eps, t, noise=None, clip_denoised=True, out=None, seed=0, offset=0, rng_state=None,
                  advance_t=None, batch_slice=(0, 0)):
        """x_{t-1} from (x_t, eps, noise) in one HIP kernel.  noise=None draws Philox noise in-kernel, keyed by
        (seed, offset) or, when rng_state (device int64 [2]) is given, by the two words it holds.
        advance_t (device int32 [1], zero; needs rng_state and the static clip): the kernel also moves t on to t - 1
        (num_timesteps - 1 after 0), for captured steps that are replayed without the host writing t; batch_slice =
        (total, first): x holds samples [first, first + b) of a batch of `total` that is stepped in independent parts, and
        draws the noise the whole batch would."""
        if not x.is_cuda:
            raise _lib.VdmiError("ddpm_step runs on the HIP device only")
        x = x.contiguous()
        eps = eps.to(torch.float32).contiguous()
        out = torch.empty_like(x) if out is None else out
        b = x.shape[0]
        if clip_denoised and self.use_dynamic_thres:
            # per-sample quantile threshold (reference vdp.py:957-968): one selection kernel, then the step kernel
            # clamps to +-s and divides by s
            lib, stream = _lib.lib(), torch.cuda.current_stream().cuda_stream
            thres = torch.empty(b, device=x.device, dtype=torch.float32)
            check(lib.vdmi_dynamic_threshold(x.data_ptr(), eps.data_ptr(), t.data_ptr(), C.byref(self._coefs()),
                                             thres.data_ptr(), b, x.numel() // b, float(self.dynamic_thres_percentile),
                                             stream), "vdmi_dynamic_threshold")
            if noise is None and rng_state is None:      # a key of its own for this call
                rng_state = torch.tensor([seed, offset], device=x.device, dtype=torch.int64)
            check(lib.vdmi_ddpm_step_thres(
                x.data_ptr(), eps.data_ptr(), None if noise is None else noise.contiguous().data_ptr(), t.data_ptr(),
                C.byref(self._coefs()), thres.data_ptr(), out.data_ptr(), b, x.numel() // b,
                None if noise is not None else rng_state.data_ptr(), stream), "vdmi_ddpm_step_thres")
            return out
        if noise is None and rng_state is not None and advance_t is not None:
            check(_lib.lib().vdmi_ddpm_step_loop(
                x.data_ptr(), eps.data_ptr(), t.data_ptr(), C.byref(self._coefs()), out.data_ptr(), b, x.numel() // b,
                int(bool(clip_denoised)), rng_state.data_ptr(), advance_t.data_ptr(), int(self.num_timesteps),
                int(batch_slice[0]), int(batch_slice[1]), torch.cuda.current_stream().cuda_stream), "vdmi_ddpm_step_loop")
            return out
        if noise is None and rng_state is not None:
            check(_lib.lib().vdmi_ddpm_step_rng(
                x.data_ptr(), eps.data_ptr(), t.data_ptr(), C.byref(self._coefs()), out.data_ptr(), b, x.numel() // b,
                int(bool(clip_denoised)), rng_state.data_ptr(), torch.cuda.current_stream().cuda_stream),
                "vdmi_ddpm_step_rng")
            return out
        check(_lib.lib().vdmi_ddpm_step(
            x.data_ptr(), eps.data_ptr(), None if noise is None else noise.contiguous().data_ptr(),
            t.data_ptr(), C.byref(self._coefs()), out.data_ptr(), b, x.numel() // b, int(bool(clip_denoised)),
            seed, offset, torch.cuda.current_stream().cuda_stream), "vdmi_ddpm_step")
        return out

    def _predict_noise(self, x, t, frame_cond_features=None):
        # the call of reference vdp.py:978-983
        return self.denoise_fn(x, t, frame_cond_features=frame_cond_features, null_frame_cond_prob=0.)

    @torch.inference_mode()
    def p_sample(self, x, t, frame_cond_features=None, clip_denoised=True, noise=None, rng_state=None):
        # reference: vdp.py:973-1002.  rng_state: with noise_source "philox", the device words a sampling loop keyed
        # its noise with (None: a fresh key per call)
        predicted_noise = self._predict_noise(x, t, frame_cond_features)
        if noise is None and self.noise_source == "torch":
            noise = torch.randn_like(x)
        seed = 0 if (noise is not None or rng_state is not None) else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        return self.ddpm_step(x.to(torch.float32), predicted_noise, t, noise=noise, clip_denoised=clip_denoised,
                              seed=seed, rng_state=rng_state)

    # -- sampling loop ------------------------------------------------------------------------
    def _graph_runner(self, shape, device):
        """One denoise step (Unet3D plan + posterior kernel) captured in a HIP graph."""
        from .modules import Unet3D
        fn = self.denoise_fn
        if not (self.use_hip_graph and isinstance(fn, Unet3D) and not fn.has_cond):
            return None
        fn.engine().ensure_packed(device, check_values=True)
        pack_id = fn.engine().pack_id
        key = (tuple(shape), str(device), self.noise_source, pack_id, bool(self.use_dynamic_thres),
               float(self.dynamic_thres_percentile), bool(self.pipeline_parts))
        if key in self._graphs:
            return self._graphs[key]
        for old in [k for k in self._graphs if k[0] != "cond" and k[3] != pack_id]:      # graphs over weights repacked since
            del self._graphs[old]
        B, Cc, Fr, H, W = shape
        # every step of the loop uses t in [0, num_timesteps): scale/shift rows come from the engine's timestep tables
        plan = fn.engine().plan(B, Fr, H, W, device, time_table=self.num_timesteps)
        z = torch.zeros(shape, device=device, dtype=torch.float32) if self.noise_source == "torch" else None
        rng = self._rng_state(device)       # read by the captured kernel at every replay: the loop rewrites it per call
        # with in-kernel noise and the static clip the posterior kernel also counts t down: the loop sets t once
        self_advancing = z is None and not self.use_dynamic_thres
        tcount = torch.zeros(1, device=device, dtype=torch.int32) if self_advancing else None

        def step():
            plan.run()
            if z is not None:
                z.normal_()
            self.ddpm_step(plan.x_in, plan.eps_out, plan.t_in, noise=z, clip_denoised=True, out=plan.x_in, rng_state=rng,
                           advance_t=tcount)

        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            saved = plan.x_in.clone()
            plan.t_in.fill_(0)
            step()                      # warm-up outside capture (sets kernel attributes, allocates)
            plan.x_in.copy_(saved)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        plan.self_advancing = self_advancing
        plan.step_counter = tcount          # the captured kernel counts in it at every replay: it lives as long as the plan
        parts = getattr(plan, "parts", None)
        if self_advancing and self.pipeline_parts and isinstance(parts, list) and len(parts) > 1:
            graph = self._staggered_steps(plan, rng, device)        # one graph per part instead of the joined step
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
        runner = (graph, plan)
        self._graphs[key] = runner
        return runner

    def _staggered_steps(self, plan, rng, device):
        """The sub-batch plans of a SplitPlan as independent loops: one captured graph per part (its forward, then the
        posterior step of its samples, which also counts its t down), each replayed on a stream of its own with nothing
        joining them between steps."""
        graphs, streams, keep = [], [], []
        first = 0
        for part in plan.parts:
            n = part.x_in.shape[0]
            tcount = torch.zeros(1, device=device, dtype=torch.int32)
            stream = torch.cuda.Stream(device=device)

            def step(part=part, tcount=tcount, first=first):
                part.run(fork=False)
                self.ddpm_step(part.x_in, part.eps_out, part.t_in, noise=None, clip_denoised=True, out=part.x_in,
                               rng_state=rng, advance_t=tcount, batch_slice=(plan.B, first))

            stream.wait_stream(torch.cuda.current_stream(device))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                step()
            graphs.append(g)
            streams.append(stream)
            keep.append(tcount)
            first += n
        torch.cuda.synchronize(device)
        return _StaggeredSteps(graphs, streams, keep, device, self.pipeline_stagger_s)

    @torch.inference_mode()
    def p_sample_loop(self, shape, x_T=None, noises=None, cond=None, cond_scale=1.):
        """reference: vdp.py:1004-1014.  x_T / noises (one tensor per step, in execution order) may
        be injected for parity checks; by default they are drawn as the reference draws them.
        cond / cond_scale: the condition vectors of a Unet3D(cond_dim=...) and the classifier-free guidance scale
        of its forward_with_cond_scale (reference vdp.py:731-748); see cond_sample_loop."""
        if cond is not None:
            return self.cond_sample_loop(shape, cond, cond_scale, x_T=x_T, noises=noises)
        device = self.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        if hasattr(self.denoise_fn, "engine"):
            # once per loop: weights rewritten through `.data` since the last packing are noticed here
            self.denoise_fn.engine().ensure_packed(device, check_values=True)
        runner = self._graph_runner(shape, device) if noises is None else None
        rng = None
        if noises is None and self.noise_source == "philox":
            self.reseed_noise(device)           # one key per loop; graph replay and the eager loop draw the same field
            rng = self._rng_state(device)
        if runner is not None:
            graph, plan = runner
            plan.x_in.copy_(img)
            plan.t_in.fill_(self.num_timesteps - 1)
            if isinstance(graph, _StaggeredSteps):
                graph.begin()
            for i in reversed(range(0, self.num_timesteps)):
                if not plan.self_advancing:
                    plan.t_in.fill_(i)
                graph.replay()      # (a self-advancing step leaves t at i - 1 for the next replay)
            if isinstance(graph, _StaggeredSteps):
                graph.finish()
            return plan.x_in.clone()
        for k, i in enumerate(reversed(range(0, self.num_timesteps))):
            t = torch.full((b,), i, device=device, dtype=torch.long)
            img = self.p_sample(img, t, noise=None if noises is None else noises[k].to(device), rng_state=rng)
        return img

    def _cond_runner(self, shape, device, rep):
        """Plan over rep * B samples, state / epsilon buffers and (with use_hip_graph) the captured step of the
        condition-vector loop; cached per shape like the unconditional runner."""
        fn = self.denoise_fn
        eng = fn.engine()
        eng.ensure_packed(device, check_values=True)
        key = ("cond", tuple(shape), rep, str(device), self.noise_source, eng.pack_id, bool(self.use_hip_graph),
               bool(self.use_dynamic_thres), float(self.dynamic_thres_percentile))
        if key in self._graphs:
            return self._graphs[key]
        for old in [k for k in self._graphs if k[0] == "cond" and k[5] != eng.pack_id]:
            del self._graphs[old]
        B, Cc, Fr, H, W = shape
        plan = eng.plan(rep * B, Fr, H, W, device)      # the condition enters the time embedding: no timestep tables
        state = torch.zeros(shape, device=device, dtype=torch.float32)
        eps = torch.empty(shape, device=device, dtype=torch.float32)
        scale = torch.ones((), device=device, dtype=torch.float32)     # guidance scale, read by the captured step
        z = torch.zeros(shape, device=device, dtype=torch.float32) if self.noise_source == "torch" else None
        rng = self._rng_state(device)
        t_b = plan.t_in[:B]

        def step(noise):
            plan.x_in[:B].copy_(state)
            if rep == 2:
                plan.x_in[B:].copy_(state)
            plan.run()
            if rep == 2:
                e_c, e_u = plan.eps_out[:B], plan.eps_out[B:]
                torch.addcmul(e_u, e_c - e_u, scale, out=eps)           # null + (cond - null) * scale
            else:
                eps.copy_(plan.eps_out)
            self.ddpm_step(state, eps, t_b, noise=noise, clip_denoised=True, out=state, rng_state=rng)

        def drawn_step():
            if z is not None:
                z.normal_()
            step(z)

        graph = None
        if self.use_hip_graph:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                plan.t_in.fill_(0)
                drawn_step()                # warm-up outside capture
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                drawn_step()
        runner = dict(plan=plan, state=state, scale=scale, step=step, drawn_step=drawn_step, graph=graph)
        self._graphs[key] = runner
        return runner

    @torch.inference_mode()
    def cond_sample_loop(self, shape, cond, cond_scale=1.0, x_T=None, noises=None, steps=None):
        """The denoising loop of a model with a condition vector: every step evaluates
        forward_with_cond_scale(x, t, cond=cond, cond_scale=cond_scale) (reference vdp.py:731-748; the reference's own
        sampler never passes a condition, this is the loop its guided forward is written for) and takes the
        static-clip posterior step (reference vdp.py:973-1002).

        With the HIP Unet3D the conditional and the null-conditioned evaluation of a guided step run as ONE pass over a
        doubled batch (rows B.. take the null embedding), and each step replays one captured HIP graph.  x_T / noises
        (one tensor per step, in execution order) may be injected for parity checks; `steps` stops early."""
        from .modules import Unet3D
        device = self.betas.device
        fn = self.denoise_fn
        B = shape[0]
        cond = cond.to(device=device, dtype=torch.float32)
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        guided = cond_scale != 1.0
        order = list(reversed(range(0, self.num_timesteps)))[:steps]
        if not (isinstance(fn, Unet3D) and fn.has_cond):
            for k, i in enumerate(order):       # any other denoiser: call for call
                t = torch.full((B,), i, device=device, dtype=torch.long)
                nz = None if noises is None else noises[k].to(device)
                e = fn.forward_with_cond_scale(img, t, cond=cond, cond_scale=cond_scale)
                nz = torch.randn_like(img) if nz is None else nz
                img = self.ddpm_step(img, e, t, noise=nz, clip_denoised=True)
            return img
        rep = 2 if guided else 1
        r = self._cond_runner(shape, device, rep)
        plan = r["plan"]
        plan.cond_in.copy_(cond.repeat(rep, 1))
        plan.null_mask.fill_(0)
        if guided:
            plan.null_mask[B:].fill_(1)
        plan.focus_in.fill_(0)
        r["scale"].fill_(float(cond_scale))
        r["state"].copy_(img)
        if noises is None and self.noise_source == "philox":
            self.reseed_noise(device)
        for k, i in enumerate(order):
            plan.t_in.fill_(i)
            if noises is not None:
                r["step"](noises[k].to(device=device, dtype=torch.float32).contiguous())
            elif r["graph"] is not None:
                r["graph"].replay()
            else:
                r["drawn_step"]()
        return r["state"].clone()

    @torch.inference_mode()
    def frame_cond_sample_loop(self, shape, frame_cond_features, frame_cond_scale=1.0, x_T=None, noises=None,
                               steps=None):
        """The denoising loop of sample() with previous-frame conditioning (reference vdp.py:1046-1080): with
        frame_cond_scale != 1 every step evaluates the denoiser with the conditioning kept and with it zeroed
        and extrapolates between the two; the step itself is the static-clip posterior step.

        With the HIP Unet3D the two evaluations run as ONE pass over a doubled batch (samples are
        independent, so the results are those of two passes), the keys/values of the context are
        projected once for the whole loop, and each step replays one captured HIP graph.
        x_T / noises (one tensor per step, in execution order) may be injected for parity checks; `steps`
        stops after that many steps (parity checks and benchmarks)."""
        from .modules import Unet3D
        device = self.betas.device
        fn = self.denoise_fn
        B = shape[0]
        feats = frame_cond_features.to(device=device, dtype=torch.float32)
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        guided = frame_cond_scale != 1.0
        order = list(reversed(range(0, self.num_timesteps)))[:steps]

        if not (isinstance(fn, Unet3D) and fn.has_frame_cross_attn and not fn.has_cond):
            # any other denoiser: the reference's loop, call for call
            for k, i in enumerate(order):
                t = torch.full((B,), i, device=device, dtype=torch.long)
                nz = None if noises is None else noises[k].to(device)
                if guided:
                    e_c = fn(img, t, frame_cond_features=feats, null_frame_cond_prob=0.)
                    e_u = fn(img, t, frame_cond_features=feats, null_frame_cond_prob=1.)
                    nz = torch.randn_like(img) if nz is None else nz
                    img = self.ddpm_step(img, e_u + frame_cond_scale * (e_c - e_u), t, noise=nz, clip_denoised=True)
                else:
                    img = self.p_sample(img, t, frame_cond_features=feats, noise=nz)
            return img

        rep = 2 if guided else 1
        _, Cc, Fr, H, W = shape
        eng = fn.engine()
        eng.ensure_packed(device, check_values=True)
        plan = eng.plan(rep * B, Fr, H, W, device, feats.shape[1], time_table=self.num_timesteps)
        plan.ctx_in.copy_(feats.repeat(rep, 1, 1))
        plan.ctx_keep.fill_(1)
        if guided:
            plan.ctx_keep[B:].fill_(0)
        plan.run_context()
        state = img.contiguous().clone()
        eps = torch.empty(shape, device=device, dtype=torch.float32)
        t_b = plan.t_in[:B]
        z = torch.zeros(shape, device=device, dtype=torch.float32) if self.noise_source == "torch" else None
        self.reseed_noise(device)
        rng = self._rng_state(device)

        def step(noise):
            plan.x_in[:B].copy_(state)
            if guided:
                plan.x_in[B:].copy_(state)
            plan.run()
            if guided:
                e_c, e_u = plan.eps_out[:B], plan.eps_out[B:]
                torch.add(e_u, e_c - e_u, alpha=frame_cond_scale, out=eps)
            else:
                eps.copy_(plan.eps_out)
            self.ddpm_step(state, eps, t_b, noise=noise, clip_denoised=True, out=state, rng_state=rng)

        def drawn_step():
            if z is not None:
                z.normal_()
            step(z)

        graph = None
        if noises is None and self.use_hip_graph:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                saved = state.clone()
                plan.t_in.fill_(order[0])
                drawn_step()                # warm-up outside capture
                state.copy_(saved)
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                drawn_step()
            state.copy_(saved)
        for k, i in enumerate(order):
            plan.t_in.fill_(i)
            if graph is not None:
                graph.replay()
            elif noises is not None:
                step(noises[k].to(device=device, dtype=torch.float32).contiguous())
            else:
                drawn_step()
        return state.clone()

    @torch.inference_mode()
    def sample(self, batch_size=16, prev_frames=None, frame_cond_scale=1.0):
        # reference: vdp.py:1016-1089 (latent-space sampling + VAE decode)
        device = next(self.denoise_fn.parameters()).device
        latent_size = self.image_size // 8
        latent_channels = self.vae.latent_dim
        shape = (batch_size, latent_channels, self.num_frames, latent_size, latent_size)
        if device.type == "cuda":
            # once per call: writes through `.data` into the VAE or the frame encoder since their plans were built
            from .vae import check_values
            for mod in (self.vae, self.frame_cond_encoder):
                if isinstance(mod, nn.Module):
                    check_values(mod)
        if exists(prev_frames) and exists(self.frame_cond_encoder):
            self.vae.eval()
            prev_normalized = (prev_frames * 2.0 - 1.0).clamp(-1, 1)
            prev_latents, _ = self.vae.encode(prev_normalized)
            frame_cond_features = self.frame_cond_encoder(prev_latents * self.latent_scale)
            img = self.frame_cond_sample_loop(shape, frame_cond_features, frame_cond_scale)
        else:
            img = self.p_sample_loop(shape)
        self.vae.eval()
        samples = self.vae.decode(img / self.latent_scale)
        samples = samples.clamp(-1, 1)
        return (samples + 1) / 2

    def q_sample(self, x_start, t, noise=None):
        noise = default(noise, lambda: torch.randn_like(x_start))
        return (extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start +
                extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def p_losses(self, x_start, t, prev_frames=None, noise=None, null_frame_cond_prob=0.,
                 prob_focus_present=0., focus_present_mask=None):
        # reference: vdp.py:1099-1163.  Forward value only: the HIP denoiser has no backward yet.
        x_start_normalized = (x_start * 2.0 - 1.0).clamp(-1, 1)
        self.vae.eval()
        with torch.no_grad():
            x_start_latent, _ = self.vae.encode(x_start_normalized)
        x_start_latent = (x_start_latent * self.latent_scale).clamp(-10, 10)
        frame_cond_features = None
        if exists(prev_frames) and exists(self.frame_cond_encoder):
            with torch.no_grad():
                prev_latents, _ = self.vae.encode((prev_frames * 2.0 - 1.0).clamp(-1, 1))
            frame_cond_features = self.frame_cond_encoder(prev_latents * self.latent_scale)
        noise = default(noise, lambda: torch.randn_like(x_start_latent))
        x_noisy = self.q_sample(x_start=x_start_latent, t=t, noise=noise)
        predicted_noise = self.denoise_fn(x_noisy, t, frame_cond_features=frame_cond_features,
                                          null_frame_cond_prob=null_frame_cond_prob,
                                          prob_focus_present=prob_focus_present,
                                          focus_present_mask=focus_present_mask)
        if self.loss_type == 'l1':
            return F.l1_loss(noise, predicted_noise)
        elif self.loss_type == 'l2':
            return F.mse_loss(noise, predicted_noise)
        raise NotImplementedError()

    def forward(self, x, prev_frames=None, null_frame_cond_prob=0., *args, **kwargs):
        b, device = x.shape[0], x.device
        x = x.clamp(0, 1)
        t = torch.randint(0, self.num_timesteps, (b,), device=device).long()
        return self.p_losses(x, t, prev_frames=prev_frames, null_frame_cond_prob=null_frame_cond_prob,
                             *args, **kwargs)
