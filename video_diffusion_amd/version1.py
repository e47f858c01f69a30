"""The reference's `version1diffusion.py` variant of the stack on the same HIP kernels.

Differences from the packaged model (reference file `version1diffusion.py`):
* ResnetBlock ends with a (3,1,1) temporal convolution before the residual add (:301-323);
* the spatial linear attention slot holds SpatioTemporalAttention: full softmax attention over all
  f*h*w tokens of a sample, 32-dim heads, no bias, no rotary (:325-345);
* RelativePositionBias uses max_distance = 300 (:406); there is no text or frame conditioning;
* p_sample takes an `injection_fn` / `guidance_fn` applied to x_t before the step (:572-592), and
  `sample()` returns (samples, latents) (:610-624).

Class names, constructor signatures and state-dict keys follow that file.  The temporal convolution is a
3-tap vdmi_tapgemm over the (frame, position) view of the rows; the full attention reuses the
cross-attention core (vdmi_xattn_pack_kv / vdmi_xattn_core) with the layer's own keys and values.
Reconstruction guidance (`temporal_consistency_guidance`, :707-747) differentiates through the denoiser and
the VAE decoder: PyTorch autograd, through the HIP training kernels where the denoiser's shapes allow
(train_hip.supports), else through torch ops.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import diffusion as _diffusion
from . import modules as _modules
from .diffusion import extract
from .modules import PreNorm, RelativePositionBias, Residual, ResnetBlock, _HipOnly


class SpatioTemporalAttention(_HipOnly):
    # reference: version1diffusion.py:325-345
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        hidden_dim = dim_head * heads
        self.to_qkv = nn.Linear(dim, hidden_dim * 3, bias=False)
        self.to_out = nn.Linear(hidden_dim, dim, bias=False)


class Unet3D(_modules.Unet3D):
    """version1diffusion.py:400-482."""

    def __init__(self, dim, out_dim=None, dim_mults=(1, 2, 4, 8), channels=4, attn_heads=8, attn_dim_head=32,
                 init_dim=None, init_kernel_size=7, use_sparse_linear_attn=True):
        super().__init__(dim, out_dim=out_dim, dim_mults=dim_mults, channels=channels, attn_heads=attn_heads,
                         attn_dim_head=attn_dim_head, init_dim=init_dim, init_kernel_size=init_kernel_size,
                         use_sparse_linear_attn=use_sparse_linear_attn, frame_cond_dim=None)
        self.time_rel_pos_bias = RelativePositionBias(heads=attn_heads, max_distance=300)
        for m in list(self.modules()):
            if isinstance(m, ResnetBlock):
                c = m.block2.proj.out_channels
                m.temporal_conv = nn.Conv3d(c, c, kernel_size=(3, 1, 1), padding=(1, 0, 0))
        if use_sparse_linear_attn:
            for stage in list(self.downs) + list(self.ups):
                c = stage[1].block2.proj.out_channels
                stage[2] = Residual(PreNorm(c, SpatioTemporalAttention(c, heads=attn_heads)))

    def forward(self, x, time, focus_present_mask=None):
        return super().forward(x, time, focus_present_mask=focus_present_mask)


class GaussianDiffusion(_diffusion.GaussianDiffusion):
    """version1diffusion.py:497-655."""

    def __init__(self, denoise_fn, vae, image_size, num_frames, channels=3, timesteps=1000, loss_type='l1',
                 use_dynamic_thres=False, dynamic_thres_percentile=0.9, latent_scale=1.0, injection_steps=600):
        super().__init__(denoise_fn, vae, image_size, num_frames, channels=channels, timesteps=timesteps,
                         loss_type=loss_type, use_dynamic_thres=use_dynamic_thres,
                         dynamic_thres_percentile=dynamic_thres_percentile, latent_scale=latent_scale)
        self.injection_steps = injection_steps

    def p_sample(self, x, t, clip_denoised=True, guidance_fn=None, guidance_scale=1.0, injection_fn=None, noise=None):
        # :572-592 — injection wins over guidance; both act on x_t before the denoiser sees it.  Not under
        # inference_mode: a guidance_fn differentiates with respect to x_t.
        with torch.no_grad():
            if injection_fn is not None:
                x = injection_fn(x.clone(), t)
            elif guidance_fn is not None:
                grad = guidance_fn(x, t)
                x = x + guidance_scale * extract(self.betas, t, x.shape) * grad
        return super().p_sample(x, t, clip_denoised=clip_denoised, noise=noise)

    def p_sample_loop(self, shape, guidance_fn=None, guidance_scale=1.0, injection_fn=None, x_T=None, noises=None):
        # :594-608
        if guidance_fn is None and injection_fn is None:
            return super().p_sample_loop(shape, x_T=x_T, noises=noises)
        device = self.betas.device
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        for k, i in enumerate(reversed(range(0, self.num_timesteps))):
            t = torch.full((shape[0],), i, device=device, dtype=torch.long)
            img = self.p_sample(img, t, guidance_fn=guidance_fn, guidance_scale=guidance_scale, injection_fn=injection_fn,
                                noise=None if noises is None else noises[k].to(device))
        return img

    @torch.inference_mode()
    def sample(self, batch_size=16, guidance_fn=None, guidance_scale=1.0, injection_fn=None):
        # :610-624
        latent_size = self.image_size // 8
        latents = self.p_sample_loop((batch_size, self.vae.latent_dim, self.num_frames, latent_size, latent_size),
                                     guidance_fn=guidance_fn, guidance_scale=guidance_scale, injection_fn=injection_fn)
        self.vae.eval()
        samples = self.vae.decode(latents / self.latent_scale)
        return (samples.clamp(-1, 1) + 1) / 2, latents

    def _predict_noise(self, x, t, frame_cond_features=None):
        return self.denoise_fn(x, t)                       # :557

    def p_losses(self, x_start, t, noise=None):
        # :633-649 (differentiable when the denoiser's train_backend is set, as for the packaged model)
        x_start_normalized = (x_start * 2.0 - 1.0).clamp(-1, 1)
        self.vae.eval()
        with torch.no_grad():
            x_start_latent, _ = self.vae.encode(x_start_normalized)
        x_start_latent = (x_start_latent * self.latent_scale).clamp(-10, 10)
        noise = torch.randn_like(x_start_latent) if noise is None else noise
        predicted_noise = self.denoise_fn(self.q_sample(x_start=x_start_latent, t=t, noise=noise), t)
        if self.loss_type == 'l1':
            return nn.functional.l1_loss(noise, predicted_noise)
        elif self.loss_type == 'l2':
            return nn.functional.mse_loss(noise, predicted_noise)
        raise NotImplementedError()

    def forward(self, x, *args, **kwargs):
        x = x.clamp(0, 1)
        t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=x.device).long()
        return self.p_losses(x, t, *args, **kwargs)


def temporal_consistency_injection(model, prev_latents, overlap_frames, save_gif=True, gif_dir="./injection_gifs"):
    """version1diffusion.py:656-706: during the first `model.injection_steps` steps the first `overlap_frames`
    frames of x_t are replaced by the previous chunk's last frames noised to the current timestep."""
    if save_gif:
        os.makedirs(gif_dir, exist_ok=True)

    def injection_fn(z_t, t):
        if prev_latents is None:
            return z_t
        t0 = int(t[0].item())
        if t0 < model.num_timesteps - model.injection_steps:
            return z_t
        prev_overlap = prev_latents[:, :, -overlap_frames:]
        noised = (extract(model.sqrt_alphas_cumprod, t, prev_overlap.shape) * prev_overlap +
                  extract(model.sqrt_one_minus_alphas_cumprod, t, prev_overlap.shape) * torch.randn_like(prev_overlap))
        z_t[:, :, :overlap_frames] = noised
        if save_gif and t0 % 100 == 0:
            from .trainer import video_tensor_to_gif
            video = (model.vae.decode(z_t / model.latent_scale).clamp(-1, 1) + 1) / 2
            video_tensor_to_gif(video[0], os.path.join(gif_dir, f"step_{t0:04d}.gif"), duration=100)
        return z_t

    return injection_fn


def temporal_consistency_guidance(model, prev_frames, overlap_frames, w_r=1.0):
    """version1diffusion.py:707-747: -w_r * alphas_cumprod_t * d/dz_t mse(decode(x0_hat(z_t))[first frames], prev_frames).
    The derivative runs through the denoiser and the VAE decoder.  On the GPU, when the denoiser's shapes allow it
    (train_hip.supports), its Blocks, temporal attention layers and resampling convolutions run forward and backward on the
    HIP kernels (train_backend "hip": only the input gradient is asked for, so no weight gradient is computed); the
    remaining layers and the VAE decoder are PyTorch autograd ops.  The step that follows runs on the HIP kernels."""
    for param in model.vae.decoder.parameters():
        param.requires_grad = False

    from . import train_hip as _th0
    packs = _th0.PackStore()    # packed weight sets of the HIP training layers, kept over the steps of the loop (weights are frozen)

    def guidance_fn(z_t, t):
        if prev_frames is None:
            return torch.zeros_like(z_t)
        fn = model.denoise_fn
        z = z_t.detach().clone().requires_grad_(True)
        backend = getattr(fn, "train_backend", None)
        # only d/dz_t is wanted: with the parameters frozen for the duration no layer computes a weight gradient
        # (autograd asks a Function for every input that requires grad, whatever the caller differentiates)
        frozen = [p for p in fn.parameters() if p.requires_grad]
        with torch.enable_grad():
            try:
                for p in frozen:
                    p.requires_grad_(False)
                if hasattr(fn, "train_backend"):
                    from . import train_hip
                    own = getattr(fn, "hip_training_supported", None)     # the guidance variant answers for its own layers
                    fits = own() if own is not None else train_hip.supports(fn)
                    fn.train_backend = "hip" if (z.is_cuda and fits) else "torch-autograd"
                from . import train_hip as _th
                with _th.fixed_weights(packs):
                    x_recon = model.predict_start_from_noise(z, t, fn(z, t))
                    overlap = x_recon[:, :, :overlap_frames] if x_recon.dim() == 5 else x_recon
                    pixels = (model.vae.decode(overlap / model.latent_scale).clamp(-1, 1) + 1) / 2
                    loss = nn.functional.mse_loss(pixels, prev_frames, reduction='mean')
                    grad = torch.autograd.grad(loss, z)[0]
            finally:
                for p in frozen:
                    p.requires_grad_(True)
                if hasattr(fn, "train_backend"):
                    fn.train_backend = backend
        return (-w_r * extract(model.alphas_cumprod, t, z_t.shape) * grad).detach()

    return guidance_fn
