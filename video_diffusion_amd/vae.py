"""VAE and FrameConditioningEncoder with the reference's layers and state-dict keys
(vdp.py:83-269 of the reference).  They sit outside p_sample_loop (the VAE decodes once after the
loop, the frame encoder feeds the cross-attention path).  The decoder runs on the HIP library for GPU
tensors under no_grad (vae_engine.py); the encoder and the frame encoder are plain torch modules.
"""
import torch
from torch import nn


class VAEEncoder(nn.Module):
    def __init__(self, channels=3, latent_dim=4, base_dim=32):
        super().__init__()
        self.channels, self.latent_dim, self.base_dim = channels, latent_dim, base_dim
        self.encoder = nn.Sequential(
            nn.Conv3d(channels, base_dim, (1, 4, 4), (1, 2, 2), (0, 1, 1)), nn.GroupNorm(8, base_dim), nn.SiLU(),
            nn.Conv3d(base_dim, base_dim * 2, (1, 4, 4), (1, 2, 2), (0, 1, 1)), nn.GroupNorm(16, base_dim * 2), nn.SiLU(),
            nn.Conv3d(base_dim * 2, base_dim * 4, (1, 4, 4), (1, 2, 2), (0, 1, 1)), nn.GroupNorm(32, base_dim * 4), nn.SiLU(),
            nn.Conv3d(base_dim * 4, base_dim * 4, (1, 3, 3), padding=(0, 1, 1)), nn.GroupNorm(32, base_dim * 4), nn.SiLU(),
        )
        self.to_mean = nn.Conv3d(base_dim * 4, latent_dim, 1)
        self.to_logvar = nn.Conv3d(base_dim * 4, latent_dim, 1)

    def forward(self, x):
        h = self.encoder(x)
        return self.to_mean(h), torch.clamp(self.to_logvar(h), -30.0, 20.0)


class VAEDecoder(nn.Module):
    def __init__(self, latent_dim=4, channels=3, base_dim=64):
        super().__init__()
        self.latent_dim, self.channels = latent_dim, channels
        self.decoder = nn.Sequential(
            nn.Conv3d(latent_dim, base_dim * 4, 1), nn.GroupNorm(32, base_dim * 4), nn.SiLU(),
            nn.Conv3d(base_dim * 4, base_dim * 4, (1, 3, 3), padding=(0, 1, 1)), nn.GroupNorm(32, base_dim * 4), nn.SiLU(),
            nn.ConvTranspose3d(base_dim * 4, base_dim * 2, (1, 4, 4), (1, 2, 2), (0, 1, 1)), nn.GroupNorm(16, base_dim * 2), nn.SiLU(),
            nn.ConvTranspose3d(base_dim * 2, base_dim, (1, 4, 4), (1, 2, 2), (0, 1, 1)), nn.GroupNorm(8, base_dim), nn.SiLU(),
            nn.ConvTranspose3d(base_dim, channels, (1, 4, 4), (1, 2, 2), (0, 1, 1)),
        )

    def forward(self, z):
        # GPU tensors outside autograd: the HIP launch plan (vae_engine.py); it raises if libvdmi.so is missing.
        # CPU tensors and training keep the torch layers the parameters live in.
        if z.is_cuda and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            if getattr(self, "_hip_engine", None) is None:
                from .vae_engine import VaeDecodeEngine
                object.__setattr__(self, "_hip_engine", VaeDecodeEngine(self))
            return self._hip_engine.decode(z)
        return self.decoder(z)


class VAE(nn.Module):
    def __init__(self, channels=3, latent_dim=4, base_dim=64):
        super().__init__()
        self.encoder = VAEEncoder(channels, latent_dim, base_dim)
        self.decoder = VAEDecoder(latent_dim, channels, base_dim)
        self.latent_dim = latent_dim
        self.register_buffer('latent_scale', torch.ones(1))
        self.register_buffer('latent_shift', torch.zeros(1))

    def encode(self, x):
        return self.encoder(x)

    def decode(self, z):
        return self.decoder(z)

    def reparameterize(self, mean, logvar):
        std = torch.exp(0.5 * logvar)
        return mean + torch.randn_like(std) * std

    def encode_to_latent(self, x):
        mean, logvar = self.encode(x)
        return (self.reparameterize(mean, logvar) - self.latent_shift) / self.latent_scale

    def decode_from_latent(self, z_scaled):
        return self.decode(z_scaled * self.latent_scale + self.latent_shift)

    def forward(self, x):
        mean, logvar = self.encode(x)
        z = self.reparameterize(mean, logvar)
        return self.decode(z), mean, logvar


class FrameConditioningEncoder(nn.Module):
    def __init__(self, latent_dim=4, feature_dim=512, num_prev_frames=4):
        super().__init__()
        self.num_prev_frames = num_prev_frames
        self.feature_extractor = nn.Sequential(
            nn.Conv3d(latent_dim, 128, (1, 3, 3), padding=(0, 1, 1)), nn.SiLU(),
            nn.Conv3d(128, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1)), nn.SiLU(),
            nn.Conv3d(256, feature_dim, (1, 3, 3), (1, 2, 2), (0, 1, 1)), nn.SiLU(),
        )

    def forward(self, prev_latents):
        features = self.feature_extractor(prev_latents)
        b, c, f, h, w = features.shape
        return features.permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c)
