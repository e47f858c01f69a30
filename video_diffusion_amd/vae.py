"""VAE and FrameConditioningEncoder with the reference's layers and state-dict keys
(vdp.py:83-269 of the reference).  They sit outside p_sample_loop (the VAE decodes once after the
loop, the encoder and the frame encoder feed the cross-attention path).  For GPU tensors outside autograd all
three run on the HIP library (vae_engine.py); the torch layers hold the parameters and serve CPU / training.
"""
import torch
from torch import nn


def _use_hip(module, x):
    """GPU tensors outside autograd run on the HIP launch plans (vae_engine.py; they raise if libvdmi.so is
    missing).  CPU tensors and training (the VAE's own trainer, the frame encoder inside p_losses) keep the
    torch layers the parameters live in."""
    return x.is_cuda and not (torch.is_grad_enabled() and
                              (x.requires_grad or any(p.requires_grad for p in module.parameters())))


def weights_changed(module):
    """Tell the HIP plans of `module` and of its sub-modules that weights were rewritten through `.data` (in-place writes
    that neither move the tensor nor advance its version counter); everything else is noticed without this."""
    for m in module.modules():
        eng = getattr(m, "_hip_engine", None)
        if eng is not None:
            eng.invalidate()


def check_values(module):
    """Once per sampling call: compare the value fingerprints of the HIP plans under `module` with the live weights."""
    for m in module.modules():
        eng = getattr(m, "_hip_engine", None)
        if eng is not None:
            eng.check_values()


def _engine(module, cls_name):
    if getattr(module, "_hip_engine", None) is None:
        from . import vae_engine
        object.__setattr__(module, "_hip_engine", getattr(vae_engine, cls_name)(module))
    return module._hip_engine


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
        if _use_hip(self, x):
            return _engine(self, "VaeEncodeEngine").encode(x)
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
        if _use_hip(self, z):
            return _engine(self, "VaeDecodeEngine").decode(z)
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

    def weights_changed(self):
        weights_changed(self)

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
        if _use_hip(self, prev_latents):
            return _engine(self, "FrameEncodeEngine").encode(prev_latents)
        features = self.feature_extractor(prev_latents)
        b, c, f, h, w = features.shape
        return features.permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c)
