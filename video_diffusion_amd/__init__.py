"""MI355X-native denoising hot path with the reference's Python surface.

Mirrors ``video_diffusion_pytorch/__init__.py:1`` of the reference: the same six names are
exported.  ``Unet3D`` runs on hand-written gfx950 HIP kernels (``libvdmi.so``); there is no
CPU or eager fallback for it.
"""
from .modules import Unet3D
from .diffusion import GaussianDiffusion
from .vae import VAE, FrameConditioningEncoder
from .trainer import Trainer, video_tensor_to_gif

__all__ = ["Unet3D", "GaussianDiffusion", "Trainer", "VAE", "FrameConditioningEncoder", "video_tensor_to_gif"]
