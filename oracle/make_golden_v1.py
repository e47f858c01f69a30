"""Generate tests/golden/v1_tiny.npz from the reference's version1diffusion.py (build container only).

    python -m oracle.make_golden_v1

Unet3D of that file (temporal conv in every ResnetBlock, SpatioTemporalAttention, max_distance 300) with
oracle.weightgen weights (salt 3); inputs and outputs only.  TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, weightgen  # noqa: E402
from oracle.make_golden import save  # noqa: E402

KW = dict(dim=32, dim_mults=(1, 2), channels=3)
X_SHAPE = (2, 3, 4, 12, 12)
G_SHAPE = (2, 4, 3, 4, 4)


def main():
  with torch.no_grad():
    _main()


def _main():
    torch.set_num_threads(8)
    ref_import.load_reference()                       # installs the import stubs and the path
    import video_diffusion_guidance as vdg
    import version1diffusion as v1
    v1.RotaryEmbedding = vdg.RotaryEmbedding
    m = v1.Unet3D(**KW)
    m.load_state_dict(weightgen.fill_state_dict(m.state_dict(), 3))
    m.eval()
    x = weightgen.synthetic_input("v1.x", X_SHAPE)
    t = torch.tensor([4, 871])
    # reconstruction guidance (version1diffusion.py:707-747) on a 4-channel latent model + the file's VAE
    m4 = v1.Unet3D(dim=32, dim_mults=(1, 2), channels=4)
    m4.load_state_dict(weightgen.fill_state_dict(m4.state_dict(), 3))
    vae = v1.VAE(channels=3, latent_dim=4, base_dim=64)
    vae.load_state_dict(weightgen.fill_state_dict(vae.state_dict(), 7))
    gd = v1.GaussianDiffusion(m4.eval(), vae.eval(), image_size=32, num_frames=3, channels=3, timesteps=250)
    z_t = weightgen.synthetic_input("v1.zt", G_SHAPE)
    prev = weightgen.synthetic_input("v1.prevpix", (2, 3, 2, 32, 32)).mul(0.25).add(0.5).clamp(0, 1)
    tg = torch.tensor([200, 40])
    with torch.enable_grad():
        guid = v1.temporal_consistency_guidance(gd, prev, 2, w_r=3.0)(z_t, tg)
    save("v1_tiny", x_shape=torch.tensor(X_SHAPE), t=t, eps=m(x, t),
         eps_focus=m(x, t, focus_present_mask=torch.tensor([False, True])),
         g_shape=torch.tensor(G_SHAPE), g_t=tg, guidance=guid)


if __name__ == "__main__":
    main()
