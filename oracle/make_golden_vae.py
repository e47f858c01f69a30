"""Generate tests/golden/vae_tiny.npz from the REAL reference VAE (build container only).

    python -m oracle.make_golden_vae

Inputs and the reference's outputs only; weights come from oracle.weightgen (salt 7) as everywhere.
TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, weightgen  # noqa: E402
from oracle.make_golden import save  # noqa: E402

VAE_KW = dict(channels=3, latent_dim=4, base_dim=64)
Z_SHAPE = (2, 4, 3, 6, 6)
X_SHAPE = (2, 3, 3, 48, 48)


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    vdp = ref_import.load_reference()
    vae = vdp.VAE(**VAE_KW)
    vae.load_state_dict(weightgen.fill_state_dict(vae.state_dict(), 7))
    vae.eval()
    z = weightgen.synthetic_input("vae.z", Z_SHAPE)
    x = weightgen.synthetic_input("vae.x", X_SHAPE)
    mean, logvar = vae.encode(x)
    save("vae_tiny", z_shape=torch.tensor(Z_SHAPE), x_shape=torch.tensor(X_SHAPE), decode=vae.decode(z),
         enc_mean=mean, enc_logvar=logvar)


if __name__ == "__main__":
    main()
