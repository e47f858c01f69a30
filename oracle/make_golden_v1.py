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


@torch.no_grad()
def main():
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
    save("v1_tiny", x_shape=torch.tensor(X_SHAPE), t=t, eps=m(x, t),
         eps_focus=m(x, t, focus_present_mask=torch.tensor([False, True])))


if __name__ == "__main__":
    main()
