"""Generate tests/golden/guidance_tiny.npz from the reference's video_diffusion_guidance.py (build container only).

    python -m oracle.make_golden_guidance

That file's Unet3D (depthwise temporal conv + GroupNorm blocks, temporal then spatial attention) with
oracle.weightgen weights (salt 4); inputs and outputs only.  TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, weightgen  # noqa: E402
from oracle.make_golden import save  # noqa: E402

KW = dict(dim=64, dim_mults=(1, 2), channels=4)
X_SHAPE = (2, 4, 3, 8, 8)


def main():
    torch.set_num_threads(8)
    ref_import.load_reference()
    import video_diffusion_guidance as vdg
    m = vdg.Unet3D(**KW)
    m.load_state_dict(weightgen.fill_state_dict(m.state_dict(), 4))
    m.eval()
    x = weightgen.synthetic_input("g.x", X_SHAPE)
    t = torch.tensor([12, 640])
    with torch.no_grad():
        save("guidance_tiny", x_shape=torch.tensor(X_SHAPE), t=t, eps=m(x, t))


if __name__ == "__main__":
    main()
