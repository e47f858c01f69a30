"""Generate tests/golden/tiny_framecond.npz from the REAL reference (build container only).

    python -m oracle.make_golden_framecond

The reference's Unet3D is run with frame_cond_features from the reference's FrameConditioningEncoder:
conditioning kept, dropped (null_frame_cond_prob = 1) and dropped for one sample of the batch; and without
conditioning but with the focus-present mask set for one sample / for all.  Inputs
and outputs only; weights come from oracle.weightgen (Unet3D salt 0, encoder salt 11).
TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, weightgen  # noqa: E402
from oracle.make_golden import TINY, build_unet, save  # noqa: E402

X_SHAPE = (2, 3, 4, 12, 12)
PREV_SHAPE = (2, 3, 2, 12, 12)


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    vdp = ref_import.load_reference()
    m = build_unet(vdp, **TINY)
    enc = vdp.FrameConditioningEncoder(latent_dim=3, feature_dim=512, num_prev_frames=2)
    enc.load_state_dict(weightgen.fill_state_dict(enc.state_dict(), 11))
    enc.eval()
    x = weightgen.synthetic_input("fc.x", X_SHAPE)
    prev = weightgen.synthetic_input("fc.prev", PREV_SHAPE)
    t = torch.tensor([5, 640])
    feats = enc(prev)
    keep = torch.tensor([1.0, 0.0])
    save("tiny_framecond", x_shape=torch.tensor(X_SHAPE), prev_shape=torch.tensor(PREV_SHAPE), t=t, features=feats,
         eps_cond=m(x, t, frame_cond_features=feats, null_frame_cond_prob=0.),
         eps_null=m(x, t, frame_cond_features=feats, null_frame_cond_prob=1.),
         keep_mixed=keep, eps_mixed=m(x, t, frame_cond_features=feats * keep[:, None, None]),
         eps_plain=m(x, t),
         focus_mixed=torch.tensor([True, False]),
         eps_focus_mixed=m(x, t, focus_present_mask=torch.tensor([True, False])),
         eps_focus_all=m(x, t, prob_focus_present=1.))


if __name__ == "__main__":
    main()
