#!/usr/bin/env python3
"""Repeatability soak of the "hip" training step: the same weights and batch every step, so the loss must repeat bit for bit
(the forward kernels are deterministic) and the gradient norm to the last digits (its channel sums use float atomics)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from video_diffusion_amd import Unet3D  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
unet = Unet3D(dim=128, dim_mults=(1, 2, 4, 8), channels=3).to(dev)
unet.train_backend = "hip"
x = torch.randn(1, 3, 16, 128, 128, device=dev)
noise = torch.randn_like(x)
t = torch.randint(0, 1000, (1,), device=dev)
losses, norms = [], []
for i in range(n):
    for p in unet.parameters():
        p.grad = None
    loss = F.l1_loss(unet(x, t).float(), noise)
    loss.backward()
    losses.append(float(loss))
    norms.append(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in unet.parameters() if p.grad is not None))))
    assert all(bool(torch.isfinite(p.grad).all()) for p in unet.parameters() if p.grad is not None), i
spread = (max(norms) - min(norms)) / norms[0]
print("steps", n, "distinct losses", len(set(losses)), "loss", losses[0], "grad norm", norms[0], "relative spread", spread)
ok = len(set(losses)) == 1 and spread < 1e-5
print("train step soak ok" if ok else "train step soak FAILED")
sys.exit(0 if ok else 1)
