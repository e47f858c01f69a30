#!/usr/bin/env python3
"""Repeatability soak of the "hip" training step: the same weights and batch every step, so the loss must repeat bit for bit
(the forward kernels are deterministic) and so must every gradient: all sums across workgroups are partial results added
in a fixed order.  With VDMI_NORM_BWD_ATOMIC=1 (channel sums through float atomics) only the gradient norm repeats, to its
last digits."""
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
losses, norms, prints = [], [], []
names = [k for k, _ in unet.named_parameters()]
for i in range(n):
    for p in unet.parameters():
        p.grad = None
    loss = F.l1_loss(unet(x, t).float(), noise)
    loss.backward()
    losses.append(float(loss))
    norms.append(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in unet.parameters() if p.grad is not None))))
    assert all(bool(torch.isfinite(p.grad).all()) for p in unet.parameters() if p.grad is not None), i
    # the gradients' bits: one 64-bit sum of the raw words per parameter
    prints.append(tuple(int(p.grad.view(torch.int32).long().sum()) if p.grad is not None else 0 for p in unet.parameters()))
spread = (max(norms) - min(norms)) / norms[0]
print("steps", n, "distinct losses", len(set(losses)), "loss", losses[0], "grad norm", norms[0], "relative spread", spread)
moving = sorted({names[k] for fp in prints for k in range(len(names)) if fp[k] != prints[0][k]})
print("distinct gradient bit patterns", len(set(prints)), "parameters whose gradient bits moved:", len(moving), moving[:8])
atomic = os.environ.get("VDMI_NORM_BWD_ATOMIC", "0") == "1"
ok = len(set(losses)) == 1 and spread < 1e-5 and (atomic or len(set(prints)) == 1)
print("train step soak ok" if ok else "train step soak FAILED")
sys.exit(0 if ok else 1)
