#!/usr/bin/env python3
"""Kernel times of one reconstruction-guidance gradient of the version1 variant (torch profiler), by kernel name."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from video_diffusion_amd import VAE, version1 as V1  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, F, S, ov = 2, 10, 32, 2
gd = V1.GaussianDiffusion(V1.Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=4).eval(), VAE(channels=3, latent_dim=4, base_dim=64).eval(),
                          image_size=8 * S, num_frames=F, channels=3, timesteps=1000).to(dev)
z = torch.randn(B, 4, F, S, S, device=dev)
t = torch.full((B,), 500, device=dev, dtype=torch.long)
fn = V1.temporal_consistency_guidance(gd, torch.rand(B, 3, ov, 8 * S, 8 * S, device=dev), ov)
for _ in range(2):
    fn(z, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    fn(z, t)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:25]
print(json.dumps({e.key[:100]: [round(e.device_time_total / 1e3, 3), e.count] for e in rows}, indent=0))
