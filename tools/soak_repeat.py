#!/usr/bin/env python3
"""Repeatability soak: replay the captured denoise step N times on fixed inputs and compare every activation
buffer bit for bit with the first run (catches races between the plan's main and side streams).
    python tools/soak_repeat.py [N] [batch] [plain|cond]        cond: condition vector, frame context and focus flags on"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from video_diffusion_amd import Unet3D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
cond = len(sys.argv) > 3 and sys.argv[3] == "cond"
if cond:
    m = Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=4, cond_dim=64).to(dev).eval()
    plan = m.engine().plan(batch, 10, 32, 32, dev, L=256, focus=True)
    plan.cond_in.copy_(torch.randn_like(plan.cond_in))
    plan.null_mask.copy_((torch.arange(batch, device=dev) % 3 == 0).to(torch.uint8))
    plan.focus_in.copy_((torch.arange(batch, device=dev) % 2).to(torch.uint8))
    plan.ctx_in.copy_(torch.randn_like(plan.ctx_in))
    plan.ctx_keep.copy_((torch.arange(batch, device=dev) % 4 != 1).float())
    plan.run_context()
else:
    m = Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=1).to(dev).eval()
    plan = m.engine().plan(batch, 10, 64, 64, dev)
plan.x_in.copy_(torch.randn_like(plan.x_in))
plan.t_in.copy_(torch.arange(batch, device=dev) * (999 // max(1, batch - 1)))
plan.run()
torch.cuda.synchronize()
plans = getattr(plan, 'parts', [plan])
bufs = [k for p in plans for k in p.keep if isinstance(k, torch.Tensor)] + [plan.eps_out]
print('sub-plans:', len(plans))
ref = [b.clone() for b in bufs]
bad = 0
for _ in range(n):
    plan.run()
    torch.cuda.synchronize()
    bad += not all(torch.equal(a, b) for a, b in zip(ref, bufs))
print(f"eager: {bad} of {n} runs differ")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    plan.run()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    plan.run()
gbad = 0
for _ in range(n):
    g.replay()
    torch.cuda.synchronize()
    gbad += not all(torch.equal(a, b) for a, b in zip(ref, bufs))
print(f"graph: {gbad} of {n} replays differ")
sys.exit(1 if bad or gbad else 0)
