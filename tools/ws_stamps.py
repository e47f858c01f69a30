#!/usr/bin/env python3
"""Diagnostic (not shipped): where the waves of the weight-stationary conv kernel spend their cycles.
Build the stamped library first:  make -C video_diffusion_amd/csrc diag
Run: VDMI_LIB_PATH=video_diffusion_amd/libvdmi_diag.so python tools/ws_stamps.py --layer downs.0.1.block1"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="downs.0.1.block1")
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
dev = torch.device("cuda:0")
diag = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
os.environ["VDMI_DIAG_PTR"] = str(diag.data_ptr())
from video_diffusion_amd import Unet3D
from video_diffusion_amd import engine as E
E.BATCH_PARTS = 1
torch.manual_seed(0)
m = Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=1).to(dev).eval()
plan = m.engine().plan(args.batch, 10, 64, 64, dev)
plan.x_in.copy_(torch.randn_like(plan.x_in)); plan.t_in.fill_(500)
plan.run(); torch.cuda.synchronize()
calls = [c for c in plan.calls if c.name.startswith(args.layer)]
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    diag.zero_()
    for c in calls: c.fn(s)
    torch.cuda.synchronize()
d = diag.cpu().reshape(-1, 8, 8)
d = d[d[:, 0, 3] != 0].double()
names = ["issue patch DMA", "epilogue: residual read, partner wait, rinv", "issue residual + init", "MFMA phase",
         "publish ss + wait DMA", "epilogue: math + staging writes", "barrier wait", "epilogue: stores"]
print(f"{args.layer}: {d.shape[0]} workgroups")
for role, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    print(" ", role)
    for i, n in enumerate(names):
        x = d[:, sl, i]
        print(f"    {n:44s} mean {x.mean():9.0f}  min {x.min():9.0f}  max {x.max():9.0f}")
    print(f"    {'sum':44s} mean {d[:, sl, :].sum(-1).mean():9.0f}")
