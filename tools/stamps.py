#!/usr/bin/env python3
"""Diagnostic (not shipped): phase shares of the tap-GEMM kernel from in-kernel cycle stamps.
Build the stamped library first:  make -C video_diffusion_amd/csrc diag
Run: VDMI_LIB_PATH=video_diffusion_amd/libvdmi_diag.so python tools/stamps.py --layer downs.0.1.block1"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="downs.0.1.block1")
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
dev = torch.device("cuda:0")
diag = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
os.environ["VDMI_DIAG_PTR"] = str(diag.data_ptr())
from video_diffusion_amd import Unet3D
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
d = diag.cpu().reshape(-1, 8)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
ph = (d[:, 1:5] - d[:, 0:4]).double()
names = ["prologue+patch load -> first barrier", "tap steps", "epilogue math", "stores"]
print(f"{args.layer}: {len(d)} workgroups, kernel span {(d[:,4].max()-t0).item()} cycles (readcyclecounter ticks)")
for i, n in enumerate(names):
    print(f"  {n:40s} mean {ph[:, i].mean():9.0f}  p50 {ph[:, i].median():9.0f}  max {ph[:, i].max():9.0f}")
print(f"  workgroup lifetime mean {(d[:,4]-d[:,0]).double().mean():.0f}")
print(f"  inside the tap steps (wave 0): waiting for the weight prefetch + LDS write {d[:,5].double().mean():.0f}, barrier wait {d[:,6].double().mean():.0f}, load issue + ds_read + MFMA section {d[:,7].double().mean():.0f}")
