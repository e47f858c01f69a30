#!/usr/bin/env python3
"""Diagnostic (not shipped): where a workgroup of the streaming conv (csrc/dconv.hip) spends its cycles.
Build the stamped library first:  make -C video_diffusion_amd/csrc diag
Run: VDMI_LIB_PATH=video_diffusion_amd/libvdmi_diag.so python tools/dconv_stamps.py --layer mid_block1.block1.conv"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="mid_block1.block1.conv")
ap.add_argument("--batch", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda:0")
diag = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
os.environ["VDMI_DIAG_PTR"] = str(diag.data_ptr())
os.environ["VDMI_BATCH_PARTS"] = "1"
from video_diffusion_amd import Unet3D
torch.manual_seed(0)
m = Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=1).to(dev).eval()
plan = m.engine().plan(args.batch, 10, 64, 64, dev)
plan.x_in.copy_(torch.randn_like(plan.x_in)); plan.t_in.fill_(500)
plan.run(); torch.cuda.synchronize()
calls = [c for c in plan.calls if c.name == args.layer]
assert calls, [c.name for c in plan.calls]
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    diag.zero_()
    for c in calls: c.fn(s)
    torch.cuda.synchronize()
d = diag.cpu().reshape(-1, 8)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
names = ["set-up + request issue", "first patch + tile landed (wait + barrier)", "main loop", "drain + barrier", "epilogue + stores"]
print(f"{args.layer}: {len(d)} workgroups; first start -> last end {(d[:, 5].max() - t0).item()} ticks; "
      f"starts spread over {(d[:, 0].max() - t0).item()} ticks")
ph = (d[:, 1:6] - d[:, 0:5]).double()
for i, n in enumerate(names):
    print(f"  {n:46s} mean {ph[:, i].mean():9.0f}  p50 {ph[:, i].median():9.0f}  max {ph[:, i].max():9.0f}")
print(f"  workgroup lifetime mean {(d[:, 5] - d[:, 0]).double().mean():.0f}")
