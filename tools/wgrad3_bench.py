#!/usr/bin/env python3
"""Time of the (1,3,3) weight gradient from rows (vdmi_conv_wgrad_rows + its reduce) at the layer shapes of the configs[4]
training step, HIP events over 20 calls each (kernel + partial-tile reduce + the workspace allocation, as the training step
calls it).  VDMI_LIB_PATH selects another build of the library for an A/B on one device."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from video_diffusion_amd import train_hip as T  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
out = []
for nf, h, w, cin, cout in [(16, 128, 128, 128, 128), (16, 128, 128, 256, 128), (16, 64, 64, 256, 256), (16, 32, 32, 512, 512),
                            (16, 16, 16, 1024, 1024), (16, 16, 16, 2048, 1024)]:
    x = torch.randn(nf * h * w, cin, generator=g).to(torch.bfloat16).to(dev)
    dy = torch.randn(nf * h * w, cout, generator=g).to(torch.bfloat16).to(dev)
    for _ in range(3):
        T.conv3x3_wgrad(x, dy, nf, h, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        T.conv3x3_wgrad(x, dy, nf, h, w)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    out.append(f"{cin}->{cout}@{h}x{w}: {us:6.1f} us ({2 * 9 * nf * h * w * cin * cout / us / 1e6:5.0f} TFLOP/s)")
print(os.environ.get("VDMI_LIB_PATH", "libvdmi.so").split("/")[-1], "|", " | ".join(out))
