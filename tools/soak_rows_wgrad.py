#!/usr/bin/env python3
"""Repeatability soak of vdmi_rows_wgrad (double-buffered LDS stages, partial tiles added in a fixed order): the same
operands must give the same bits every time, also with other work on a second stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from video_diffusion_amd import train_hip as T  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device=dev)
for rows, cin, cout in ((262144, 128, 768), (40000, 256, 768), (777, 72, 40), (65536, 256, 256)):
    x = torch.randn(rows, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(rows, cout, device=dev).to(torch.bfloat16)
    ref = T.rows_wgrad(x, dy).clone()
    for i in range(n):
        if i % 2:
            with torch.cuda.stream(side):
                junk = junk @ junk * 1e-4          # traffic and MFMA work beside the kernel
        if not torch.equal(T.rows_wgrad(x, dy), ref):
            bad += 1
    torch.cuda.synchronize()
    print(rows, cin, cout, "differing runs:", bad, flush=True)
print("rows_wgrad soak ok" if bad == 0 else "rows_wgrad soak FAILED")
sys.exit(1 if bad else 0)
