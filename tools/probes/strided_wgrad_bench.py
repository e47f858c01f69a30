import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_diffusion_amd import train_hip as T
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for nf, h, w, cb, cs in [(16, 128, 128, 128, 128), (16, 64, 64, 256, 256), (16, 32, 32, 512, 512)]:
    big = torch.randn(nf * h * w, cb, generator=g).to(torch.bfloat16).to(dev)
    small = torch.randn(nf * h * w // 4, cs, generator=g).to(torch.bfloat16).to(dev)
    for _ in range(3):
        T.strided_wgrad(big, small, nf, h, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        T.strided_wgrad(big, small, nf, h, w)
    e1.record()
    torch.cuda.synchronize()
    print(f"strided_wgrad {cb}->{cs} big {h}x{w}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
