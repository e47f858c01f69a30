#!/usr/bin/env python3
"""A/B of one (1,3,3) convolution layer on the streaming kernel (dconv) and on the general tap-GEMM.

    python tools/conv_ab.py --w 64 --h 64 --frames 16 --cin 256 --cout 256 [--norm]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=64)
    ap.add_argument("--h", type=int, default=64)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    from video_diffusion_amd import _lib, train_hip as T
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(args.frames * args.h * args.w, args.cin, generator=g).to(torch.bfloat16).to(dev)
    wt = (torch.randn(args.cout, args.cin, 1, 3, 3, generator=g) * 0.05).to(dev)
    pack = T.DevPack(wt, False)
    flops = 2.0 * args.frames * args.h * args.w * args.cin * args.cout * 9
    outs = {}
    for on in (0, 1, 0, 1):
        lib.vdmi_dconv_enable(on)
        for _ in range(3):
            y = T.conv_rows(x, pack, None, args.frames, args.frames, args.h, args.w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            y = T.conv_rows(x, pack, None, args.frames, args.frames, args.h, args.w)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        outs[on] = y
        print(f"dconv={on}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s (tile_bn {pack.tile_bn})")
    print("max abs diff between the two kernels:", float((outs[0].float() - outs[1].float()).abs().max()))


if __name__ == "__main__":
    main()
