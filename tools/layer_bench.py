#!/usr/bin/env python3
"""Run one layer of the Unet3D plan in a loop (for rocprofv3 counter passes and A/B timing).

    python tools/layer_bench.py --layer downs.0.1 --iters 20
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="downs.0.1", help="prefix of the plan call names to run")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--all", action="store_true", help="time every launch of the plan on its own and print a table")
    args = ap.parse_args()
    from video_diffusion_amd import Unet3D
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Unet3D(dim=args.dim, dim_mults=(1, 2, 4, 8), channels=args.channels).to(dev).eval()
    eng = m.engine()
    plan = eng.plan(args.batch, args.frames, args.size, args.size, dev)
    plan.x_in.copy_(torch.randn_like(plan.x_in))
    plan.t_in.fill_(500)
    plan.run()                                  # fills every buffer with realistic data
    torch.cuda.synchronize()
    if args.all:
        s = torch.cuda.current_stream().cuda_stream
        total = 0.0
        for c in plan.calls:
            c.fn(s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                c.fn(s)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            total += ms
            meta = c.args or {}
            fl = meta.get("flops", 0.0)
            print(f"{c.name:34s} {meta.get('kernel', '-'):24s} {ms * 1e3:8.1f} us {fl / ms / 1e9 if fl else 0:7.1f} TFLOP/s "
                  f"{meta.get('bytes', 0.0) / ms / 1e6:7.0f} GB/s")
        print(f"sum {total:.3f} ms")
        return
    calls = [c for c in plan.calls if c.name.startswith(args.layer)]
    assert calls, [c.name for c in plan.calls]
    s = torch.cuda.current_stream().cuda_stream
    for c in calls:
        c.fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        for c in calls:
            assert c.fn(s) == 0
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    flops = sum((c.args or {}).get("flops", 0.0) for c in calls)
    print(f"{args.layer}: {len(calls)} launches, {ms * 1e3:.1f} us per pass, {flops / ms / 1e9:.1f} TFLOP/s "
          f"[{', '.join(c.name for c in calls)}]")


if __name__ == "__main__":
    main()
