#!/usr/bin/env python3
"""Does it pay to run the two sub-batch chains of the sampler out of step?

The sampler runs a batch as two independent sub-batch plans on two streams, joined after every forward (lockstep:
both parts are in the same layer at the same time).  Samples never interact, so the two chains could also run skewed
by half a forward: the deep, small-grid layers of one part would then run beside the wide first / last layers of the
other.  This probe times N forwards of both parts (a) in lockstep as one captured graph, (b) as two graphs replayed on
two streams with no join, part 1 started `--skew` of a forward early.

    python tools/stagger_probe.py [--steps 40] [--skew 0.5]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--skew", type=float, default=0.5)
    args = ap.parse_args()
    from video_diffusion_amd import Unet3D
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=1).to(dev).eval()
    eng = m.engine()
    plan = eng.plan(args.batch, 10, 64, 64, dev)
    assert hasattr(plan, "parts") and len(plan.parts) == 2
    plan.x_in.copy_(torch.randn_like(plan.x_in))
    plan.t_in.fill_(500)
    plan.run()
    torch.cuda.synchronize()
    p0, p1 = plan.parts
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def capture(fn, stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            fn()                                   # warm
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=stream):
                fn()
        torch.cuda.synchronize()
        return g

    g_lock = capture(lambda: plan.run(), s0)
    g0 = capture(lambda: p0.run(fork=False), s0)
    g1 = capture(lambda: p1.run(fork=False), s1)
    ncall = len(p1.calls)
    # the first `skew` of part 1's forward by time: launch durations are uneven, so split by measured time
    ev = []
    with torch.cuda.stream(s1):
        for c in p1.calls:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            c.fn(s1.cuda_stream)
            e1.record()
            ev.append((e0, e1))
    torch.cuda.synchronize()
    dur = [a.elapsed_time(b) for a, b in ev]
    acc, cut = 0.0, 0
    while cut < ncall and acc < args.skew * sum(dur):
        acc += dur[cut]
        cut += 1
    print(f"part 1: {ncall} launches, {sum(dur):.3f} ms serial; the first {cut} make {acc:.3f} ms")

    def head():
        for c in p1.calls[:cut]:
            c.fn(s1.cuda_stream)

    g1_head = capture(head, s1)

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    N = args.steps

    def lock():
        with torch.cuda.stream(s0):
            for _ in range(N):
                g_lock.replay()

    def free(skewed):
        if skewed:
            with torch.cuda.stream(s1):
                g1_head.replay()
        for _ in range(N):
            with torch.cuda.stream(s0):
                g0.replay()
            with torch.cuda.stream(s1):
                g1.replay()

    t_lock = timed(lock)
    t_free = timed(lambda: free(False))
    t_skew = timed(lambda: free(True))
    print(f"lockstep graph            : {t_lock / N * 1e3:.3f} ms per forward of {args.batch}")
    print(f"two graphs, no join       : {t_free / N * 1e3:.3f} ms")
    print(f"two graphs, part 1 skewed : {(t_skew) / N * 1e3:.3f} ms (includes the {acc:.2f} ms head once per {N})")


if __name__ == "__main__":
    main()
