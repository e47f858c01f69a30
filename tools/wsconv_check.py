#!/usr/bin/env python3
"""A/B of the weight-stationary conv kernel against the general tap-GEMM, launch by launch, inside a real plan.

    python tools/wsconv_check.py [--batch 16] [--size 64] [--iters 20] [--rounds 3]

For every tap-GEMM launch of the plan that the weight-stationary kernel takes: run it once on each kernel from the
same inputs and compare the output rows (both are bf16 results of the same fp32-accumulated sums, so they agree to
rounding), then time both in alternating rounds with HIP events.
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--parts", type=int, default=1)
    ap.add_argument("--new", type=int, default=1, help="kernel generation to test (vdmi_wsconv_enable value)")
    args = ap.parse_args()
    from video_diffusion_amd import Unet3D, _lib
    from video_diffusion_amd import engine as E
    E.BATCH_PARTS = args.parts
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=args.channels).to(dev).eval()
    # non-trivial norm gammas / biases so that every epilogue term is exercised
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("gamma"):
                p.mul_(1.0 + 0.2 * torch.randn_like(p))
    eng = m.engine()
    plan = eng.plan(args.batch, args.frames, args.size, args.size, dev)
    plan.x_in.copy_(torch.randn_like(plan.x_in))
    plan.t_in.copy_(torch.randint(0, 1000, plan.t_in.shape, device=dev))
    lib.vdmi_wsconv_enable(0)
    plan.run()
    torch.cuda.synchronize()
    eps_old = plan.eps_out.clone()

    bufs = {}
    for p in (plan.parts if hasattr(plan, "parts") else [plan]):
        for t in p.keep:
            if isinstance(t, torch.Tensor):
                bufs[t.data_ptr()] = t
    s = torch.cuda.current_stream().cuda_stream
    cands = []
    for c in plan.calls:
        meta = c.args or {}
        if meta.get("kernel", "") in ("wsconv64", "tapgemm<epi=1,bn=64>"):
            cands.append(c)
    print(f"{len(cands)} candidate launches")
    worst = 0.0
    taken = []
    for c in cands:
        a = c.fn.__defaults__[0] if c.fn.__defaults__ else None
        out = bufs.get(a.out) if a is not None else None
        if out is None:
            continue
        lib.vdmi_wsconv_enable(0)
        assert c.fn(s) == 0
        torch.cuda.synchronize()
        ref = out.float().clone()
        out.zero_()
        lib.vdmi_wsconv_enable(args.new)
        assert c.fn(s) == 0
        torch.cuda.synchronize()
        got = out.float()
        err = float((got - ref).norm() / ref.norm())
        mx = float((got - ref).abs().max())
        bad = int(((got - ref).abs() > 0.05 * ref.abs().clamp_min(0.5)).sum())
        worst = max(worst, err)

        def timeit():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                c.fn(s)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3
        t_old, t_new = [], []
        for _ in range(args.rounds):
            lib.vdmi_wsconv_enable(0)
            t_old.append(timeit())
            lib.vdmi_wsconv_enable(args.new)
            t_new.append(timeit())
        fl = (c.args or {}).get("flops", 0.0)
        same = abs(min(t_old) - min(t_new)) < 0.02 * min(t_old)
        print(f"{c.name:28s} rel {err:.2e} max {mx:.3f} bad {bad:6d}  old {min(t_old):7.1f} us ({fl / min(t_old) / 1e6:6.0f} TF)"
              f"  new {min(t_new):7.1f} us ({fl / min(t_new) / 1e6:6.0f} TF){'  [same kernel?]' if same else ''}")
        taken.append((min(t_old), min(t_new)))
    lib.vdmi_wsconv_enable(args.new)
    plan.run()
    torch.cuda.synchronize()
    e = float((plan.eps_out - eps_old).norm() / eps_old.norm())
    print(f"eps of the whole forward, new vs old kernels: rel {e:.3e}; worst single launch rel {worst:.3e}")
    print(f"sum over launches: old {sum(t[0] for t in taken):.1f} us, new {sum(t[1] for t in taken):.1f} us")
    assert worst < 5e-3, worst


if __name__ == "__main__":
    main()
