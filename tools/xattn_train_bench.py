#!/usr/bin/env python3
"""Time of the cross-attention training core (vdmi_xattn_core_train_fwd / _bwd) at one layer's shape.

    python tools/xattn_train_bench.py [--b 1] [--n 262144] [--l 1024] [--heads 8]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--l", type=int, default=1024)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    from video_diffusion_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    b, n, L, heads = a.b, a.n, a.l, a.heads
    hid = heads * 32
    g = torch.Generator().manual_seed(0)
    q = (torch.randn(b * n, hid, generator=g) * 0.8).to(torch.bfloat16).to(dev)
    kv = torch.randn(b * L, 2 * hid, generator=g).to(torch.bfloat16).to(dev)
    dout = torch.randn(b * n, hid, generator=g).to(torch.bfloat16).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    pack = torch.empty(2 * lib.vdmi_xattn_kv_elems(1, b, L, heads), device=dev, dtype=torch.bfloat16)
    _lib.check(lib.vdmi_xattn_pack_kv_train(kv.data_ptr(), 2 * hid, pack.data_ptr(), b, L, heads, s))
    npad = (n + 31) // 32 * 32
    o = torch.empty_like(q)
    lse = torch.zeros(b, heads, npad, device=dev)
    delta = torch.zeros(b, heads, npad, device=dev)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    ws = torch.empty(lib.vdmi_xattn_bwd_workspace_bytes(b, n, L, heads), device=dev, dtype=torch.uint8)

    def fwd():
        _lib.check(lib.vdmi_xattn_core_train_fwd(q.data_ptr(), hid, o.data_ptr(), pack.data_ptr(), lse.data_ptr(), npad, b, n, L, heads, s))

    def bwd():
        _lib.check(lib.vdmi_xattn_core_train_bwd(q.data_ptr(), hid, o.data_ptr(), dout.data_ptr(), pack.data_ptr(), kv.data_ptr(),
                                                 2 * hid, lse.data_ptr(), delta.data_ptr(), npad, dq.data_ptr(), dkv.data_ptr(),
                                                 2 * hid, b, n, L, heads, ws.data_ptr(), s))
    pairs = b * heads * ((n + 31) // 32) * ((L + 31) // 32)
    for name, fn, mfma in (("forward", fwd, 4), ("backward", bwd, 14)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f"{name}: {ms:.3f} ms, {pairs * mfma * 32768 / ms / 1e9:.0f} TFLOP/s of matrix work")


if __name__ == "__main__":
    main()
