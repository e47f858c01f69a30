#!/usr/bin/env python3
"""Time of the temporal attention training core (vdmi_tattn_core_fwd / _bwd) at one layer's shape, on the matrix-core
form and on the vector kernels, with the bytes each call has to move.

    python tools/tattn_core_bench.py [--b 1] [--f 16] [--hw 16384] [--heads 8]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--f", type=int, default=16)
    ap.add_argument("--hw", type=int, default=16384)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--forms", default="mfma,vector")
    a = ap.parse_args()
    from video_diffusion_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    b, f, hw, heads = a.b, a.f, a.hw, a.heads
    rows, hid = b * f * hw, heads * 32
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(rows, 3 * hid, generator=g).to(torch.bfloat16).to(dev)
    dout = torch.randn(rows, hid, generator=g).to(torch.bfloat16).to(dev)
    bias = (0.5 * torch.randn(heads, f, f, generator=g)).to(dev)
    ang = torch.arange(f).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))[None, :]
    cos, sin = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
    o = torch.empty(rows, hid, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(rows, heads, device=dev)
    dqkv = torch.empty_like(qkv)
    dbias = torch.zeros_like(bias)
    s = torch.cuda.current_stream().cuda_stream
    scale = 32 ** -0.5
    for form in a.forms.split(","):
        lib.vdmi_tattn_core_mfma_enable(1 if form == "mfma" else 0)
        ws = torch.empty(max(16, lib.vdmi_tattn_core_bwd_workspace_bytes(b, f, hw, heads)), device=dev, dtype=torch.uint8)

        def fwd():
            _lib.check(lib.vdmi_tattn_core_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), bias.data_ptr(), cos.data_ptr(),
                                               sin.data_ptr(), None, b, f, hw, heads, scale, s))

        def bwd():
            _lib.check(lib.vdmi_tattn_core_bwd(qkv.data_ptr(), o.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                               dbias.data_ptr(), bias.data_ptr(), cos.data_ptr(), sin.data_ptr(), None, b, f, hw,
                                               heads, scale, ws.data_ptr(), s))
        nbytes = {"forward": rows * (4 * hid * 2 + heads * 4), "backward": rows * (8 * hid * 2 + heads * 4)}
        for name, fn in (("forward", fwd), ("backward", bwd)):
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
            print(f"{form:6s} {name:8s}: {ms * 1e3:8.1f} us, {nbytes[name] / 1e6:7.1f} MB -> {nbytes[name] / ms / 1e9:.2f} TB/s")
    lib.vdmi_tattn_core_mfma_enable(1)


if __name__ == "__main__":
    main()
