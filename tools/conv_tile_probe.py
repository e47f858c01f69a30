#!/usr/bin/env python3
"""A/B: (1,3,3) conv as full-N fused NORM kernel vs PLAIN with 64-channel output tiles (timing only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_diffusion_amd import _lib
from video_diffusion_amd.engine import PackedTaps, TAPS_3x3, _fill_pass
from video_diffusion_amd._lib import TapGemmArgs, EPI_PLAIN, EPI_NORM_SILU

dev = torch.device("cuda:0")
lib = _lib.lib()
B, F = 16, 10
for (cin, cout, hw) in [(64, 64, 64), (128, 128, 32), (256, 256, 16), (512, 512, 8), (1024, 256, 8), (128, 64, 64)]:
    frames = B * F
    rows = frames * hw * hw
    x = torch.randn(rows, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 9) * (1.0 / (cin * 9) ** 0.5)
    bias = torch.zeros(cout, device=dev)
    gain = torch.ones(cout, device=dev)
    out = torch.empty(rows, cout, device=dev, dtype=torch.bfloat16)
    res = {}
    for name, epi, tbn in (("fused-norm", EPI_NORM_SILU, 0), ("plain-auto", EPI_PLAIN, 0), ("plain-bn64", EPI_PLAIN, 64)):
        pk = PackedTaps(w, cin, 0, epi, dev)
        a = TapGemmArgs()
        a.frames, a.frames_per_batch = frames, F
        a.grid_h = a.grid_w = a.out_h = a.out_w = hw
        a.out_mul = 1
        a.out, a.ldo, a.n, a.epilogue = out.data_ptr(), cout, cout, epi
        _fill_pass(a.main, x, cin, None, 0, hw, hw, 1, TAPS_3x3, pk.data)
        a.bias, a.gain, a.norm_scale = bias.data_ptr(), gain.data_ptr(), cout ** 0.5
        a.tile_bn = tbn
        s = torch.cuda.current_stream().cuda_stream
        assert lib.vdmi_tapgemm(C.byref(a), s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.vdmi_tapgemm(C.byref(a), s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res[name] = us
    fl = 2.0 * rows * cout * cin * 9
    print(f"{cin:5d}->{cout:4d} @{hw:2d}^2: " + "  ".join(f"{k} {v:7.1f} us ({fl / v / 1e6:6.0f} TF)" for k, v in res.items()))
