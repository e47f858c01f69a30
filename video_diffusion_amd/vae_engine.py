"""Launch plan of the VAE decoder (reference vdp.py:122-150) on the HIP library.

The decoder runs once after p_sample_loop in GaussianDiffusion.sample (vdp.py:1083-1087).  Layers:
1x1 conv from the latent (vdmi_init_conv, fp32 channels-first in -> bf16 rows), GroupNorm+SiLU
(vdmi_groupnorm_silu), a (1,3,3) conv and three (1,4,4)/stride-2 transposed convs (vdmi_tapgemm; the
transposed convs as four output-parity phases of one launch), and the conversion of the last layer's
rows back to fp32 channels-first (vdmi_final_conv with a channel-selection matrix).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib
from ._lib import EPI_PLAIN, TapGemmArgs, check
from .engine import PackedTaps, TAPS_3x3, _UP_PAIRS, _fill_pass, _ptr, _stream


def _pack_convt(convt: torch.nn.ConvTranspose3d, device, pad_out_to: int = 0):
    """ConvTranspose3d weight [cin][cout][1][4][4] -> four 4-tap weight sets (output parities), optionally with
    the output channels zero-padded to `pad_out_to`."""
    w = convt.weight.detach().float().cpu()[:, :, 0]                      # [cin][cout][4][4]
    bias = convt.bias.detach().float().cpu()
    cin, cout = w.shape[0], w.shape[1]
    n = max(cout, pad_out_to)
    if n != cout:
        w = torch.cat([w, torch.zeros(cin, n - cout, 4, 4)], dim=1)
        bias = torch.cat([bias, torch.zeros(n - cout)])
    packs = []
    for p in (0, 1):
        for q in (0, 1):
            taps = [w[:, :, ky, kx].t() for (ky, _) in _UP_PAIRS[p] for (kx, _) in _UP_PAIRS[q]]   # each [n][cin]
            packs.append(PackedTaps(torch.stack(taps, dim=-1), cin, 0, EPI_PLAIN, device).data)
    return {"w": torch.cat(packs).contiguous(), "phase_stride": packs[0].numel(), "b": bias.to(device).contiguous(),
            "n": n, "cout": cout, "cin": cin}


class VaeDecodePlan:
    """All launches of one decode for a fixed latent shape (B, latent, F, h, w)."""

    def __init__(self, dec, B: int, F: int, h: int, w: int, device):
        lib = _lib.lib()
        self.B, self.F, self.device = B, F, device
        seq = dec.decoder
        latent, c0 = seq[0].in_channels, seq[0].out_channels
        self.z_in = torch.zeros(B, latent, F, h, w, device=device, dtype=torch.float32)
        self.keep, self.calls = [], []
        frames = B * F

        def buf(rows, ch):
            t = torch.empty(rows, ch, device=device, dtype=torch.bfloat16)
            self.keep.append(t)
            return t

        def f32(t):
            t = t.detach().float().to(device).contiguous()
            self.keep.append(t)
            return t

        def groupnorm(x, gn, rows_per_sample, c):
            ws = torch.empty(max(16, lib.vdmi_groupnorm_workspace_bytes(B, rows_per_sample, c)), device=device,
                             dtype=torch.uint8)
            ga, be = f32(gn.weight), f32(gn.bias)
            self.keep.append(ws)
            self.calls.append((lambda s: lib.vdmi_groupnorm_silu(
                x.data_ptr(), x.data_ptr(), B, rows_per_sample, c, gn.num_groups, ga.data_ptr(), be.data_ptr(),
                float(gn.eps), ws.data_ptr(), s), "groupnorm_silu"))

        def base(grid_h, grid_w, out, n, oh=None, ow=None):
            a = TapGemmArgs()
            a.frames, a.frames_per_batch = frames, F
            a.grid_h, a.grid_w = grid_h, grid_w
            a.out_mul, a.out_off_y, a.out_off_x = 1, 0, 0
            a.out_h, a.out_w = oh or grid_h, ow or grid_w
            a.out, a.ldo, a.n, a.epilogue = _ptr(out), out.shape[1], n, EPI_PLAIN
            self.keep.append(a)
            return a

        # 0: 1x1 conv from the latent (fp32 channels-first) -> bf16 rows
        w0 = f32(seq[0].weight.reshape(c0, latent))
        b0 = f32(seq[0].bias)
        x = buf(frames * h * w, c0)
        self.calls.append((lambda s, x=x: lib.vdmi_init_conv(
            self.z_in.data_ptr(), w0.data_ptr(), b0.data_ptr(), x.data_ptr(), B, latent, F, h, w, c0, 1, s), "latent_conv"))
        groupnorm(x, seq[1], F * h * w, c0)
        # 3: (1,3,3) conv
        conv = seq[3]
        pk = PackedTaps(conv.weight.detach().float().cpu().reshape(conv.out_channels, conv.in_channels, -1), c0, 0,
                        EPI_PLAIN, device, tile_bn=64 if conv.out_channels % 64 == 0 else 0)
        self.keep.append(pk)
        y = buf(frames * h * w, conv.out_channels)
        a = base(h, w, y, conv.out_channels)
        _fill_pass(a.main, x, c0, None, 0, h, w, 1, TAPS_3x3, pk.data)
        a.bias = _ptr(f32(conv.bias))
        a.tile_bn = pk.tile_bn
        self.calls.append((lambda s, a=a: lib.vdmi_tapgemm(C.byref(a), s), "conv3x3"))
        x, c = y, conv.out_channels
        groupnorm(x, seq[4], F * h * w, c)
        # 6, 9, 12: transposed convs (x2 each); the last one has no norm and `channels` outputs (padded to 32)
        ch, cw = h, w
        taps = [(dy, dx) for p in (0, 1) for q in (0, 1) for (_, dy) in _UP_PAIRS[p] for (_, dx) in _UP_PAIRS[q]]
        for idx, gn_idx in ((6, 7), (9, 10), (12, None)):
            ct = seq[idx]
            pk = _pack_convt(ct, device, pad_out_to=32 if gn_idx is None else 0)
            self.keep.append(pk)
            oh, ow = 2 * ch, 2 * cw
            y = buf(frames * oh * ow, pk["n"])
            a = base(ch, cw, y, pk["n"], oh, ow)
            a.out_mul = 2
            _fill_pass(a.main, x, c, None, 0, ch, cw, 1, taps, pk["w"])
            a.main.phases, a.main.w_phase_stride = 4, pk["phase_stride"]
            a.bias = _ptr(pk["b"])
            self.calls.append((lambda s, a=a: lib.vdmi_tapgemm(C.byref(a), s), f"convt{idx}"))
            x, c, ch, cw = y, pk["n"], oh, ow
            if gn_idx is not None:
                groupnorm(x, seq[gn_idx], F * ch * cw, c)
        # rows [.., 32] -> fp32 channels-first, first `channels` columns
        cout = pk["cout"]
        self.out = torch.empty(B, cout, F, ch, cw, device=device, dtype=torch.float32)
        sel = torch.zeros(cout, c, device=device, dtype=torch.float32)
        sel[torch.arange(cout), torch.arange(cout)] = 1.0
        self.keep.append(sel)
        self.calls.append((lambda s, x=x, c=c: lib.vdmi_final_conv(
            x.data_ptr(), sel.data_ptr(), None, self.out.data_ptr(), B, c, F, ch, cw, cout, s), "to_channels_first"))

    def run(self):
        s = _stream()
        for fn, name in self.calls:
            check(fn(s), name)


class VaeDecodeEngine:
    """Plans per latent shape; weights are re-packed when the decoder's parameters change."""

    def __init__(self, decoder):
        self.decoder = decoder
        self._plans: Dict[Tuple, VaeDecodePlan] = {}
        self._key = None

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        if z.device.type != "cuda":
            raise _lib.VdmiError("the HIP VAE decoder needs a GPU tensor")
        key = (str(z.device),) + tuple((p.data_ptr(), p._version) for p in self.decoder.parameters())
        if key != self._key:
            self._plans.clear()
            self._key = key
        B, _, F, h, w = z.shape
        pkey = (B, F, h, w)
        plan = self._plans.get(pkey)
        if plan is None:
            plan = self._plans[pkey] = VaeDecodePlan(self.decoder, B, F, h, w, z.device)
        plan.z_in.copy_(z.to(torch.float32))
        plan.run()
        return plan.out.clone()
