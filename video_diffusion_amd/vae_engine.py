"""Launch plans of the VAE decoder / encoder (reference vdp.py:83-150) and of the FrameConditioningEncoder
(vdp.py:239-269) on the HIP library.

The encoders take fp32 channels-first input: a 1x1 "selection" vdmi_init_conv lays it out as bf16 rows with
the channels zero-padded to 16, then every conv is a vdmi_tapgemm ((1,4,4)/stride-2 convs through the
space-to-depth view when the input width allows, SiLU fused for the frame encoder), GroupNorm+SiLU is
vdmi_groupnorm_silu, and the VAE's two 1x1 heads are one vdmi_final_conv back to fp32 channels-first.

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
from .engine import PackedTaps, TAPS_3x3, TAPS_4x4_S2, _UP_PAIRS, _fill_pass, _ptr, _stream


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


class _ConvPlan:
    """Helpers shared by the plans below: buffers, parameter copies, and the recurring launches."""

    def __init__(self, B: int, F: int, device):
        self.B, self.F, self.device = B, F, device
        self.frames = B * F
        self.keep, self.calls = [], []

    def buf(self, rows, ch):
        t = torch.empty(rows, ch, device=self.device, dtype=_lib.ACT_DTYPE)
        self.keep.append(t)
        return t

    def f32(self, t):
        t = t.detach().float().to(self.device).contiguous()
        self.keep.append(t)
        return t

    def base(self, grid_h, grid_w, out, n, oh=None, ow=None):
        a = TapGemmArgs()
        a.frames, a.frames_per_batch = self.frames, self.F
        a.grid_h, a.grid_w = grid_h, grid_w
        a.out_mul, a.out_off_y, a.out_off_x = 1, 0, 0
        a.out_h, a.out_w = oh or grid_h, ow or grid_w
        a.out, a.ldo, a.n, a.epilogue = _ptr(out), out.shape[1], n, EPI_PLAIN
        self.keep.append(a)
        return a

    def groupnorm(self, x, gn, rows_per_sample, c):
        lib = _lib.lib()
        ws = torch.empty(max(16, lib.vdmi_groupnorm_workspace_bytes(self.B, rows_per_sample, c)), device=self.device,
                         dtype=torch.uint8)
        ga, be = self.f32(gn.weight), self.f32(gn.bias)
        self.keep.append(ws)
        self.calls.append((lambda s: lib.vdmi_groupnorm_silu(
            x.data_ptr(), x.data_ptr(), self.B, rows_per_sample, c, gn.num_groups, ga.data_ptr(), be.data_ptr(),
            float(gn.eps), ws.data_ptr(), s), "groupnorm_silu"))

    def to_rows(self, x_in, cin, h, w):
        """fp32 channels-first (B, cin, F, h, w) -> bf16 rows [B*F*h*w][16], channels zero-padded."""
        lib = _lib.lib()
        if cin > 16:
            raise _lib.VdmiError("the encoders take at most 16 input channels")
        sel = torch.zeros(16, cin, device=self.device, dtype=torch.float32)
        sel[torch.arange(cin), torch.arange(cin)] = 1.0
        self.keep.append(sel)
        rows = self.buf(self.frames * h * w, 16)
        self.calls.append((lambda s: lib.vdmi_init_conv(x_in.data_ptr(), sel.data_ptr(), None, rows.data_ptr(), self.B, cin,
                                                        self.F, h, w, 16, 1, s), "to_rows"))
        return rows

    def conv(self, name, conv, x, c, h, w, act=0):
        """Conv3d with a (1,k,k) kernel, stride (1,s,s), padding (0,p,p) on bf16 rows x [.][c] (c >= conv.in_channels:
        extra input channels get zero weights).  Returns (rows, out_h, out_w)."""
        lib = _lib.lib()
        k, st, pad = conv.kernel_size[1], conv.stride[1], conv.padding[1]
        cin, n = conv.in_channels, conv.out_channels
        wt = conv.weight.detach().float().cpu()[:, :, 0]                      # [n][cin][k][k]
        if c > cin:
            wt = torch.cat([wt, torch.zeros(n, c - cin, k, k)], dim=1)
        oh, ow = (h + 2 * pad - k) // st + 1, (w + 2 * pad - k) // st + 1
        y = self.buf(self.frames * oh * ow, n)
        a = self.base(oh, ow, y, n)
        if (k, st, pad) == (4, 2, 1) and c % 64 == 0 and h % 2 == 0 and w % 2 == 0:
            # space-to-depth view: channel (2a+b)*c + ci, tap (dY, dX)  <-  kernel element (2dY+a, 2dX+b)
            subs = [torch.stack([wt[:, :, 2 * dy + a_, 2 * dx + b_] for dy in (0, 1) for dx in (0, 1)], dim=-1)
                    for a_ in (0, 1) for b_ in (0, 1)]
            pk = PackedTaps(torch.cat(subs, dim=1), 4 * c, 0, EPI_PLAIN, self.device, tile_bn=64)
            _fill_pass(a.main, x, 4 * c, None, 0, h, w, 1, [(0, 0), (0, 1), (1, 0), (1, 1)], pk.data)
            a.main.space_to_depth = 1
        else:
            taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]
            pk = PackedTaps(wt.reshape(n, c, k * k), c, 0, EPI_PLAIN, self.device,
                            tile_bn=64 if (n % 64 == 0 and st == 1 and c >= 64) else 0)
            _fill_pass(a.main, x, c, None, 0, h, w, st, taps, pk.data)
        self.keep.append(pk)
        a.tile_bn = pk.tile_bn
        a.bias = _ptr(self.f32(conv.bias))
        a.act = act
        self.calls.append((lambda s, a=a: lib.vdmi_tapgemm(C.byref(a), s), name))
        return y, oh, ow

    def run(self):
        s = _stream()
        for fn, name in self.calls:
            check(fn(s), name)


class VaeEncodePlan(_ConvPlan):
    """VAEEncoder.forward (reference vdp.py:83-119) for a fixed input shape (B, channels, F, H, W):
    self.mean / self.logvar are fp32 (B, latent, F, H/8, W/8); logvar is clamped to [-30, 20]."""

    def __init__(self, enc, B: int, F: int, H: int, W: int, device):
        super().__init__(B, F, device)
        lib = _lib.lib()
        seq = enc.encoder
        cin = seq[0].in_channels
        self.x_in = torch.zeros(B, cin, F, H, W, device=device, dtype=torch.float32)
        x, c, h, w = self.to_rows(self.x_in, cin, H, W), 16, H, W
        for idx in (0, 3, 6, 9):
            x, h, w = self.conv(f"conv{idx}", seq[idx], x, c, h, w)
            c = seq[idx].out_channels
            self.groupnorm(x, seq[idx + 1], F * h * w, c)
        latent = enc.to_mean.out_channels
        if 2 * latent > 8:
            raise _lib.VdmiError("the fused VAE heads support latent_dim <= 4")
        wh = self.f32(torch.cat([enc.to_mean.weight.reshape(latent, c), enc.to_logvar.weight.reshape(latent, c)], 0))
        bh = self.f32(torch.cat([enc.to_mean.bias, enc.to_logvar.bias], 0))
        self.heads = torch.empty(B, 2 * latent, F, h, w, device=device, dtype=torch.float32)
        self.calls.append((lambda s, x=x: lib.vdmi_final_conv(x.data_ptr(), wh.data_ptr(), bh.data_ptr(), self.heads.data_ptr(),
                                                              B, c, F, h, w, 2 * latent, s), "heads"))
        self.latent = latent

    def outputs(self):
        return self.heads[:, :self.latent].clone(), self.heads[:, self.latent:].clamp(-30.0, 20.0)


class FrameEncodePlan(_ConvPlan):
    """FrameConditioningEncoder.forward (reference vdp.py:239-269): three (1,3,3) convs (strides 1, 2, 2) with
    SiLU; self.out is bf16 rows [B*F*h*w][feature_dim], i.e. 'b (f h w) c' already."""

    def __init__(self, fe, B: int, F: int, H: int, W: int, device):
        super().__init__(B, F, device)
        seq = fe.feature_extractor
        cin = seq[0].in_channels
        self.x_in = torch.zeros(B, cin, F, H, W, device=device, dtype=torch.float32)
        x, c, h, w = self.to_rows(self.x_in, cin, H, W), 16, H, W
        for idx in (0, 2, 4):
            x, h, w = self.conv(f"conv{idx}", seq[idx], x, c, h, w, act=1)
            c = seq[idx].out_channels
        self.out, self.tokens, self.dim = x, F * h * w, c


class VaeDecodePlan(_ConvPlan):
    """All launches of one decode for a fixed latent shape (B, latent, F, h, w)."""

    def __init__(self, dec, B: int, F: int, h: int, w: int, device):
        super().__init__(B, F, device)
        lib = _lib.lib()
        seq = dec.decoder
        latent, c0 = seq[0].in_channels, seq[0].out_channels
        self.z_in = self.x_in = torch.zeros(B, latent, F, h, w, device=device, dtype=torch.float32)
        frames = self.frames
        # 0: 1x1 conv from the latent (fp32 channels-first) -> bf16 rows
        w0 = self.f32(seq[0].weight.reshape(c0, latent))
        b0 = self.f32(seq[0].bias)
        x = self.buf(frames * h * w, c0)
        self.calls.append((lambda s, x=x: lib.vdmi_init_conv(
            self.z_in.data_ptr(), w0.data_ptr(), b0.data_ptr(), x.data_ptr(), B, latent, F, h, w, c0, 1, s), "latent_conv"))
        self.groupnorm(x, seq[1], F * h * w, c0)
        # 3: (1,3,3) conv
        x, _, _ = self.conv("conv3x3", seq[3], x, c0, h, w)
        c = seq[3].out_channels
        self.groupnorm(x, seq[4], F * h * w, c)
        # 6, 9, 12: transposed convs (x2 each); the last one has no norm and `channels` outputs (padded to 32)
        ch, cw = h, w
        taps = [(dy, dx) for p in (0, 1) for q in (0, 1) for (_, dy) in _UP_PAIRS[p] for (_, dx) in _UP_PAIRS[q]]
        for idx, gn_idx in ((6, 7), (9, 10), (12, None)):
            pk = _pack_convt(seq[idx], device, pad_out_to=32 if gn_idx is None else 0)
            self.keep.append(pk)
            oh, ow = 2 * ch, 2 * cw
            y = self.buf(frames * oh * ow, pk["n"])
            a = self.base(ch, cw, y, pk["n"], oh, ow)
            a.out_mul = 2
            _fill_pass(a.main, x, c, None, 0, ch, cw, 1, taps, pk["w"])
            a.main.phases, a.main.w_phase_stride = 4, pk["phase_stride"]
            a.bias = _ptr(pk["b"])
            self.calls.append((lambda s, a=a: lib.vdmi_tapgemm(C.byref(a), s), f"convt{idx}"))
            x, c, ch, cw = y, pk["n"], oh, ow
            if gn_idx is not None:
                self.groupnorm(x, seq[gn_idx], F * ch * cw, c)
        # rows [.., 32] -> fp32 channels-first, first `channels` columns
        cout = pk["cout"]
        self.out = torch.empty(B, cout, F, ch, cw, device=device, dtype=torch.float32)
        sel = torch.zeros(cout, c, device=device, dtype=torch.float32)
        sel[torch.arange(cout), torch.arange(cout)] = 1.0
        self.keep.append(sel)
        self.calls.append((lambda s, x=x, c=c: lib.vdmi_final_conv(
            x.data_ptr(), sel.data_ptr(), None, self.out.data_ptr(), B, c, F, ch, cw, cout, s), "to_channels_first"))


class _ShapeCachedEngine:
    """Plans per input shape; weights are re-packed when the module's parameters change."""
    plan_cls = None

    def __init__(self, module):
        self.module = module
        self._plans: Dict[Tuple, object] = {}
        self._key = None
        self._values = None
        self._generation = 0

    def invalidate(self):
        """The module's weights were rewritten in a way the version counters do not see (through `.data`): plans and
        packed weights are rebuilt at the next call."""
        self._generation += 1

    def check_values(self):
        """Compare a fingerprint of the parameter values with the one taken when the plans were built and rebuild on a
        difference.  One host synchronisation: called once per sampling call (GaussianDiffusion.sample), never per
        encode / decode and never under graph capture."""
        from .engine import values_fingerprint
        vals = values_fingerprint(self.module.parameters())
        if self._values is not None and vals != self._values:
            self.invalidate()
        self._values = vals

    def _plan(self, x: torch.Tensor):
        if x.device.type != "cuda":
            raise _lib.VdmiError("the HIP encoders need a GPU tensor")
        # per call: pointers and autograd version counters only (no device synchronisation, usable under capture once
        # the plan exists); writes through `.data` are caught by check_values() / invalidate()
        key = (str(x.device), self._generation) + tuple((p.data_ptr(), p._version) for p in self.module.parameters())
        if key != self._key:
            # (re)build: the weights are packed from the live values now, so this is also when their fingerprint is taken
            # (a host copy, once per rebuild - never on the calls that find their plan)
            from .engine import values_fingerprint
            self._plans.clear()
            self._key = key
            self._values = values_fingerprint(self.module.parameters())
        B, _, F, H, W = x.shape
        plan = self._plans.get((B, F, H, W))
        if plan is None:
            plan = self._plans[(B, F, H, W)] = self.plan_cls(self.module, B, F, H, W, x.device)
        plan.x_in.copy_(x.to(torch.float32))
        plan.run()
        return plan


class VaeEncodeEngine(_ShapeCachedEngine):
    plan_cls = VaeEncodePlan

    def encode(self, x):
        return self._plan(x).outputs()


class FrameEncodeEngine(_ShapeCachedEngine):
    plan_cls = FrameEncodePlan

    def encode(self, prev_latents):
        plan = self._plan(prev_latents)
        return plan.out.float().reshape(plan.B, plan.tokens, plan.dim)


class VaeDecodeEngine(_ShapeCachedEngine):
    plan_cls = VaeDecodePlan

    def decode(self, z):
        return self._plan(z).out.clone()
