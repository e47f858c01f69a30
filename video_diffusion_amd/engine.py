"""Execution plan of the Unet3D forward on MI355X.

`UnetEngine` packs the module's parameters once into the layouts the HIP kernels stream
(bf16 MFMA-fragment order, LayerNorm gammas folded into the following projection) and, per
input shape, builds a `Plan`: a fixed list of C-ABI calls over preallocated channels-last
bf16 activation buffers.  Running the plan is ~100 asynchronous launches on the current HIP
stream with no host synchronisation, so a whole denoise step can be captured in a HIP graph.

The order of operations follows the reference's Unet3D.forward
(video_diffusion_pytorch/video_diffusion_pytorch.py:756-831 of the reference).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import (EPI_NORM_SILU, EPI_PLAIN, EPI_QKV_ATTN, EPI_QKV_SLA, TapGemmArgs, TapPass, check)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def tile_for(epilogue: int, n: int) -> Tuple[int, int]:
    bn, bkc = C.c_int(), C.c_int()
    check(_lib.lib().vdmi_tapgemm_tile(epilogue, n, C.byref(bn), C.byref(bkc)), "vdmi_tapgemm_tile")
    return bn.value, bkc.value


class PackedTaps:
    """Weights of one implicit-GEMM pass, packed on the host by vdmi_pack_taps."""

    def __init__(self, w: torch.Tensor, cin0: int, cin1: int, epilogue: int, device, want_colsum=False,
                 tile_bn: int = 0):
        # w: float32 [n][cin0+cin1][ntaps] on the CPU
        w = w.detach().to(torch.float32).contiguous().cpu()
        n, cin, ntaps = w.shape
        assert cin == cin0 + cin1
        self.n, self.ntaps, self.cin0, self.cin1 = n, ntaps, cin0, cin1
        self.tile_bn = tile_bn
        # tile_bn = 64 (PLAIN only): 64-channel output tiles, 64-channel input chunks
        self.bn, self.bkc = (64, 64) if tile_bn == 64 else tile_for(epilogue, n)
        self.npad = (n + self.bn - 1) // self.bn * self.bn
        lib = _lib.lib()
        elems = lib.vdmi_pack_taps_elems(n, self.npad, cin0, cin1, ntaps, self.bkc)
        out = torch.empty(elems, dtype=torch.int16)
        colsum = torch.empty(n, dtype=torch.float32) if want_colsum else None
        check(lib.vdmi_pack_taps(w.data_ptr(), n, self.npad, cin0, cin1, ntaps, self.bkc, out.data_ptr(),
                                 _ptr(colsum)), "vdmi_pack_taps")
        self.data = out.view(_lib.ACT_DTYPE).to(device)
        self.colsum = colsum.to(device) if want_colsum else None


SPLIT_MIN_CHANNELS = int(os.environ.get("VDMI_SPLIT_MIN_CHANNELS", "256"))
TIME_TABLES = os.environ.get("VDMI_TIME_TABLES", "1") != "0"          # sampling loops copy scale/shift rows from per-timestep tables
SIDE_RES_CONV = os.environ.get("VDMI_SIDE_RES_CONV", "1") != "0"      # res_conv GEMMs of the wide layers on the side stream
# ... and of the 64-channel blocks the weight-stationary conv covers (their block2 then has an identity residual).  Off:
# measured 4.06 -> 4.15 ms per step; the extra 1x1 GEMM moves 126 MB more per layer than the fused second pass it replaces
SIDE_RES_WS = os.environ.get("VDMI_SIDE_RES_WS", "0") != "0"

TAPS_3x3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
TAPS_4x4_S2 = [(ky - 1, kx - 1) for ky in range(4) for kx in range(4)]
# transposed 4x4 / stride 2 / pad 1 conv, output parity p: (kernel index, input offset) pairs
_UP_PAIRS = {0: [(1, 0), (3, -1)], 1: [(0, 1), (2, 0)]}


def _fill_pass(p: TapPass, src0, c0, src1, c1, in_h, in_w, stride, taps, wdata, group_stride=0, group_frames=0):
    p.src0, p.src1 = _ptr(src0), _ptr(src1)
    p.c0, p.c1 = c0, c1
    p.in_h, p.in_w, p.stride, p.ntaps = in_h, in_w, stride, len(taps)
    for i, (dy, dx) in enumerate(taps):
        p.dy[i], p.dx[i] = dy, dx
    p.w = _ptr(wdata)
    p.w_group_stride, p.w_group_frames = group_stride, group_frames


class _Call:
    """One launch.  side: it may run on the plan's side stream (nothing before the next `join` call reads its
    output); join: the side stream's work must have finished before this launch."""
    __slots__ = ("fn", "args", "name", "side", "join")

    def __init__(self, fn, args, name, side=False, join=False):
        self.fn, self.args, self.name, self.side, self.join = fn, args, name, side, join


class Plan:
    """All launches of one Unet3D forward for a fixed (B, F, H, W)."""

    def __init__(self, eng: "UnetEngine", B: int, F: int, H: int, W: int, device, build: bool = True, L: int = 0,
                 focus: bool = False, io: Optional[Dict[str, torch.Tensor]] = None, time_table: int = 0):
        self.eng, self.B, self.F, self.H, self.W, self.device = eng, B, F, H, W, device
        self.time_table = time_table          # > 0: scale/shift rows come from the engine's table of that many timesteps
        self._io = io or {}                   # input / output tensors handed in by an owner (views of its batch)
        self.focus_enabled = focus            # temporal attentions read the focus-present flags (focus_in)
        self.L = L                            # context tokens per sample of the frame cross-attention (0: layers skipped)
        self.calls: List[_Call] = []
        self.ctx_calls: List[_Call] = []      # key/value projection of the context: once per context, not per step
        self.keep: List[object] = []          # tensors / ctypes objects the calls point into
        u = eng.unet
        self.x_in = self._io_tensor("x_in", (B, u.channels, F, H, W), torch.float32)
        self.t_in = self._io_tensor("t_in", (B,), torch.int64)
        self.cond_in = self._io_tensor("cond_in", (B, u.cond_dim), torch.float32) if u.has_cond else None
        self.null_mask = self._io_tensor("null_mask", (B,), torch.uint8) if u.has_cond else None
        self.focus_in = self._io_tensor("focus_in", (B,), torch.uint8)       # focus-present mask, all False by default
        self.eps_out = self._io_tensor("eps_out", (B, u.out_dim, F, H, W), torch.float32)
        self.taps: Dict[str, torch.Tensor] = {}   # named intermediate activations (for tests)
        self._side = None
        self._join_next = False     # the next launch added must wait for the side stream
        self._setup()
        for c in self.calls:        # the time-embedding launches do not depend on the image: side stream
            c.side = True
        self._tables_pending = True
        if build:
            self._build()

    # -- buffer / call helpers -------------------------------------------------------------
    def _io_tensor(self, name, shape, dtype, fill=0) -> torch.Tensor:
        t = self._io.get(name)
        if t is None:
            return torch.full(shape, fill, device=self.device, dtype=dtype)
        if tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous():
            raise _lib.VdmiError(f"plan tensor {name}: expected contiguous {tuple(shape)} {dtype}, got {tuple(t.shape)} {t.dtype}")
        return t

    def _buf(self, rows: int, ch: int) -> torch.Tensor:
        t = torch.empty(rows, ch, device=self.device, dtype=_lib.ACT_DTYPE)
        self.keep.append(t)
        return t

    def _add(self, fn, args, name, side=False):
        self.calls.append(_Call(fn, args, name, side, self._join_next))
        self._join_next = False

    def _tapgemm(self, a: TapGemmArgs, name: str, side: bool = False):
        self.keep.append(a)
        lib = _lib.lib()
        # algorithmic work of the launch: 2*M*N*K over the positions computed (main pass + fused 1x1 pass)
        m = a.frames * a.grid_h * a.grid_w
        k = (a.main.c0 + a.main.c1) * a.main.ntaps + ((a.res.c0 + a.res.c1) if a.has_res == 1 else 0)
        bn = 64 if a.tile_bn == 64 else tile_for(a.epilogue, a.n)[0]
        in_rows = a.frames * a.main.in_h * a.main.in_w
        kernel = ("wsconv64" if lib.vdmi_wsconv_covers(C.byref(a)) else
                  ("dconv" if a.epilogue == EPI_PLAIN else f"dconv<epi={a.epilogue}>") if lib.vdmi_dconv_covers(C.byref(a))
                  else f"tapgemm<epi={a.epilogue},bn={bn}>")      # one class per kernel instantiation family
        meta = {"kernel": kernel, "flops": 2.0 * m * a.n * k,
                "bytes": 2.0 * (in_rows * (a.main.c0 + a.main.c1) / (4 if a.main.space_to_depth else 1)
                                + m * a.n + a.n * k)}
        self._add(lambda s, a=a: lib.vdmi_tapgemm(C.byref(a), s), meta, name, side)

    def _base_args(self, frames, grid_h, grid_w, out, n, epilogue, out_h=None, out_w=None) -> TapGemmArgs:
        a = TapGemmArgs()
        a.frames, a.frames_per_batch = frames, self.F
        a.grid_h, a.grid_w = grid_h, grid_w
        a.out_mul, a.out_off_y, a.out_off_x = 1, 0, 0
        a.out_h, a.out_w = out_h or grid_h, out_w or grid_w
        a.out, a.ldo, a.n, a.epilogue = _ptr(out), out.shape[1], n, epilogue
        return a

    # -- layer builders ---------------------------------------------------------------------
    def _block(self, name, pk, srcs, h, w, gain, shift, bstride, res=None) -> torch.Tensor:
        """Block: conv3x3 + RMSNorm + scale/shift + SiLU, optionally + residual (res_conv or identity)."""
        if self._tables_pending:    # first reader of the scale/shift tables
            self._tables_pending = False
            self._join_next = True
        frames = self.B * self.F
        (s0, c0), (s1, c1) = srcs[0], (srcs[1] if len(srcs) > 1 else (None, 0))
        w_pack, bias = pk
        if w_pack.tile_bn == 64:
            return self._block_split(name, pk, srcs, h, w, gain, shift, bstride, res)
        # a 128-channel block with a long contraction on a small image is one workgroup's latency long with the fused
        # epilogue (a nominal 8 samples give 160 workgroups of 72 tap steps): the split form deals its contraction to twice
        # the workgroups.  The packed weights serve both forms (same fragment order).
        if (SPLIT_LONG_K and w_pack.n == 128 and (c0 + c1) >= SPLIT_LONG_K and 8 * self.F * h * w <= 128 * 200
                and (res is None or res[0] != "conv")):
            return self._block_split(name, pk, srcs, h, w, gain, shift, bstride, res)
        out = self._buf(frames * h * w, w_pack.n)
        a = self._base_args(frames, h, w, out, w_pack.n, EPI_NORM_SILU)
        _fill_pass(a.main, s0, c0, s1, c1, h, w, 1, TAPS_3x3, w_pack.data)
        a.bias = _ptr(bias)
        a.gain, a.shift = _ptr(gain), _ptr(shift)
        a.gain_bstride = a.shift_bstride = bstride
        a.norm_scale = float(w_pack.n) ** 0.5
        if res is not None:
            kind = res[0]
            if kind == "conv":
                _, rpack, rbias, rsrcs = res
                (r0, rc0), (r1, rc1) = rsrcs[0], (rsrcs[1] if len(rsrcs) > 1 else (None, 0))
                a.has_res = 1
                _fill_pass(a.res, r0, rc0, r1, rc1, h, w, 1, [(0, 0)], rpack.data)
                a.res_bias = _ptr(rbias)
            else:
                rt = res[1]
                if len(res) > 2 and res[2]:
                    self._join_next = True      # the residual was produced on the side stream
                a.has_res = 2
                a.res_identity, a.ld_res = _ptr(rt), rt.shape[1]
        self._tapgemm(a, name)
        return out

    def _block_split(self, name, pk, srcs, h, w, gain, shift, bstride, res=None) -> torch.Tensor:
        """Block for wide, small-image layers: the conv runs over split output-channel tiles (more pixels
        per workgroup, so each weight byte is fetched by far fewer workgroups), RMSNorm + scale/shift +
        SiLU (+ identity residual) is a separate row kernel, res_conv a plain 1x1 GEMM on top."""
        lib = _lib.lib()
        frames = self.B * self.F
        rows = frames * h * w
        (s0, c0), (s1, c1) = srcs[0], (srcs[1] if len(srcs) > 1 else (None, 0))
        w_pack, bias = pk
        n = w_pack.n
        # Small images launch few workgroups (rows / 256 pixel tiles x n / 64 channel tiles) and each walks the whole
        # contraction on its own: below ~one workgroup per CU the channel chunks are dealt to 2 or 4 workgroups per
        # tile, which write partial sums that the normalisation kernel adds on its way in.  The choice is made for a
        # nominal 8 samples per launch, not the actual batch: the order of a sample's additions must not depend on
        # what else is in the batch (shards and sub-batches give the same bits).
        wgs = max(1, (8 * self.F * h * w + 255) // 256) * (n // 64)
        ksplit = 1
        if K_SPLIT_BELOW and wgs < K_SPLIT_BELOW:
            ksplit = min(4, max(1, K_SPLIT_TARGET // wgs), (c0 + c1) // 64)
            if not (K_SPLIT_ODD and ksplit == 3):
                ksplit = 1 if ksplit < 2 else (2 if ksplit < 4 else 4)
        raw = self._buf(ksplit * rows, n)
        a = self._base_args(frames, h, w, raw, n, EPI_PLAIN)
        _fill_pass(a.main, s0, c0, s1, c1, h, w, 1, TAPS_3x3, w_pack.data)
        a.bias = _ptr(bias)
        a.tile_bn = 64
        if ksplit > 1:
            a.k_split, a.out_split_stride = ksplit, rows * n
        self._tapgemm(a, name + ".conv")
        identity = res[1] if (res is not None and res[0] == "identity") else None
        if res is not None and res[0] == "identity" and len(res) > 2 and res[2]:
            self._join_next = True      # the residual was produced on the side stream
        out = self._buf(rows, n)
        meta = {"kernel": "rmsnorm_silu", "flops": 0.0, "bytes": 2.0 * rows * n * (3 if identity is not None else 2)}
        self._add(lambda s, raw=raw, out=out, identity=identity: lib.vdmi_rmsnorm_silu_sum(
            raw.data_ptr(), ksplit, rows * n, n, out.data_ptr(), n, rows, n, _ptr(gain), bstride, _ptr(shift), bstride,
            self.F * h * w, float(n) ** 0.5, _ptr(identity), identity.shape[1] if identity is not None else 0, s),
            meta, name + ".norm")
        if res is not None and res[0] == "conv":
            _, rpack, rbias, rsrcs = res
            (r0, rc0), (r1, rc1) = rsrcs[0], (rsrcs[1] if len(rsrcs) > 1 else (None, 0))
            y = self._buf(rows, n)
            b = self._base_args(frames, h, w, y, n, EPI_PLAIN)
            _fill_pass(b.main, r0, rc0, r1, rc1, h, w, 1, [(0, 0)], rpack.data)
            b.bias = _ptr(rbias)
            b.has_res, b.res_identity, b.ld_res = 2, _ptr(out), n
            b.tile_bn = rpack.tile_bn
            self._tapgemm(b, name + ".res")
            return y
        return out

    def _resnet(self, name, rb, srcs, h, w) -> torch.Tensor:
        """ResnetBlock (reference vdp.py:429-452)."""
        if rb["ss_off"] is not None:
            off = rb["ss_off"]
            gain1 = self.gain[:, off:]
            shift1 = self.shift[:, off:]
            bstride = self.gain.shape[1]
            g1, s1 = gain1, shift1
        else:
            g1, s1, bstride = rb["gamma1"], None, 0
        early = None
        # 64 -> 64 channel block2 of a 32- or 64-pixel-wide image: with an identity residual it runs on the
        # weight-stationary kernel, which has no fused 1x1 pass
        ws_block2 = (SIDE_RES_WS and rb["w2"].n == 64 and rb["w1"].n == 64 and w in (32, 64) and h % (256 // w) == 0)
        if SIDE_RES_CONV and rb["wres"] is not None and "wt" not in rb and (rb["w2"].tile_bn == 64 or ws_block2):
            # res_conv(x) is its own 1x1 GEMM.  It depends on the block's input only, so it runs on the side stream
            # beside block1 / block2 and comes back as the residual of block2 (wide layers: of its normalisation kernel).
            if self._tables_pending:
                self._tables_pending = False
                self._join_next = True
            frames, n = self.B * self.F, rb["w2"].n
            (r0, rc0), (r1, rc1) = srcs[0], (srcs[1] if len(srcs) > 1 else (None, 0))
            early = self._buf(frames * h * w, n)
            b = self._base_args(frames, h, w, early, n, EPI_PLAIN)
            _fill_pass(b.main, r0, rc0, r1, rc1, h, w, 1, [(0, 0)], rb["wres"].data)
            b.bias = _ptr(rb["bres"])
            b.tile_bn = rb["wres"].tile_bn
            self._tapgemm(b, name + ".block2.res", side=True)
        hbuf = self._block(name + ".block1", (rb["w1"], rb["b1"]), srcs, h, w, g1, s1, bstride)
        cmid = rb["w1"].n
        if "wt" in rb:
            return self._resnet_temporal(name, rb, srcs, hbuf, cmid, h, w)
        if early is not None:
            res = ("identity", early, True)
        elif rb["wres"] is not None:
            res = ("conv", rb["wres"], rb["bres"], srcs)
        else:
            res = ("identity", srcs[0][0])
        return self._block(name + ".block2", (rb["w2"], rb["b2"]), [(hbuf, cmid)], h, w, rb["gamma2"], None, 0, res)

    def _resnet_temporal(self, name, rb, srcs, hbuf, cmid, h, w) -> torch.Tensor:
        """Tail of the variant ResnetBlock (reference version1diffusion.py:320-323): block2, then a (3,1,1)
        convolution over frames, then + res_conv(x).  The temporal conv is a 3-tap pass over the rows seen as
        B images of F x (h*w) pixels."""
        frames, hw = self.B * self.F, h * w
        h2 = self._block(name + ".block2", (rb["w2"], rb["b2"]), [(hbuf, cmid)], h, w, rb["gamma2"], None, 0, None)
        if rb["wres"] is not None:
            (r0, rc0), (r1, rc1) = srcs[0], (srcs[1] if len(srcs) > 1 else (None, 0))
            res = self._buf(frames * hw, cmid)
            b = self._base_args(frames, h, w, res, cmid, EPI_PLAIN)
            _fill_pass(b.main, r0, rc0, r1, rc1, h, w, 1, [(0, 0)], rb["wres"].data)
            b.bias = _ptr(rb["bres"])
            b.tile_bn = rb["wres"].tile_bn
            self._tapgemm(b, name + ".res")
        else:
            res = srcs[0][0]
        y = self._buf(frames * hw, cmid)
        a = self._base_args(self.B, self.F, hw, y, cmid, EPI_PLAIN)
        _fill_pass(a.main, h2, cmid, None, 0, self.F, hw, 1, [(-1, 0), (0, 0), (1, 0)], rb["wt"].data)
        a.bias = _ptr(rb["bt"])
        a.has_res, a.res_identity, a.ld_res = 2, _ptr(res), cmid
        self._tapgemm(a, name + ".temporal_conv")
        return y

    def _sta(self, name, st, x, c, h, w, per_frame: bool = False) -> torch.Tensor:
        """Residual(PreNorm(SpatioTemporalAttention)) (reference version1diffusion.py:325-345): full softmax
        attention over the F*h*w tokens of a sample; the keys/values are the layer's own projection, repacked
        into the operand order of the cross-attention core.  per_frame: the h*w tokens of each frame instead
        (SpatialAttention of video_diffusion_guidance.py:431-459)."""
        lib = _lib.lib()
        frames, hw = self.B * self.F, h * w
        rows, hid = frames * hw, st["hid"]
        n_tok, ngrp = (hw, frames) if per_frame else (self.F * hw, self.B)
        heads = hid // 32
        if per_frame and st.get("fused") is not None and hw <= 64:
            y = self._buf(rows, c)
            wp = st["fused"]
            meta = {"kernel": "sattn_fused", "flops": 2.0 * rows * c * 4 * hid + 4.0 * rows * hw * hid, "bytes": 4.0 * rows * c}
            self._add(lambda s, x=x, y=y: lib.vdmi_sattn_fused_wide(
                x.data_ptr(), y.data_ptr(), wp.data_ptr(), frames, hw, c, heads, st["scale"], 1e-5, s), meta, name + ".fused")
            return y
        qkv = self._buf(rows, 3 * hid)
        a = self._base_args(frames, h, w, qkv, 3 * hid, EPI_QKV_ATTN)
        _fill_pass(a.main, x, c, None, 0, h, w, 1, [(0, 0)], st["wqkv"].data)
        a.colsum = _ptr(st["wqkv"].colsum)
        a.ln_eps, a.q_scale = 1e-5, st["scale"] * math.log2(math.e)
        self._tapgemm(a, name + ".qkv")
        kvp = torch.empty(lib.vdmi_xattn_kv_elems(1, ngrp, n_tok, heads), device=self.device, dtype=_lib.ACT_DTYPE)
        self.keep.append(kvp)
        self._add(lambda s, qkv=qkv: lib.vdmi_xattn_pack_kv(qkv.data_ptr() + 2 * hid, 3 * hid, kvp.data_ptr(), 1, ngrp,
                                                            n_tok, heads, s), None, name + ".pack_kv")
        o = self._buf(rows, hid)
        meta = {"kernel": "xattn_core", "flops": 4.0 * rows * hid * ((n_tok + 31) // 32) * 32, "bytes": 4.0 * rows * hid}
        self._add(lambda s, qkv=qkv, o=o: lib.vdmi_xattn_core(qkv.data_ptr(), 3 * hid, o.data_ptr(), kvp.data_ptr(), ngrp,
                                                              n_tok, n_tok, heads, s), meta, name + ".core")
        y = self._buf(rows, c)
        b = self._base_args(frames, h, w, y, c, EPI_PLAIN)
        _fill_pass(b.main, o, hid, None, 0, h, w, 1, [(0, 0)], st["wout"].data)
        b.has_res, b.res_identity, b.ld_res = 2, _ptr(x), c
        self._tapgemm(b, name + ".out")
        return y

    def _attention(self, name, at, x, c, h, w, temporal: bool, focus: bool = False) -> torch.Tensor:
        """Residual(PreNorm(Attention)) over frames (temporal) or over the h*w positions (spatial).
        focus: the layer receives the focus-present mask (self.focus_in, reference vdp.py:800, 812, 821)."""
        fptr = self.focus_in.data_ptr() if (focus and self.focus_enabled) else None
        lib = _lib.lib()
        frames, hw = self.B * self.F, h * w
        rows = frames * hw
        hid = at["hid"]
        # 256 and 512 channels: the wide form of the fused kernel (one wave per head, weights streamed into registers).
        # The choice depends on the layer's shape only, never on the batch, so a sample's bits do not depend on the batch.
        wide = hid == 256 and (c == 512 or (c == 256 and hw < TATTN_WIDE_BELOW_HW))
        if temporal and at.get("fused") is not None and self.F <= 32 and (c < 256 or hw >= 256 or wide):
            y = self._buf(rows, c)
            wp = at["fused"]
            meta = {"kernel": "tattn_fused", "flops": 2.0 * rows * c * 4 * hid + 4.0 * rows * self.F * hid,
                    "bytes": 4.0 * rows * c}
            fn = lib.vdmi_tattn_fused_wide if wide else lib.vdmi_tattn_fused
            self._add(lambda s, x=x, y=y: fn(
                x.data_ptr(), y.data_ptr(), wp.data_ptr(), _ptr(self.pos_bias), _ptr(self.rot_cos),
                _ptr(self.rot_sin), self.B, self.F, hw, c, hid // 32, at["scale"], 1e-5, fptr, s), meta, name + ".fused")
            return y
        qkv = self._buf(rows, 3 * hid)
        a = self._base_args(frames, h, w, qkv, 3 * hid, EPI_QKV_ATTN)
        _fill_pass(a.main, x, c, None, 0, h, w, 1, [(0, 0)], at["wqkv"].data)
        a.colsum = _ptr(at["wqkv"].colsum)
        a.ln_eps, a.q_scale = 1e-5, at["scale"]
        if temporal:
            a.rot_cos, a.rot_sin = _ptr(self.rot_cos), _ptr(self.rot_sin)
            a.rot_frames, a.rot_hw = self.F, hw
        self._tapgemm(a, name + ".qkv")
        o = self._buf(rows, hid)
        heads = hid // 32
        if temporal:
            n_tok, stride, bias = self.F, hw, self.pos_bias
        else:
            n_tok, stride, bias = hw, 1, None
        self._add(lambda s, qkv=qkv, o=o, bias=bias: lib.vdmi_attn_core(
            qkv.data_ptr(), o.data_ptr(), rows, heads, n_tok, stride, _ptr(bias), fptr, self.F * hw, s), None, name + ".core")
        y = self._buf(rows, c)
        b = self._base_args(frames, h, w, y, c, EPI_PLAIN)
        _fill_pass(b.main, o, hid, None, 0, h, w, 1, [(0, 0)], at["wout"].data)
        b.has_res, b.res_identity, b.ld_res = 2, _ptr(x), c
        self._tapgemm(b, name + ".out")
        return y

    def _sla(self, name, sl, x, c, h, w) -> torch.Tensor:
        """Residual(PreNorm(SpatialLinearAttention)) (reference vdp.py:454-479) in two passes."""
        lib = _lib.lib()
        frames, hw = self.B * self.F, h * w
        rows = frames * hw
        hid = sl["hid"]
        heads = hid // 32
        if sl.get("fused") is not None:
            y = self._buf(rows, c)
            ws = torch.empty(lib.vdmi_sla_workspace_bytes(frames, hw, c, heads), device=self.device, dtype=torch.uint8)
            self.keep.append(ws)
            wp, wout, bout = sl["fused"], sl["wout_f32"], sl["bout"]
            meta = {"kernel": "sla_fused", "flops": 2.0 * rows * c * 4 * hid + 4.0 * rows * 32 * hid,
                    "bytes": 6.0 * rows * c}
            self._add(lambda s, x=x, y=y, ws=ws: lib.vdmi_sla_fused(
                x.data_ptr(), y.data_ptr(), wp.data_ptr(), wout.data_ptr(), bout.data_ptr(), ws.data_ptr(),
                frames, hw, c, heads, sl["scale"], 1e-5, s), meta, name + ".fused")
            return y
        # (the two-sweep form walks a frame serially in one workgroup: beyond 16x16 the three launches are faster)
        if sl.get("fused_wide") is not None and (hw <= 64 or (c == 256 and SLA_WIDE_LONG and hw <= 256)):
            y = self._buf(rows, c)
            wp, bout = sl["fused_wide"], sl["bout"]
            meta = {"kernel": "sla_fused_wide", "flops": 2.0 * rows * c * 4 * hid + 4.0 * rows * 32 * hid, "bytes": 4.0 * rows * c}
            self._add(lambda s, x=x, y=y: lib.vdmi_sla_fused_wide(
                x.data_ptr(), y.data_ptr(), wp.data_ptr(), bout.data_ptr(), frames, hw, c, heads, sl["scale"], 1e-5, s),
                meta, name + ".fused")
            return y
        qs = self._buf(rows, hid)
        kvt = torch.empty(frames, 2 * hid, hw, device=self.device, dtype=_lib.ACT_DTYPE)
        self.keep.append(kvt)
        a = self._base_args(frames, h, w, qs, 3 * hid, EPI_QKV_SLA)
        _fill_pass(a.main, x, c, None, 0, h, w, 1, [(0, 0)], sl["wqkv"].data)
        a.colsum = _ptr(sl["wqkv"].colsum)
        a.ln_eps, a.q_scale = 1e-5, sl["scale"]
        a.out2, a.sla_hw = _ptr(kvt), hw
        self._tapgemm(a, name + ".qkv")
        bn, bkc = tile_for(EPI_PLAIN, c)
        cpad = (c + bn - 1) // bn * bn
        wf = torch.zeros(frames, hid * cpad, device=self.device, dtype=_lib.ACT_DTYPE)
        self.keep.append(wf)
        wout = sl["wout_f32"]
        self._add(lambda s, kvt=kvt, wf=wf: lib.vdmi_sla_context(
            kvt.data_ptr(), frames, heads, hw, wout.data_ptr(), c, cpad, bkc, wf.data_ptr(), s), None, name + ".ctx")
        y = self._buf(rows, c)
        b = self._base_args(frames, h, w, y, c, EPI_PLAIN)
        _fill_pass(b.main, qs, hid, None, 0, h, w, 1, [(0, 0)], wf, group_stride=hid * cpad, group_frames=1)
        b.bias = _ptr(sl["bout"])
        b.has_res, b.res_identity, b.ld_res = 2, _ptr(x), c
        self._tapgemm(b, name + ".out")
        return y

    def _setup_context(self):
        """Buffers and launches that turn the context [B][L][D] into the packed keys/values of every
        cross-attention layer (reference vdp.py:511-513: to_k / to_v act on the context only)."""
        lib = _lib.lib()
        X = self.eng.packed["xattn"]
        B, L, D, dev = self.B, self.L, X["ctx_dim"], self.device
        nl, hid = X["nl"], X["hid"]
        self.ctx_in = self._io_tensor("ctx_in", (B, L, D), torch.float32)
        self.ctx_keep = self._io_tensor("ctx_keep", (B,), torch.float32, fill=1)
        ctx_rows = self._buf(B * L, D)
        kv_rows = self._buf(B * L, nl * 2 * hid)
        self.kv_layer_elems = lib.vdmi_xattn_kv_elems(1, B, L, hid // 32)
        self.kv_pack = torch.empty(nl * self.kv_layer_elems, device=dev, dtype=_lib.ACT_DTYPE)
        calls, self.calls = self.calls, self.ctx_calls
        self._add(lambda s: lib.vdmi_xattn_context(self.ctx_in.data_ptr(), self.ctx_keep.data_ptr(), ctx_rows.data_ptr(),
                                                   B, L, D, s), None, "xattn.context")
        a = self._base_args(1, 1, B * L, kv_rows, nl * 2 * hid, EPI_PLAIN)
        _fill_pass(a.main, ctx_rows, D, None, 0, 1, B * L, 1, [(0, 0)], X["wkv"].data)
        self._tapgemm(a, "xattn.kv")
        self._add(lambda s: lib.vdmi_xattn_pack_kv(kv_rows.data_ptr(), nl * 2 * hid, self.kv_pack.data_ptr(), nl, B, L,
                                                   hid // 32, s), None, "xattn.pack_kv")
        self.calls = calls

    def _cross_attention(self, name, idx, x, c, h, w) -> torch.Tensor:
        """Residual(PreNorm(CrossAttention)) to the context (reference vdp.py:480-528, 803-805, 824-826)."""
        lib = _lib.lib()
        X = self.eng.packed["xattn"]
        xa, hid = X["layers"][idx], X["hid"]
        frames, hw = self.B * self.F, h * w
        rows = frames * hw
        q = self._buf(rows, hid)
        a = self._base_args(frames, h, w, q, hid, EPI_QKV_ATTN)
        _fill_pass(a.main, x, c, None, 0, h, w, 1, [(0, 0)], xa["wq"].data)
        a.colsum = _ptr(xa["wq"].colsum)
        a.ln_eps, a.q_scale, a.q_cols = 1e-5, X["scale"] * math.log2(math.e), hid
        self._tapgemm(a, name + ".q")
        o = self._buf(rows, hid)
        kvp = self.kv_pack.data_ptr() + idx * self.kv_layer_elems * 2
        nt = (self.L + 31) // 32
        meta = {"kernel": "xattn_core", "flops": 4.0 * rows * hid * nt * 32, "bytes": 4.0 * rows * hid}
        self._add(lambda s, q=q, o=o: lib.vdmi_xattn_core(q.data_ptr(), hid, o.data_ptr(), kvp, self.B, self.F * hw, self.L,
                                                          hid // 32, s), meta, name + ".core")
        y = self._buf(rows, c)
        b = self._base_args(frames, h, w, y, c, EPI_PLAIN)
        _fill_pass(b.main, o, hid, None, 0, h, w, 1, [(0, 0)], xa["wout"].data)
        b.bias = _ptr(xa["bout"])
        b.has_res, b.res_identity, b.ld_res = 2, _ptr(x), c
        self._tapgemm(b, name + ".out")
        return y

    def _down(self, name, dn, x, c, h, w) -> torch.Tensor:
        frames = self.B * self.F
        oh, ow = (h + 2 - 4) // 2 + 1, (w + 2 - 4) // 2 + 1
        y = self._buf(frames * oh * ow, c)
        a = self._base_args(frames, oh, ow, y, c, EPI_PLAIN)
        if dn["s2d"]:
            # 4x4 / stride 2 / pad 1 as 2x2 / stride 1 over the space-to-depth view (4*c channels)
            _fill_pass(a.main, x, 4 * c, None, 0, h, w, 1, [(0, 0), (0, 1), (1, 0), (1, 1)], dn["w"].data)
            a.main.space_to_depth = 1
            a.tile_bn = dn["w"].tile_bn
        else:
            _fill_pass(a.main, x, c, None, 0, h, w, 2, TAPS_4x4_S2, dn["w"].data)
        a.bias = _ptr(dn["b"])
        self._tapgemm(a, name)
        return y

    def _up(self, name, up, x, c, h, w) -> torch.Tensor:
        frames = self.B * self.F
        oh, ow = 2 * h, 2 * w
        y = self._buf(frames * oh * ow, c)
        # the four output parities (p, q) are four 4-tap convolutions of one launch
        taps = [(dy, dx) for p in (0, 1) for q in (0, 1) for (_, dy) in _UP_PAIRS[p] for (_, dx) in _UP_PAIRS[q]]
        a = self._base_args(frames, h, w, y, c, EPI_PLAIN, out_h=oh, out_w=ow)
        a.out_mul, a.out_off_y, a.out_off_x = 2, 0, 0
        _fill_pass(a.main, x, c, None, 0, h, w, 1, taps, up["w"])
        a.main.phases, a.main.w_phase_stride = 4, up["phase_stride"]
        a.bias = _ptr(up["b"])
        self._tapgemm(a, name)
        return y

    # -- the forward graph ------------------------------------------------------------------
    def _setup(self):
        """Constants and the time/condition embedding launches (everything the layers share)."""
        eng, u = self.eng, self.eng.unet
        lib = _lib.lib()
        P = eng.packed
        B, F = self.B, self.F
        dev = self.device

        # constants that depend on F only
        self.pos_bias = eng.pos_bias(F, dev)
        self.rot_cos, self.rot_sin = eng.rotary_tables(F, dev)

        # time / condition embedding -> per-block gain/shift tables
        ed = P["emb_dim"]
        self.emb_silu = torch.empty(B, ed, device=dev, dtype=torch.float32)
        self.emb_raw = torch.empty(B, ed, device=dev, dtype=torch.float32)
        half_total = P["half_total"]
        self.gain = torch.empty(B, half_total, device=dev, dtype=torch.float32)
        self.shift = torch.empty(B, half_total, device=dev, dtype=torch.float32)
        if self.time_table:
            # a sampling loop without a condition vector: the tables of all timesteps exist already, copy this step's rows
            gt, st = eng.time_tables(self.time_table, dev)
            self._add(lambda s: lib.vdmi_gather_scale_shift(
                gt.data_ptr(), st.data_ptr(), self.time_table, self.t_in.data_ptr(), B, half_total,
                self.gain.data_ptr(), self.shift.data_ptr(), s), None, "gather_scale_shift")
            return
        tm = P["time"]
        self._add(lambda s: lib.vdmi_time_embed(
            self.t_in.data_ptr(), B, u.dim, tm["w1"].data_ptr(), tm["b1"].data_ptr(), tm["w2"].data_ptr(),
            tm["b2"].data_ptr(), _ptr(self.cond_in), _ptr(P["null_cond"]), _ptr(self.null_mask),
            u.cond_dim or 0, self.emb_silu.data_ptr(), self.emb_raw.data_ptr(), s), None, "time_embed")
        self._add(lambda s: lib.vdmi_block_scale_shift(
            self.emb_silu.data_ptr(), B, ed, P["w_all"].data_ptr(), P["b_all"].data_ptr(),
            P["gamma_all"].data_ptr(), P["blk_off"].data_ptr(), P["blk_ch"].data_ptr(), P["nblocks"], half_total,
            self.gain.data_ptr(), self.shift.data_ptr(), s), None, "block_scale_shift")

    def _build(self):
        eng, u = self.eng, self.eng.unet
        lib = _lib.lib()
        P = eng.packed
        B, F, H, W = self.B, self.F, self.H, self.W
        if self.L:
            self._setup_context()
        frames = B * F
        dims = u.dims

        # init conv + init temporal attention
        ic = P["init_conv"]
        x = self._buf(frames * H * W, dims[0])
        self._add(lambda s, x=x: lib.vdmi_init_conv(
            self.x_in.data_ptr(), ic["w"].data_ptr(), ic["b"].data_ptr(), x.data_ptr(), B, u.channels, F, H, W,
            dims[0], u.init_kernel_size, s), None, "init_conv")
        self.taps["init_conv"] = x
        x = self._attention("init_temporal_attn", P["init_temporal_attn"], x, dims[0], H, W, True)
        self.taps["init_temporal_attn"] = x
        r = x

        h, w = H, W
        skips = []
        nres = len(dims) - 1
        c = dims[0]
        for i in range(nres):
            L = P["downs"][i]
            x = self._resnet(f"downs.{i}.0", L["rb0"], [(x, c)], h, w)
            c = dims[i + 1]
            x = self._resnet(f"downs.{i}.1", L["rb1"], [(x, c)], h, w)
            if L["sla"] is not None:
                x = (self._sta if L["sla"].get("sta") else self._sla)(f"downs.{i}.2", L["sla"], x, c, h, w)
            x = self._attention(f"downs.{i}.3", L["tattn"], x, c, h, w, True, focus=True)
            if self.L:
                x = self._cross_attention(f"frame_cross_attns_down.{i}", i, x, c, h, w)
            self.taps[f"downs.{i}"] = x
            skips.append((x, c, h, w))
            if L["down"] is not None:
                x = self._down(f"downs.{i}.4", L["down"], x, c, h, w)
                h, w = (h + 2 - 4) // 2 + 1, (w + 2 - 4) // 2 + 1

        x = self._resnet("mid_block1", P["mid_block1"], [(x, c)], h, w)
        x = self._sta("mid_spatial_attn", P["mid_spatial_attn"], x, c, h, w, per_frame=True)
        x = self._attention("mid_temporal_attn", P["mid_temporal_attn"], x, c, h, w, True, focus=True)
        x = self._resnet("mid_block2", P["mid_block2"], [(x, c)], h, w)
        self.taps["mid"] = x

        for i in range(nres):
            L = P["ups"][i]
            sk, sc, sh, sw = skips.pop()
            if (sh, sw) != (h, w):
                raise _lib.VdmiError(
                    f"skip connection {sh}x{sw} does not match the upsampled map {h}x{w}: "
                    "H and W must be divisible by 2**(levels-1)")
            x = self._resnet(f"ups.{i}.0", L["rb0"], [(x, c), (sk, sc)], h, w)
            c = L["rb0"]["w1"].n
            x = self._resnet(f"ups.{i}.1", L["rb1"], [(x, c)], h, w)
            if L["sla"] is not None:
                x = (self._sta if L["sla"].get("sta") else self._sla)(f"ups.{i}.2", L["sla"], x, c, h, w)
            x = self._attention(f"ups.{i}.3", L["tattn"], x, c, h, w, True, focus=True)
            if self.L:
                x = self._cross_attention(f"frame_cross_attns_up.{i}", nres + i, x, c, h, w)
            self.taps[f"ups.{i}"] = x
            if L["up"] is not None:
                x = self._up(f"ups.{i}.4", L["up"], x, c, h, w)
                h, w = 2 * h, 2 * w

        x = self._resnet("final_conv.0", P["final_rb"], [(x, c), (r, dims[0])], h, w)
        self.taps["final_conv.0"] = x
        fc = P["final_conv"]
        self._add(lambda s, x=x: lib.vdmi_final_conv(
            x.data_ptr(), fc["w"].data_ptr(), fc["b"].data_ptr(), self.eps_out.data_ptr(), B, u.dim, F, H, W,
            u.out_dim, s), None, "final_conv.1")

    def run_context(self):
        """Project ctx_in (masked by ctx_keep) to the packed keys/values of all cross-attention layers."""
        s = _stream()
        for call in self.ctx_calls:
            st = call.fn(s)
            if st != 0:
                check(st, call.name)

    def run(self, fork: bool = True):
        """Enqueue every launch (no host synchronisation).  Launches flagged `side` (the time-embedding kernels, which
        do not depend on the image; the res_conv GEMMs of the wide layers, which depend on the block input only) go
        to a side stream; a launch flagged `join` first waits for it.  Inside a captured HIP graph the fork / join
        become graph edges.  fork=False keeps every launch on the current stream, in order."""
        main = torch.cuda.current_stream(self.device)
        forked = False
        for call in self.calls:
            if call.join and forked:
                main.wait_stream(self._side)
                forked = False
            stream = main
            if call.side and fork:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                if not forked:
                    self._side.wait_stream(main)
                    forked = True
                stream = self._side
            st = call.fn(stream.cuda_stream)
            if st != 0:
                check(st, call.name)
        if forked:
            main.wait_stream(self._side)


# channel widths whose temporal attention runs as the one fused kernel (VDMI_TATTN_256=0: 256 stays on the three-launch path)
FUSED_TATTN_CHANNELS = (32, 64, 128) + ((256,) if os.environ.get("VDMI_TATTN_256", "1") != "0" else ()) + \
    ((512,) if os.environ.get("VDMI_TATTN_WIDE", "1") != "0" else ())
# 256-channel temporal attention on images of fewer positions than this runs on the wide fused kernel (0: never).
# Measured at 8 samples per launch: 8x8 30 (three launches) -> 13.6 us, 16x16 36 (tattn_fused) -> 28 us.
TATTN_WIDE_BELOW_HW = int(os.environ.get("VDMI_TATTN_WIDE_BELOW_HW", str(1 << 30))) if os.environ.get("VDMI_TATTN_WIDE", "1") != "0" else 0
# wide-layer convs that would launch fewer workgroups than this split their contraction (0: never)
K_SPLIT_BELOW = int(os.environ.get("VDMI_K_SPLIT_BELOW", "256"))
K_SPLIT_TARGET = int(os.environ.get("VDMI_K_SPLIT_TARGET", "320"))      # workgroups per launch the split aims at
K_SPLIT_ODD = os.environ.get("VDMI_K_SPLIT_ODD", "0") != "0"           # allow a three-way split (unequal shares of the chunks)
# SpatialLinearAttention at 256 channels on frames of more than 64 positions: the one-launch two-sweep kernel
SLA_WIDE_LONG = os.environ.get("VDMI_SLA_WIDE_LONG", "1") != "0"
SPLIT_LONG_K = int(os.environ.get("VDMI_SPLIT_LONG_K", "512"))      # input channels from which such a block takes the split form (0: never)
BATCH_PARTS = int(os.environ.get("VDMI_BATCH_PARTS", "2"))
BATCH_PARTS_MIN = int(os.environ.get("VDMI_BATCH_PARTS_MIN", "4"))     # smallest batch that is split


def batch_parts(B: int) -> int:
    """Number of independent sub-batches a forward of B samples is run as (1: not split)."""
    n = BATCH_PARTS
    return n if n > 1 and B >= BATCH_PARTS_MIN and B >= 2 * n else 1


class SplitPlan:
    """One forward run as `parts` independent sub-batch plans on concurrent HIP streams.

    Samples never interact in Unet3D (reference vdp.py:756-831: every norm, attention and conv is per sample), so the
    result is the same bit for bit; the point is occupancy: between two dependent launches of one stream the device
    idles for a few microseconds, the tail of every launch leaves compute units empty, and the deep, small-image
    layers launch too few workgroups for 256 of them - a second stream fills all three.  The owner sees one set of
    input / output tensors (x_in, t_in, eps_out ...); the sub-plans work on views of them.  Under graph capture the
    fork / join become graph edges.  The sub-plans keep all their launches on their own stream (a fork from an already
    forked stream does not survive graph capture), and the streams stay within the four hardware queues of a process:
    with more, HIP serialises streams and the overlap is lost."""

    def __init__(self, eng: "UnetEngine", B, F, H, W, device, parts: int, L: int = 0, focus: bool = False,
                 time_table: int = 0):
        if parts < 1 or B < parts:
            raise ValueError(f"batch {B} does not split into {parts} parts")
        u = eng.unet
        self.eng, self.B, self.F, self.H, self.W, self.device, self.L = eng, B, F, H, W, device, L
        self.focus_enabled = focus
        f32 = dict(device=device, dtype=torch.float32)
        self.x_in = torch.zeros(B, u.channels, F, H, W, **f32)
        self.t_in = torch.zeros(B, device=device, dtype=torch.int64)
        self.cond_in = torch.zeros(B, u.cond_dim, **f32) if u.has_cond else None
        self.null_mask = torch.zeros(B, device=device, dtype=torch.uint8) if u.has_cond else None
        self.focus_in = torch.zeros(B, device=device, dtype=torch.uint8)
        self.eps_out = torch.empty(B, u.out_dim, F, H, W, **f32)
        names = ["x_in", "t_in", "cond_in", "null_mask", "focus_in", "eps_out"]
        if L:
            self.ctx_in = torch.zeros(B, L, eng.packed["xattn"]["ctx_dim"], **f32)
            self.ctx_keep = torch.ones(B, **f32)
            names += ["ctx_in", "ctx_keep"]
        self.parts: List[Plan] = []
        start = 0
        for i in range(parts):       # near-equal parts; the first B % parts take one sample more
            n = B // parts + (1 if i < B % parts else 0)
            io = {k: getattr(self, k)[start:start + n] for k in names if getattr(self, k) is not None}
            self.parts.append(eng.plan_class(eng, n, F, H, W, device, L=L, focus=focus, io=io, time_table=time_table))
            start += n
        self._streams = [torch.cuda.Stream(device=device) for _ in range(parts)]
        self.calls = [c for p in self.parts for c in p.calls]              # flat list, for per-launch timing
        self.ctx_calls = [c for p in self.parts for c in p.ctx_calls]

    def _fan_out(self, fn):
        main = torch.cuda.current_stream(self.device)
        for p, s in zip(self.parts, self._streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                fn(p)
        for s in self._streams:
            main.wait_stream(s)

    def run_context(self):
        self._fan_out(lambda p: p.run_context())

    def run(self, fork: bool = True):
        if fork:
            self._fan_out(lambda p: p.run(fork=False))
        else:
            for p in self.parts:
                p.run(fork=False)


def values_fingerprint(params):
    """Per tensor its L2 norm and its dot product with a fixed ramp (1 .. 2 over the flattened tensor), as one tuple of
    host floats.  The ramp makes the fingerprint position-sensitive: sign flips, swapped or permuted rows and other
    norm-preserving rewrites of a tensor change it.  One device->host copy: not for use under graph capture."""
    ps = [p.detach() for p in params]
    if not ps:
        return ()
    with torch.no_grad():
        vals = list(torch._foreach_norm(ps, 2))
        for p in ps:
            flat = p.reshape(-1).float()
            vals.append(torch.dot(flat, torch.linspace(1.0, 2.0, flat.numel(), device=flat.device)))
        return tuple(torch.stack([v.double() for v in vals]).cpu().tolist())


class UnetEngine:
    plan_class = Plan

    def __init__(self, unet):
        self.unet = unet
        self.packed = None
        self._packed_key = None
        self._packed_values = None
        self._generation = 0
        self.pack_id = 0                      # advances at every repack: owners of captured graphs key their caches on it
        self._plans: Dict[Tuple, Plan] = {}
        self._const_cache: Dict[Tuple, object] = {}

    # -- parameter packing -----------------------------------------------------------------
    def _rotary(self):
        pre = self.unet.init_temporal_attn.fn
        return (pre.fn.fn if hasattr(pre.fn, "fn") else pre.fn).rotary_emb

    def _param_key(self, device):
        rot = self._rotary()
        return (str(device), getattr(rot, "convention", "half"), self._generation) + \
            tuple((p.data_ptr(), p._version) for p in self.unet.parameters())

    def invalidate(self):
        """The module's weights were changed in a way autograd's version counters do not see (writes through
        `.data`): repack them before the next forward."""
        self._generation += 1

    def _values_key(self):
        """Fingerprint of the parameter VALUES (values_fingerprint below: a handful of launches per tensor and one
        device->host copy).  Version counters miss in-place writes through `.data` (`p.data.copy_`, `p.data.mul_`,
        the reference's EMA `ma.data = ...` when the allocator hands back the same pointer); the sampling loops,
        which run thousands of launches on the packed copy, check this once per loop."""
        return values_fingerprint(self.unet.parameters())

    def ensure_packed(self, device, check_values: bool = False):
        """check_values: also compare the parameter values' fingerprint with the packed copy's (costs a host
        synchronisation; not for use under graph capture)."""
        key = self._param_key(device)
        stale = key != self._packed_key
        vals = None
        if not stale and check_values:
            vals = self._values_key()
            stale = vals != self._packed_values
        if stale:
            self.packed = self._pack(device)
            self._packed_key = key
            self._packed_values = self._values_key() if vals is None else vals
            self.pack_id += 1
            self._plans.clear()
            self._const_cache.clear()

    @staticmethod
    def _conv_w(w: torch.Tensor) -> torch.Tensor:
        """Conv3d weight [n][cin][1][kh][kw] -> [n][cin][kh*kw] float32 (CPU)."""
        n, cin = w.shape[0], w.shape[1]
        return w.detach().float().cpu().reshape(n, cin, -1)

    def _pack_resnet(self, rb, device, cin_split, stacked):
        cin0, cin1 = cin_split
        d = {}
        # wide layers (small images): split-channel conv + separate norm kernel, see Plan._block_split
        split = rb.block1.proj.weight.shape[0] >= SPLIT_MIN_CHANNELS
        epi, tbn = (EPI_PLAIN, 64) if split else (EPI_NORM_SILU, 0)
        d["w1"] = PackedTaps(self._conv_w(rb.block1.proj.weight), cin0, cin1, epi, device, tile_bn=tbn)
        d["b1"] = rb.block1.proj.bias.detach().float().to(device).contiguous()
        cmid = d["w1"].n
        d["w2"] = PackedTaps(self._conv_w(rb.block2.proj.weight), cmid, 0, epi, device, tile_bn=tbn)
        d["b2"] = rb.block2.proj.bias.detach().float().to(device).contiguous()
        d["gamma1"] = rb.block1.norm.gamma.detach().float().reshape(-1).to(device).contiguous()
        d["gamma2"] = rb.block2.norm.gamma.detach().float().reshape(-1).to(device).contiguous()
        if isinstance(rb.res_conv, torch.nn.Conv3d):
            d["wres"] = PackedTaps(self._conv_w(rb.res_conv.weight), cin0, cin1, epi, device)
            d["bres"] = rb.res_conv.bias.detach().float().to(device).contiguous()
        else:
            d["wres"], d["bres"] = None, None
        tc = getattr(rb, "temporal_conv", None)      # version1diffusion.py:308: (3,1,1) conv after block2
        if tc is not None:
            d["wt"] = PackedTaps(tc.weight.detach().float().cpu().reshape(cmid, cmid, 3), cmid, 0, EPI_PLAIN, device)
            d["bt"] = tc.bias.detach().float().to(device).contiguous()
        if rb.mlp is not None:
            d["ss_off"] = stacked["half"]
            stacked["w"].append(rb.mlp[1].weight.detach().float().cpu())
            stacked["b"].append(rb.mlp[1].bias.detach().float().cpu())
            stacked["g"].append(rb.block1.norm.gamma.detach().float().reshape(-1).cpu())
            stacked["off"].append(stacked["half"])
            stacked["ch"].append(cmid)
            stacked["half"] += cmid
        else:
            d["ss_off"] = None
        return d

    def _pack_attn(self, res_mod, device, dim_head):
        """Residual(PreNorm(EinopsToAndFrom(Attention)))"""
        pre = res_mod.fn
        att = pre.fn.fn if hasattr(pre.fn, "fn") else pre.fn       # with or without the EinopsToAndFrom wrapper
        gamma = pre.norm.gamma.detach().float().reshape(-1).cpu()
        wq = att.to_qkv.weight.detach().float().cpu() * gamma[None, :]        # LayerNorm gamma folded in
        hid = att.to_out.weight.shape[1]
        cdim = wq.shape[1]
        if att.rotary_emb is not None and getattr(att.rotary_emb, "convention", "half") == "interleaved":
            # rot_interleaved(x) = P^-1 rot_half(P x) with P the even/odd de-interleave of a head's features:
            # apply P to the q and k rows (v and the scores are unaffected)
            dh = getattr(att, "dim_head", 32)
            perm = torch.cat([torch.arange(0, dh, 2), torch.arange(1, dh, 2)])
            idx = torch.arange(3 * hid)
            for blk in (0, 1):
                for hd in range(hid // dh):
                    base = blk * hid + hd * dh
                    idx[base:base + dh] = base + perm
            wq = wq[idx]
        fused = None
        if cdim in FUSED_TATTN_CHANNELS and att.rotary_emb is not None:
            lib = _lib.lib()
            buf = torch.empty(lib.vdmi_pack_tattn_elems(cdim, hid // 32), dtype=torch.int16)
            wo = att.to_out.weight.detach().float().cpu().contiguous()
            wqc = wq.contiguous()
            check(lib.vdmi_pack_tattn(wqc.data_ptr(), wo.data_ptr(), cdim, hid // 32, buf.data_ptr()), "vdmi_pack_tattn")
            fused = buf.view(_lib.ACT_DTYPE).to(device)
        return {
            "fused": fused,
            "wqkv": PackedTaps(wq[:, :, None], wq.shape[1], 0, EPI_QKV_ATTN, device, want_colsum=True),
            "wout": PackedTaps(att.to_out.weight.detach().float().cpu()[:, :, None], hid, 0, EPI_PLAIN, device),
            "hid": hid,
            "scale": float(dim_head) ** -0.5,
        }

    def _pack_xattn(self, device):
        """frame_cross_attns_down / _up: Residual(PreNorm(CrossAttention)).  The key/value projections of all
        layers are stacked into one row GEMM over the context; the out-projection's input columns follow
        the feature order vdmi_xattn_core writes."""
        u = self.unet
        mods = list(u.frame_cross_attns_down) + list(u.frame_cross_attns_up)
        hid = mods[0].fn.fn.to_q.weight.shape[0]
        # column 32*head + 16*h + r of the core's output holds feature 32*head + (r&3) + 8*(r>>2) + 4*h
        perm = torch.tensor([32 * hd + (r & 3) + 8 * (r >> 2) + 4 * hh
                             for hd in range(hid // 32) for hh in (0, 1) for r in range(16)])
        layers, wkv = [], []
        for m in mods:
            pre, ca = m.fn, m.fn.fn
            gamma = pre.norm.gamma.detach().float().reshape(-1).cpu()
            wq = ca.to_q.weight.detach().float().cpu() * gamma[None, :]
            wo = ca.to_out.weight.detach().float().cpu()[:, perm]
            layers.append({"wq": PackedTaps(wq[:, :, None], wq.shape[1], 0, EPI_QKV_ATTN, device, want_colsum=True),
                           "wout": PackedTaps(wo[:, :, None], hid, 0, EPI_PLAIN, device),
                           "bout": ca.to_out.bias.detach().float().to(device).contiguous()})
            wkv += [ca.to_k.weight.detach().float().cpu(), ca.to_v.weight.detach().float().cpu()]
        wkv = torch.cat(wkv, 0)
        return {"layers": layers, "nl": len(mods), "hid": hid, "ctx_dim": wkv.shape[1],
                "scale": float(mods[0].fn.fn.scale),
                "wkv": PackedTaps(wkv[:, :, None], wkv.shape[1], 0, EPI_PLAIN, device)}

    def _pack_sta(self, res_mod, device):
        """Residual(PreNorm(SpatioTemporalAttention)): q|k|v projection with the LayerNorm gamma folded in; the
        out-projection's input columns follow the feature order vdmi_xattn_core writes."""
        pre = res_mod.fn
        st = pre.fn.fn if hasattr(pre.fn, "fn") else pre.fn       # with or without the EinopsToAndFrom wrapper
        gamma = pre.norm.gamma.detach().float().reshape(-1).cpu()
        wq = st.to_qkv.weight.detach().float().cpu() * gamma[None, :]
        hid = st.to_out.weight.shape[1]
        perm = torch.tensor([32 * hd + (r & 3) + 8 * (r >> 2) + 4 * hh
                             for hd in range(hid // 32) for hh in (0, 1) for r in range(16)])
        wo = st.to_out.weight.detach().float().cpu()[:, perm]
        fused = None
        cdim = wq.shape[1]
        if cdim in (256, 512) and hid == 256 and TATTN_WIDE_BELOW_HW:
            # per-frame attention over at most 64 positions: the one-launch kernel (weights in vdmi_pack_tattn order)
            lib = _lib.lib()
            buf = torch.empty(lib.vdmi_pack_tattn_elems(cdim, hid // 32), dtype=torch.int16)
            wqc, won = wq.contiguous(), st.to_out.weight.detach().float().cpu().contiguous()
            check(lib.vdmi_pack_tattn(wqc.data_ptr(), won.data_ptr(), cdim, hid // 32, buf.data_ptr()), "vdmi_pack_tattn")
            fused = buf.view(_lib.ACT_DTYPE).to(device)
        return {"sta": True, "hid": hid, "scale": float(st.scale), "fused": fused,
                "wqkv": PackedTaps(wq[:, :, None], wq.shape[1], 0, EPI_QKV_ATTN, device, want_colsum=True),
                "wout": PackedTaps(wo[:, :, None], hid, 0, EPI_PLAIN, device)}

    def _pack_sla(self, res_mod, device):
        pre = res_mod.fn
        sl = pre.fn
        if not hasattr(sl, "to_qkv") or isinstance(sl.to_qkv, torch.nn.Linear):
            return self._pack_sta(res_mod, device)
        gamma = pre.norm.gamma.detach().float().reshape(-1).cpu()
        wq = sl.to_qkv.weight.detach().float().cpu().reshape(sl.to_qkv.weight.shape[0], -1) * gamma[None, :]
        hid = sl.to_out.weight.shape[1]
        cdim = wq.shape[1]
        fused = None
        if cdim in (32, 64, 128) and hid // 32 <= 8:
            lib = _lib.lib()
            buf = torch.empty(lib.vdmi_pack_sla_elems(cdim, hid // 32), dtype=torch.int16)
            wqc = wq.contiguous()
            check(lib.vdmi_pack_sla(wqc.data_ptr(), cdim, hid // 32, buf.data_ptr()), "vdmi_pack_sla")
            fused = buf.view(_lib.ACT_DTYPE).to(device)
        fused_wide = None
        if cdim in (256, 512) and hid == 256 and TATTN_WIDE_BELOW_HW:
            # frames of at most 64 positions: the one-launch kernel (weights in vdmi_pack_tattn order)
            lib = _lib.lib()
            buf = torch.empty(lib.vdmi_pack_tattn_elems(cdim, hid // 32), dtype=torch.int16)
            wqc = wq.contiguous()
            won = sl.to_out.weight.detach().float().cpu().reshape(sl.to_out.weight.shape[0], -1).contiguous()
            check(lib.vdmi_pack_tattn(wqc.data_ptr(), won.data_ptr(), cdim, hid // 32, buf.data_ptr()), "vdmi_pack_tattn")
            fused_wide = buf.view(_lib.ACT_DTYPE).to(device)
        return {
            "fused": fused,
            "fused_wide": fused_wide,
            "wqkv": PackedTaps(wq[:, :, None], wq.shape[1], 0, EPI_QKV_SLA, device, want_colsum=True),
            "wout_f32": sl.to_out.weight.detach().float().reshape(sl.to_out.weight.shape[0], -1).to(device).contiguous(),
            "bout": sl.to_out.bias.detach().float().to(device).contiguous(),
            "hid": hid,
            "scale": float(sl.dim_head) ** -0.5,
        }

    def _pack_down(self, conv, device):
        bias = conv.bias.detach().float().to(device).contiguous()
        cin = conv.weight.shape[1]
        if cin % 64 == 0 and tuple(conv.weight.shape[2:]) == (1, 4, 4):
            # space-to-depth form: channel (2a+b)*cin + ci, tap (dY, dX)  <-  kernel element (2dY+a, 2dX+b)
            w = conv.weight.detach().float().cpu()[:, :, 0]                  # [n][cin][4][4]
            subs = []
            for a in (0, 1):
                for b in (0, 1):
                    subs.append(torch.stack([w[:, :, 2 * dy + a, 2 * dx + b] for dy in (0, 1) for dx in (0, 1)], dim=-1))
            w4 = torch.cat(subs, dim=1)                                       # [n][4*cin][4]
            return {"w": PackedTaps(w4, 4 * cin, 0, EPI_PLAIN, device, tile_bn=64), "b": bias, "s2d": True}
        return {"w": PackedTaps(self._conv_w(conv.weight), cin, 0, EPI_PLAIN, device), "b": bias, "s2d": False}

    def _pack_up(self, convt, device):
        # ConvTranspose3d weight [cin][cout][1][4][4]; one 4-tap weight set per output parity (p, q)
        w = convt.weight.detach().float().cpu()[:, :, 0]                     # [cin][cout][4][4]
        cin = w.shape[0]
        packs = []
        for p in (0, 1):
            for q in (0, 1):
                taps = [w[:, :, ky, kx].t() for (ky, _) in _UP_PAIRS[p] for (kx, _) in _UP_PAIRS[q]]   # each [cout][cin]
                packs.append(PackedTaps(torch.stack(taps, dim=-1), cin, 0, EPI_PLAIN, device).data)
        return {"w": torch.cat(packs).contiguous(), "phase_stride": packs[0].numel(),
                "b": convt.bias.detach().float().to(device).contiguous()}

    def _pack(self, device):
        u = self.unet
        for d_ in u.dims:
            if d_ % 32:
                raise _lib.VdmiError(f"channel widths must be multiples of 32 for the gfx950 kernels, got {u.dims}")
        if u.attn_dim_head != 32:
            raise _lib.VdmiError("the gfx950 attention kernels are built for attn_dim_head = 32")
        f32 = lambda t: t.detach().float().to(device).contiguous()
        P = {}
        stacked = {"w": [], "b": [], "g": [], "off": [], "ch": [], "half": 0}
        dims = u.dims
        P["init_conv"] = {"w": f32(u.init_conv.weight[:, :, 0]), "b": f32(u.init_conv.bias)}
        P["init_temporal_attn"] = self._pack_attn(u.init_temporal_attn, device, u.attn_dim_head)
        P["time"] = {"w1": f32(u.time_mlp[1].weight), "b1": f32(u.time_mlp[1].bias),
                     "w2": f32(u.time_mlp[3].weight), "b2": f32(u.time_mlp[3].bias)}
        P["null_cond"] = f32(u.null_cond_emb.reshape(-1)) if u.has_cond else None
        P["emb_dim"] = u.dim * 8 + (u.cond_dim or 0)

        P["downs"] = []
        for i, (b1, b2, sla, tat, down) in enumerate(u.downs):
            L = {"rb0": self._pack_resnet(b1, device, (dims[i], 0), stacked),
                 "rb1": self._pack_resnet(b2, device, (dims[i + 1], 0), stacked),
                 "sla": self._pack_sla(sla, device) if u.use_sparse_linear_attn else None,
                 "tattn": self._pack_attn(tat, device, u.attn_dim_head),
                 "down": self._pack_down(down, device) if isinstance(down, torch.nn.Conv3d) else None}
            P["downs"].append(L)
        mid = dims[-1]
        P["mid_block1"] = self._pack_resnet(u.mid_block1, device, (mid, 0), stacked)
        P["mid_spatial_attn"] = self._pack_sta(u.mid_spatial_attn, device)     # MFMA attention core, one group per frame
        P["mid_temporal_attn"] = self._pack_attn(u.mid_temporal_attn, device, u.attn_dim_head)
        P["mid_block2"] = self._pack_resnet(u.mid_block2, device, (mid, 0), stacked)
        P["ups"] = []
        in_out = list(zip(dims[:-1], dims[1:]))
        for i, ((dim_in, dim_out), (b1, b2, sla, tat, up)) in enumerate(zip(reversed(in_out), u.ups)):
            L = {"rb0": self._pack_resnet(b1, device, (dim_out, dim_out), stacked),
                 "rb1": self._pack_resnet(b2, device, (dim_in, 0), stacked),
                 "sla": self._pack_sla(sla, device) if u.use_sparse_linear_attn else None,
                 "tattn": self._pack_attn(tat, device, u.attn_dim_head),
                 "up": self._pack_up(up, device) if isinstance(up, torch.nn.ConvTranspose3d) else None}
            P["ups"].append(L)
        P["final_rb"] = self._pack_resnet(u.final_conv[0], device, (dims[0], dims[0]), stacked)
        fcw = u.final_conv[1].weight
        P["final_conv"] = {"w": f32(fcw.reshape(fcw.shape[0], -1)), "b": f32(u.final_conv[1].bias)}

        P["xattn"] = self._pack_xattn(device) if u.has_frame_cross_attn else None

        P["w_all"] = torch.cat(stacked["w"], 0).to(device).contiguous()
        P["b_all"] = torch.cat(stacked["b"], 0).to(device).contiguous()
        P["gamma_all"] = torch.cat(stacked["g"], 0).to(device).contiguous()
        P["blk_off"] = torch.tensor(stacked["off"], dtype=torch.int32, device=device)
        P["blk_ch"] = torch.tensor(stacked["ch"], dtype=torch.int32, device=device)
        P["nblocks"] = len(stacked["off"])
        P["half_total"] = stacked["half"]
        return P

    # -- small constants ---------------------------------------------------------------------
    def pos_bias(self, nf: int, device) -> torch.Tensor:
        """RelativePositionBias table [heads][F][F] (reference vdp.py:274-312): integer buckets from
        the library's host routine, gathered from the embedding weight."""
        key = ("pos_bias", nf)
        if key not in self._const_cache:
            rpb = self.unet.time_rel_pos_bias
            buckets = np.empty((nf, nf), dtype=np.int32)
            check(_lib.lib().vdmi_relpos_buckets(nf, rpb.num_buckets, rpb.max_distance, buckets.ctypes.data),
                  "vdmi_relpos_buckets")
            idx = torch.from_numpy(buckets.astype(np.int64)).to(device)
            wt = rpb.relative_attention_bias.weight.detach().float().to(device)
            self._const_cache[key] = wt[idx].permute(2, 0, 1).contiguous()
        return self._const_cache[key]

    def rotary_tables(self, nf: int, device):
        """cos/sin of position * inv_freq, [F][16] float32 (half-split rotary, 32-dim heads)."""
        key = ("rot", nf)
        if key not in self._const_cache:
            inv = self._rotary().inv_freq.detach().float().cpu()
            ang = torch.arange(nf).float()[:, None] * inv[None, :]
            self._const_cache[key] = (ang.cos().to(device).contiguous(), ang.sin().to(device).contiguous())
        return self._const_cache[key]

    # -- execution -------------------------------------------------------------------------------
    def time_tables(self, T: int, device):
        """gain / shift rows of every timestep 0..T-1, float [T][half_total] each: the same two kernels a step would
        run, once over b = T rows (reference vdp.py:339-351, 432-448, 651-656; only valid without a condition vector)."""
        self.ensure_packed(device)
        key = ("time_tables", T, str(device))
        if key not in self._const_cache:
            if self.unet.has_cond:
                raise _lib.VdmiError("timestep tables need a model without a condition vector")
            lib, P, u = _lib.lib(), self.packed, self.unet
            ed, half_total, tm = P["emb_dim"], P["half_total"], P["time"]
            t_all = torch.arange(T, device=device, dtype=torch.int64)
            silu = torch.empty(T, ed, device=device, dtype=torch.float32)
            raw = torch.empty(T, ed, device=device, dtype=torch.float32)
            gain = torch.empty(T, half_total, device=device, dtype=torch.float32)
            shift = torch.empty(T, half_total, device=device, dtype=torch.float32)
            s = _stream()
            check(lib.vdmi_time_embed(t_all.data_ptr(), T, u.dim, tm["w1"].data_ptr(), tm["b1"].data_ptr(),
                                      tm["w2"].data_ptr(), tm["b2"].data_ptr(), None, None, None, 0,
                                      silu.data_ptr(), raw.data_ptr(), s), "vdmi_time_embed")
            check(lib.vdmi_block_scale_shift(silu.data_ptr(), T, ed, P["w_all"].data_ptr(), P["b_all"].data_ptr(),
                                             P["gamma_all"].data_ptr(), P["blk_off"].data_ptr(), P["blk_ch"].data_ptr(),
                                             P["nblocks"], half_total, gain.data_ptr(), shift.data_ptr(), s),
                  "vdmi_block_scale_shift")
            torch.cuda.current_stream(device).synchronize()      # t_all, silu, raw go out of scope
            self._const_cache[key] = (gain, shift)
        return self._const_cache[key]

    def plan(self, B, F, H, W, device, L: int = 0, focus: bool = False, time_table: int = 0) -> Plan:
        """L > 0: with the frame cross-attention layers, for contexts of L tokens per sample.  focus: the temporal
        attentions honour the focus-present flags (a separate plan: without them the fused kernel folds bias and
        mask into its score product).  time_table = T > 0 (sampling loops; ignored for models with a condition vector):
        the per-step time-embedding launches are replaced by a row copy from the tables of all T timesteps, so
        t_in must stay within [0, T)."""
        self.ensure_packed(device)
        if self.unet.has_cond or not TIME_TABLES:
            time_table = 0
        key = (B, F, H, W, str(device), L, focus, time_table)
        if key not in self._plans:
            parts = batch_parts(B)
            # plans are cached and reused by callers in and out of inference mode: their buffers must be ordinary
            # tensors (an inference tensor cannot be written outside inference mode)
            with torch.inference_mode(False):
                if parts > 1:
                    self._plans[key] = SplitPlan(self, B, F, H, W, device, parts, L=L, focus=focus, time_table=time_table)
                else:
                    self._plans[key] = self.plan_class(self, B, F, H, W, device, L=L, focus=focus, time_table=time_table)
        return self._plans[key]

    @torch.no_grad()
    def forward(self, x: torch.Tensor, time: torch.Tensor, cond=None, null_cond_prob: float = 0.,
                frame_cond_features=None, null_frame_cond_prob: float = 0., focus_present_mask=None,
                prob_focus_present: float = 0., null_cond_mask=None) -> torch.Tensor:
        """null_cond_mask (bool / uint8 [B]): which samples take the null condition embedding, instead of drawing the
        mask with probability null_cond_prob (a guided evaluation runs its two branches as one doubled batch)."""
        if not x.is_cuda:
            raise _lib.VdmiError("Unet3D runs on the HIP device only (no CPU fallback); move the module and inputs to 'cuda'")
        u = self.unet
        B, Cc, F, H, W = x.shape
        if Cc != u.channels:
            raise ValueError(f"expected {u.channels} input channels, got {Cc}")
        L = 0
        if frame_cond_features is not None and u.has_frame_cross_attn:
            if frame_cond_features.dim() != 3 or frame_cond_features.shape[0] != B:
                raise ValueError("frame_cond_features must be (batch, tokens, frame_cond_dim)")
            L = frame_cond_features.shape[1]
        use_focus = focus_present_mask is not None or prob_focus_present != 0
        plan = self.plan(B, F, H, W, x.device, L, focus=use_focus)
        plan.x_in.copy_(x.to(torch.float32))
        plan.t_in.copy_(time.to(torch.int64))
        if L:
            if frame_cond_features.shape[2] != plan.ctx_in.shape[2]:
                raise ValueError(f"frame_cond_features has {frame_cond_features.shape[2]} features, "
                                 f"the model was built with frame_cond_dim={plan.ctx_in.shape[2]}")
            plan.ctx_in.copy_(frame_cond_features.to(torch.float32))
            if null_frame_cond_prob > 0:    # reference vdp.py:783-789; prob_mask_like, vdp.py:68-74
                if null_frame_cond_prob == 1:
                    plan.ctx_keep.fill_(0)
                else:
                    plan.ctx_keep.copy_((torch.rand(B, device=x.device) >= null_frame_cond_prob).to(torch.float32))
            else:
                plan.ctx_keep.fill_(1)
            plan.run_context()
        # reference vdp.py:769: the mask defaults to prob_mask_like((batch,), prob_focus_present)
        if focus_present_mask is not None:
            plan.focus_in.copy_(focus_present_mask.to(device=x.device, dtype=torch.uint8))
        elif prob_focus_present == 1:
            plan.focus_in.fill_(1)
        elif prob_focus_present == 0:
            plan.focus_in.fill_(0)
        else:
            plan.focus_in.copy_((torch.rand(B, device=x.device) < prob_focus_present).to(torch.uint8))
        if u.has_cond:
            plan.cond_in.copy_(cond.to(torch.float32))
            if null_cond_mask is not None:
                plan.null_mask.copy_(null_cond_mask.to(device=x.device, dtype=torch.uint8))
            elif null_cond_prob == 1:
                plan.null_mask.fill_(1)
            elif null_cond_prob == 0:
                plan.null_mask.fill_(0)
            else:   # prob_mask_like (reference vdp.py:68-74)
                plan.null_mask.copy_((torch.rand(B, device=x.device) < null_cond_prob).to(torch.uint8))
        plan.run()
        return plan.eps_out.clone()
