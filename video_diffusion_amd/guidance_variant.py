"""The reference's `video_diffusion_guidance.py` variant of the denoiser on the HIP kernels.

Differences from the packaged model (reference file `video_diffusion_guidance.py`):
* ResnetBlock is two TemporalSpatialBlocks (:319-344): depthwise (k,1,1) temporal conv -> GroupNorm(8) -> SiLU ->
  (1,3,3) conv -> GroupNorm(8) -> SiLU, with the time-embedding scale/shift applied between them (:378-382) and a
  residual branch that is a full (k,1,1) conv + GroupNorm(8) when the width changes (:361-366);
* each level has temporal attention over frames followed by full softmax SpatialAttention over the h*w positions
  of every frame (:390-459, :562-563);  RelativePositionBias uses max_distance = 300 (:526);
* the final ResnetBlock owns a time MLP but is called without the embedding (:591-594, :635).

Class names, constructor signatures and state-dict keys follow that file.  Kernels: vdmi_temporal_dwconv,
vdmi_groupnorm (activation, time scale/shift and residual fused into its second pass), vdmi_tapgemm for the
(1,3,3) and (k,1,1) convs, the fused temporal attention, and the cross-attention core for the spatial attention.
Channel widths must be 64 * 2^n (the GroupNorm kernel's row split).  The sampler around it
(`p_sample(guidance_fn=, injection_fn=)`) is `version1.GaussianDiffusion`, whose code the two files share.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib
from ._lib import EPI_PLAIN
from .engine import PackedTaps, Plan, TAPS_3x3, UnetEngine, _fill_pass, _ptr
from .modules import (Downsample, PreNorm, RelativePositionBias, Residual, RotaryEmbedding, SinusoidalPosEmb, Upsample,
                      _HipOnly, default, exists, is_odd)
from .version1 import GaussianDiffusion, temporal_consistency_guidance, temporal_consistency_injection  # noqa: F401


class TemporalSpatialBlock(_HipOnly):
    def __init__(self, dim, dim_out, kernel_size=3):
        super().__init__()
        self.temporal_conv = nn.Conv3d(dim, dim, kernel_size=(kernel_size, 1, 1), padding=(kernel_size // 2, 0, 0), groups=dim)
        self.temporal_norm = nn.GroupNorm(8, dim)
        self.spatial_conv = nn.Conv3d(dim, dim_out, kernel_size=(1, 3, 3), padding=(0, 1, 1))
        self.spatial_norm = nn.GroupNorm(8, dim_out)
        self.act = nn.SiLU()


class ResnetBlock(_HipOnly):
    def __init__(self, dim, dim_out, *, time_emb_dim=None, temporal_kernel=3):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, dim_out * 2)) if exists(time_emb_dim) else None
        self.ts_block1 = TemporalSpatialBlock(dim, dim_out, kernel_size=temporal_kernel)
        self.ts_block2 = TemporalSpatialBlock(dim_out, dim_out, kernel_size=temporal_kernel)
        if dim != dim_out:
            self.res_conv = nn.Sequential(
                nn.Conv3d(dim, dim_out, kernel_size=(temporal_kernel, 1, 1), padding=(temporal_kernel // 2, 0, 0)),
                nn.GroupNorm(8, dim_out))
        else:
            self.res_conv = nn.Identity()


class TemporalAttention(_HipOnly):
    def __init__(self, dim, heads=4, dim_head=32, rotary_emb=None):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        hidden_dim = dim_head * heads
        self.rotary_emb = rotary_emb
        self.to_qkv = nn.Linear(dim, hidden_dim * 3, bias=False)
        self.to_out = nn.Linear(hidden_dim, dim, bias=False)


class SpatialAttention(_HipOnly):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        hidden_dim = dim_head * heads
        self.to_qkv = nn.Linear(dim, hidden_dim * 3, bias=False)
        self.to_out = nn.Linear(hidden_dim, dim, bias=False)


class Unet3D(nn.Module):
    """video_diffusion_guidance.py:515-635."""

    def __init__(self, dim, out_dim=None, dim_mults=(1, 2, 4, 8), channels=4, attn_heads=8, attn_dim_head=32,
                 init_dim=None, init_kernel_size=7, temporal_kernel=3):
        super().__init__()
        self.channels, self.dim, self.dim_mults = channels, dim, tuple(dim_mults)
        self.attn_heads, self.attn_dim_head = attn_heads, attn_dim_head
        self.init_kernel_size, self.temporal_kernel = init_kernel_size, temporal_kernel
        self.has_cond, self.cond_dim, self.has_frame_cross_attn = False, None, False
        rotary_emb = RotaryEmbedding(min(32, attn_dim_head))
        self.time_rel_pos_bias = RelativePositionBias(heads=attn_heads, max_distance=300)
        init_dim = default(init_dim, dim)
        assert is_odd(init_kernel_size)
        pad = init_kernel_size // 2
        self.init_conv = nn.Conv3d(channels, init_dim, (1, init_kernel_size, init_kernel_size), padding=(0, pad, pad))
        tattn = lambda d: Residual(PreNorm(d, TemporalAttention(d, heads=attn_heads, dim_head=attn_dim_head,
                                                                rotary_emb=rotary_emb)))
        sattn = lambda d: Residual(PreNorm(d, SpatialAttention(d, heads=attn_heads, dim_head=attn_dim_head)))
        self.init_temporal_attn = tattn(init_dim)
        dims = [init_dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.dims = dims
        time_dim = dim * 8
        self.time_mlp = nn.Sequential(SinusoidalPosEmb(dim), nn.Linear(dim, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        rb = lambda a, b: ResnetBlock(a, b, time_emb_dim=time_dim, temporal_kernel=temporal_kernel)
        self.downs = nn.ModuleList([])
        n = len(in_out)
        for ind, (dim_in, dim_out) in enumerate(in_out):
            last = ind >= n - 1
            self.downs.append(nn.ModuleList([rb(dim_in, dim_out), rb(dim_out, dim_out), tattn(dim_out), sattn(dim_out),
                                             Downsample(dim_out) if not last else nn.Identity()]))
        mid = dims[-1]
        self.mid_block1 = rb(mid, mid)
        self.mid_temporal_attn = tattn(mid)
        self.mid_spatial_attn = sattn(mid)
        self.mid_block2 = rb(mid, mid)
        self.ups = nn.ModuleList([])
        for ind, (dim_in, dim_out) in enumerate(reversed(in_out)):
            last = ind >= n - 1
            self.ups.append(nn.ModuleList([rb(dim_out * 2, dim_in), rb(dim_in, dim_in), tattn(dim_in), sattn(dim_in),
                                           Upsample(dim_in) if not last else nn.Identity()]))
        out_dim = default(out_dim, channels)
        self.out_dim = out_dim
        self.final_conv = nn.Sequential(rb(dim * 2, dim), nn.Conv3d(dim, out_dim, 1))
        self._engine = None
        self.train_backend = None       # "torch-autograd" (the Trainer sets it): forward as differentiable torch ops

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_engine" else copy.deepcopy(v, memo)
        return new

    def engine(self):
        if self._engine is None:
            object.__setattr__(self, "_engine", GuidanceEngine(self))
        return self._engine

    def hip_training_supported(self) -> bool:
        from . import train_hip_variant
        return train_hip_variant.supports(self)

    def forward(self, x, time):
        if torch.is_grad_enabled() and (x.requires_grad or self.train_backend is not None):
            # a gradient with respect to x_t (the guidance sampler, video_diffusion_guidance.py:725-761) or to the
            # parameters (training): "hip" = every layer forward and backward on the HIP kernels (train_hip_variant.py),
            # None / "torch-autograd" = torch ops over the same parameters, as the reference differentiates
            if self.train_backend not in (None, "torch-autograd", "hip"):
                raise _lib.VdmiError(f"guidance_variant.Unet3D differentiates through train_backend 'hip' or "
                                     f"'torch-autograd', not {self.train_backend!r}")
            if self.train_backend == "hip" and not (x.is_cuda and self.hip_training_supported()):
                raise _lib.VdmiError("train_backend 'hip' needs CUDA (HIP) tensors and layer shapes the training kernels "
                                     "take (train_hip_variant.supports); there is no CPU path")
            from .autograd import guidance_variant_forward
            return guidance_variant_forward(self, x, time)
        return self.engine().forward(x, time)


class GuidancePlan(Plan):
    """Launch list of the variant; reuses the base plan's buffers, time-embedding tables and attention builders."""

    def _groupnorm(self, name, x, out, c, gn, act, gain=None, shift=None, bstride=0, res=None):
        lib = _lib.lib()
        rps = self.F * (x.shape[0] // (self.B * self.F))
        ws = torch.empty(max(16, lib.vdmi_groupnorm_workspace_bytes(self.B, rps, c)), device=self.device, dtype=torch.uint8)
        self.keep.append(ws)
        meta = {"kernel": "groupnorm", "flops": 0.0, "bytes": 2.0 * x.shape[0] * c * (4 if res is not None else 3)}
        self._add(lambda s: lib.vdmi_groupnorm(
            x.data_ptr(), out.data_ptr(), self.B, rps, c, 8, gn["g"].data_ptr(), gn["b"].data_ptr(), 1e-5, act,
            _ptr(gain), _ptr(shift), bstride, _ptr(res), res.shape[1] if res is not None else 0, ws.data_ptr(), s),
            meta, name)

    def _ts_block(self, name, tb, srcs, h, w, gain=None, shift=None, bstride=0, res=None):
        """TemporalSpatialBlock; the depthwise conv also materialises the concatenation of two sources."""
        lib = _lib.lib()
        hw, rows = h * w, self.B * self.F * h * w
        cin = sum(c for _, c in srcs)
        t1 = self._buf(rows, cin)
        col = 0
        for (src, c), (wd, bd) in zip(srcs, tb["dw"]):
            self._add(lambda s, src=src, c=c, wd=wd, bd=bd, col=col: lib.vdmi_temporal_dwconv(
                src.data_ptr(), src.shape[1], t1.data_ptr() + 2 * col, cin, self.B, self.F, hw, c, tb["k"], wd.data_ptr(),
                bd.data_ptr(), s), {"kernel": "temporal_dwconv", "flops": 0.0, "bytes": 4.0 * rows * c}, name + ".temporal_conv")
            col += c
        self._groupnorm(name + ".temporal_norm", t1, t1, cin, tb["tn"], 1)
        n = tb["w"].n
        s1 = self._buf(rows, n)
        a = self._base_args(self.B * self.F, h, w, s1, n, EPI_PLAIN)
        _fill_pass(a.main, t1, cin, None, 0, h, w, 1, TAPS_3x3, tb["w"].data)
        a.bias = _ptr(tb["b"])
        a.tile_bn = tb["w"].tile_bn
        self._tapgemm(a, name + ".spatial_conv")
        out = self._buf(rows, n)
        self._groupnorm(name + ".spatial_norm", s1, out, n, tb["sn"], 1, gain, shift, bstride, res)
        return out, n

    def _gres(self, name, rb, srcs, h, w, use_time=True):
        """ResnetBlock of the variant (video_diffusion_guidance.py:345-388)."""
        if self._tables_pending:    # first reader of the scale/shift tables: the side stream must be done
            self._tables_pending = False
            self._join_next = True
        rows = self.B * self.F * h * w
        n = rb["ts1"]["w"].n
        if rb["res"] is not None:
            r = rb["res"]
            (s0, c0), (s1, c1) = srcs[0], (srcs[1] if len(srcs) > 1 else (None, 0))
            res = self._buf(rows, n)
            a = self._base_args(self.B, self.F, h * w, res, n, EPI_PLAIN)
            k = r["k"]
            _fill_pass(a.main, s0, c0, s1, c1, self.F, h * w, 1, [(j - k // 2, 0) for j in range(k)], r["w"].data)
            a.bias = _ptr(r["b"])
            self._tapgemm(a, name + ".res_conv.0")
            self._groupnorm(name + ".res_conv.1", res, res, n, r["gn"], 0)
        else:
            if len(srcs) != 1:
                raise _lib.VdmiError("an identity residual needs a single input tensor")
            res = srcs[0][0]
        gain = shift = None
        bstride = 0
        if use_time and rb["ss_off"] is not None:
            gain, shift, bstride = self.gain[:, rb["ss_off"]:], self.shift[:, rb["ss_off"]:], self.gain.shape[1]
        h1, _ = self._ts_block(name + ".ts_block1", rb["ts1"], srcs, h, w, gain, shift, bstride)
        out, _ = self._ts_block(name + ".ts_block2", rb["ts2"], [(h1, n)], h, w, res=res)
        return out

    def _build(self):
        eng, u = self.eng, self.eng.unet
        lib = _lib.lib()
        P = eng.packed
        B, F, H, W = self.B, self.F, self.H, self.W
        dims = u.dims
        ic = P["init_conv"]
        x = self._buf(B * F * H * W, dims[0])
        self._add(lambda s, x=x: lib.vdmi_init_conv(
            self.x_in.data_ptr(), ic["w"].data_ptr(), ic["b"].data_ptr(), x.data_ptr(), B, u.channels, F, H, W, dims[0],
            u.init_kernel_size, s), None, "init_conv")
        x = self._attention("init_temporal_attn", P["init_temporal_attn"], x, dims[0], H, W, True)
        r = x
        h, w, c = H, W, dims[0]
        skips = []
        nres = len(dims) - 1
        for i in range(nres):
            L = P["downs"][i]
            x = self._gres(f"downs.{i}.0", L["rb0"], [(x, c)], h, w)
            c = dims[i + 1]
            x = self._gres(f"downs.{i}.1", L["rb1"], [(x, c)], h, w)
            x = self._attention(f"downs.{i}.2", L["tattn"], x, c, h, w, True)
            x = self._sta(f"downs.{i}.3", L["sattn"], x, c, h, w, per_frame=True)
            self.taps[f"downs.{i}"] = x
            skips.append((x, c, h, w))
            if L["down"] is not None:
                x = self._down(f"downs.{i}.4", L["down"], x, c, h, w)
                h, w = (h + 2 - 4) // 2 + 1, (w + 2 - 4) // 2 + 1
        x = self._gres("mid_block1", P["mid_block1"], [(x, c)], h, w)
        x = self._attention("mid_temporal_attn", P["mid_temporal_attn"], x, c, h, w, True)
        x = self._sta("mid_spatial_attn", P["mid_spatial_attn"], x, c, h, w, per_frame=True)
        x = self._gres("mid_block2", P["mid_block2"], [(x, c)], h, w)
        for i in range(nres):
            L = P["ups"][i]
            sk, sc, sh, sw = skips.pop()
            if (sh, sw) != (h, w):
                raise _lib.VdmiError("H and W must be divisible by 2**(levels-1)")
            x = self._gres(f"ups.{i}.0", L["rb0"], [(x, c), (sk, sc)], h, w)
            c = L["rb0"]["ts1"]["w"].n
            x = self._gres(f"ups.{i}.1", L["rb1"], [(x, c)], h, w)
            x = self._attention(f"ups.{i}.2", L["tattn"], x, c, h, w, True)
            x = self._sta(f"ups.{i}.3", L["sattn"], x, c, h, w, per_frame=True)
            if L["up"] is not None:
                x = self._up(f"ups.{i}.4", L["up"], x, c, h, w)
                h, w = 2 * h, 2 * w
        x = self._gres("final_conv.0", P["final_rb"], [(x, c), (r, dims[0])], h, w, use_time=False)
        fc = P["final_conv"]
        self._add(lambda s, x=x: lib.vdmi_final_conv(
            x.data_ptr(), fc["w"].data_ptr(), fc["b"].data_ptr(), self.eps_out.data_ptr(), B, u.dim, F, H, W, u.out_dim, s),
            None, "final_conv.1")


class GuidanceEngine(UnetEngine):
    plan_class = GuidancePlan

    def _pack_ts(self, tb, device, cin_split):
        f32 = lambda t: t.detach().float().to(device).contiguous()
        k = tb.temporal_conv.kernel_size[0]
        wd = tb.temporal_conv.weight.detach().float().reshape(-1, k)
        bd = tb.temporal_conv.bias.detach().float()
        dw, off = [], 0
        for c in cin_split:
            if c:
                dw.append((wd[off:off + c].to(device).contiguous(), bd[off:off + c].to(device).contiguous()))
                off += c
        n, cin = tb.spatial_conv.weight.shape[0], tb.spatial_conv.weight.shape[1]
        w = PackedTaps(self._conv_w(tb.spatial_conv.weight), cin, 0, EPI_PLAIN, device, tile_bn=64 if n >= 256 else 0)
        gn = lambda g: {"g": f32(g.weight), "b": f32(g.bias)}
        return {"k": k, "dw": dw, "tn": gn(tb.temporal_norm), "w": w, "b": f32(tb.spatial_conv.bias), "sn": gn(tb.spatial_norm)}

    def _pack_gres(self, rb, device, cin_split, stacked):
        f32 = lambda t: t.detach().float().to(device).contiguous()
        n = rb.ts_block1.spatial_conv.weight.shape[0]
        for c in (sum(cin_split), n):
            if c < 64 or c & (c - 1):
                raise _lib.VdmiError(f"channel widths of this variant must be 64 * 2^n for the GroupNorm kernel, got {c}")
        d = {"ts1": self._pack_ts(rb.ts_block1, device, cin_split), "ts2": self._pack_ts(rb.ts_block2, device, (n, 0))}
        if isinstance(rb.res_conv, nn.Sequential):
            conv, gn = rb.res_conv[0], rb.res_conv[1]
            k = conv.kernel_size[0]
            d["res"] = {"k": k, "w": PackedTaps(conv.weight.detach().float().cpu().reshape(n, sum(cin_split), k), cin_split[0],
                                                cin_split[1], EPI_PLAIN, device),
                        "b": f32(conv.bias), "gn": {"g": f32(gn.weight), "b": f32(gn.bias)}}
        else:
            d["res"] = None
        if rb.mlp is not None:
            d["ss_off"] = stacked["half"]
            stacked["w"].append(rb.mlp[1].weight.detach().float().cpu())
            stacked["b"].append(rb.mlp[1].bias.detach().float().cpu())
            stacked["g"].append(torch.ones(n))          # gain table = scale + 1 (no norm gamma to fold in here)
            stacked["off"].append(stacked["half"])
            stacked["ch"].append(n)
            stacked["half"] += n
        else:
            d["ss_off"] = None
        return d

    def _pack(self, device):
        u = self.unet
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
        P["null_cond"], P["emb_dim"], P["xattn"] = None, u.dim * 8, None
        P["downs"] = []
        for i, (b1, b2, tat, sat, down) in enumerate(u.downs):
            P["downs"].append({"rb0": self._pack_gres(b1, device, (dims[i], 0), stacked),
                               "rb1": self._pack_gres(b2, device, (dims[i + 1], 0), stacked),
                               "tattn": self._pack_attn(tat, device, u.attn_dim_head),
                               "sattn": self._pack_sta(sat, device),
                               "down": self._pack_down(down, device) if isinstance(down, nn.Conv3d) else None})
        mid = dims[-1]
        P["mid_block1"] = self._pack_gres(u.mid_block1, device, (mid, 0), stacked)
        P["mid_temporal_attn"] = self._pack_attn(u.mid_temporal_attn, device, u.attn_dim_head)
        P["mid_spatial_attn"] = self._pack_sta(u.mid_spatial_attn, device)
        P["mid_block2"] = self._pack_gres(u.mid_block2, device, (mid, 0), stacked)
        P["ups"] = []
        in_out = list(zip(dims[:-1], dims[1:]))
        for (dim_in, dim_out), (b1, b2, tat, sat, up) in zip(reversed(in_out), u.ups):
            P["ups"].append({"rb0": self._pack_gres(b1, device, (dim_out, dim_out), stacked),
                             "rb1": self._pack_gres(b2, device, (dim_in, 0), stacked),
                             "tattn": self._pack_attn(tat, device, u.attn_dim_head),
                             "sattn": self._pack_sta(sat, device),
                             "up": self._pack_up(up, device) if isinstance(up, nn.ConvTranspose3d) else None})
        P["final_rb"] = self._pack_gres(u.final_conv[0], device, (dims[0], dims[0]), stacked)
        fcw = u.final_conv[1].weight
        P["final_conv"] = {"w": f32(fcw.reshape(fcw.shape[0], -1)), "b": f32(u.final_conv[1].bias)}
        P["w_all"] = torch.cat(stacked["w"], 0).to(device).contiguous()
        P["b_all"] = torch.cat(stacked["b"], 0).to(device).contiguous()
        P["gamma_all"] = torch.cat(stacked["g"], 0).to(device).contiguous()
        P["blk_off"] = torch.tensor(stacked["off"], dtype=torch.int32, device=device)
        P["blk_ch"] = torch.tensor(stacked["ch"], dtype=torch.int32, device=device)
        P["nblocks"] = len(stacked["off"])
        P["half_total"] = stacked["half"]
        return P
