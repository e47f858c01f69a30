"""The layers of the reference's `video_diffusion_guidance.py` denoiser forward AND backward on the HIP kernels.

The reference's guidance sampler differentiates a reconstruction loss through the denoiser with PyTorch autograd
(video_diffusion_guidance.py:725-761, 860-900).  Here the layers of that denoiser are torch.autograd.Functions over
channels-last bf16 rows, as train_hip.py does for the packaged model:

  TemporalSpatialBlock (:319-344)   vdmi_temporal_dwconv -> vdmi_groupnorm(SiLU) -> vdmi_tapgemm (1,3,3) -> vdmi_groupnorm(SiLU,
                                    time scale / shift, residual);  backward: vdmi_groupnorm_bwd, vdmi_tapgemm over the
                                    transposed tap-reversed weights, vdmi_temporal_dwconv with reversed taps,
                                    vdmi_conv_wgrad / vdmi_temporal_dwconv_wgrad for the parameters
  res_conv (:361-366)               the (k,1,1) convolution over frames as a k-tap vdmi_tapgemm on the (frame, position)
                                    view of the rows, then vdmi_groupnorm without activation
  TemporalAttention / SpatialAttention (:390-459)   train_hip.TemporalAttnFn (spatial: its per-frame form, up to 256
                                    positions; beyond that torch's fused attention over the same bf16 rows)

There is no fallback for the convolution / normalisation layers: a shape the kernels do not take raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check
from .train_hip import (TemporalAttnFn, _act_dtype, _plain, _stream, attention_core_fits, colsum_rows, conv3x3_wgrad, conv_rows,
                        frame_conv_rows, frame_conv_wgrad, from_rows, param_pack, rows_wgrad, to_rows, torch_ops)


def _f(t):
    return t.detach().float().contiguous()


def groupnorm_rows(x, groups, gamma, beta, eps, nsamples, act, post_gain=None, post_shift=None, residual=None):
    rows, c = x.shape
    lib = _lib.lib()
    rps = rows // nsamples
    out = torch.empty_like(x)
    ws = torch.empty(lib.vdmi_groupnorm_workspace_bytes(nsamples, rps, c), device=x.device, dtype=torch.uint8)
    ptr = lambda t: None if t is None else t.data_ptr()
    check(lib.vdmi_groupnorm(x.data_ptr(), out.data_ptr(), nsamples, rps, c, groups, gamma.data_ptr(), beta.data_ptr(), eps,
                             act, ptr(post_gain), ptr(post_shift), c if post_gain is not None else 0, ptr(residual),
                             0 if residual is None else residual.shape[1], ws.data_ptr(), _stream()), "vdmi_groupnorm")
    return out


def groupnorm_bwd_rows(x, dout, groups, gamma, beta, eps, nsamples, act, post_gain=None):
    """-> (dx rows, chan float [nsamples][c][4] = per sample {sum dy, sum dy * xhat, sum dout * act(y), sum dout})."""
    rows, c = x.shape
    lib = _lib.lib()
    rps = rows // nsamples
    dx = torch.empty_like(x)
    chan = torch.empty(nsamples, c, 4, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.vdmi_groupnorm_bwd_workspace_bytes(nsamples, rps, c, groups), device=x.device, dtype=torch.uint8)
    check(lib.vdmi_groupnorm_bwd(x.data_ptr(), dout.data_ptr(), dout.stride(0), dx.data_ptr(), nsamples, rps, c, groups,
                                 gamma.data_ptr(), beta.data_ptr(), eps, act,
                                 None if post_gain is None else post_gain.data_ptr(), c if post_gain is not None else 0,
                                 chan.data_ptr(), ws.data_ptr(), _stream()), "vdmi_groupnorm_bwd")
    return dx, chan


def dwconv_rows(x, w2, bias, b, f, hw):
    """Depthwise temporal convolution of rows; w2 float [c][k]."""
    c, k = w2.shape
    out = torch.empty_like(x)
    check(_lib.lib().vdmi_temporal_dwconv(x.data_ptr(), c, out.data_ptr(), c, b, f, hw, c, k, w2.data_ptr(),
                                          None if bias is None else bias.data_ptr(), _stream()), "vdmi_temporal_dwconv")
    return out


def dwconv_wgrad_rows(x, du, b, f, hw, k):
    """-> float [c][k + 1]: the gradient of the taps and (last column) of the bias."""
    c = x.shape[1]
    lib = _lib.lib()
    part = torch.empty(lib.vdmi_temporal_dwconv_wgrad_elems(b, f, hw, c, k), device=x.device, dtype=torch.float32)
    ns = C.c_int()
    check(lib.vdmi_temporal_dwconv_wgrad(x.data_ptr(), du.data_ptr(), part.data_ptr(), C.byref(ns), b, f, hw, c, k, _stream()),
          "vdmi_temporal_dwconv_wgrad")
    return part.view(ns.value, c, k + 1).sum(0)


class TemporalSpatialFn(torch.autograd.Function):
    """TemporalSpatialBlock.forward (video_diffusion_guidance.py:334-344), then optionally the time scale / shift of
    ResnetBlock.forward (:378-382) and the residual add (:386): x (B, C, F, H, W) -> (B, Cout, F, H, W)."""

    @staticmethod
    @_plain
    def forward(ctx, x, wt, bt, gt, bet, ws, bs, gs, bes, scale, shift, residual, eps_t, eps_s, groups):
        if not x.is_cuda:
            raise _lib.VdmiError("the HIP training backend needs CUDA (HIP) tensors; there is no CPU path")
        b, c, f, h, w = x.shape
        cout, k, hw = ws.shape[0], wt.shape[2], h * w
        xr = to_rows(x)
        wt2 = _f(wt).reshape(c, k)
        gtv, betv, gsv, besv = _f(gt), _f(bet), _f(gs), _f(bes)
        u = dwconv_rows(xr, wt2, _f(bt), b, f, hw)
        hmid = groupnorm_rows(u, groups, gtv, betv, eps_t, b, 1)
        v = conv_rows(hmid, param_pack(ws, False), _f(bs), b * f, f, h, w)
        pg = None if scale is None else (_f(scale) + 1).contiguous()
        ps = None if shift is None else _f(shift)
        rr = None if residual is None else to_rows(residual)
        out = groupnorm_rows(v, groups, gsv, besv, eps_s, b, 1, pg, ps, rr)
        ctx.save_for_backward(xr, u, hmid, v, wt2, gtv, betv, gsv, besv, pg, ws)
        ctx.meta = (b, c, cout, f, h, w, k, eps_t, eps_s, groups, residual is not None)
        ctx.act = _act_dtype(x)
        return from_rows(out, b, f, h, w, ctx.act)

    @staticmethod
    @_plain
    def backward(ctx, dout):
        xr, u, hmid, v, wt2, gtv, betv, gsv, besv, pg, ws = ctx.saved_tensors
        b, c, cout, f, h, w, k, eps_t, eps_s, groups, has_res = ctx.meta
        need = ctx.needs_input_grad
        hw, frames = h * w, b * f
        do = to_rows(dout)
        # spatial half: GroupNorm / SiLU / scale-shift, then the (1,3,3) convolution
        dv, chan_s = groupnorm_bwd_rows(v, do, groups, gsv, besv, eps_s, b, 1, pg)
        dws = dbs = dgs = dbes = dscale = dshift = None
        if need[5]:
            dws = conv3x3_wgrad(hmid, dv, frames, h, w)
        if need[6] or need[7] or need[8]:
            tot = chan_s.sum(0)
            dbes, dgs = tot[:, 0].contiguous(), tot[:, 1].contiguous()
            dbs = colsum_rows(dv) if need[6] else None
        if pg is not None:
            dscale, dshift = chan_s[:, :, 2].contiguous(), chan_s[:, :, 3].contiguous()
        dh = conv_rows(dv, param_pack(ws, True), None, frames, f, h, w)
        # temporal half: GroupNorm / SiLU, then the depthwise convolution over frames
        du, chan_t = groupnorm_bwd_rows(u, dh, groups, gtv, betv, eps_t, b, 1)
        dwt = dbt = dgt = dbet = None
        if need[1] or need[2]:
            g = dwconv_wgrad_rows(xr, du, b, f, hw, k)
            dwt, dbt = g[:, :k].reshape(c, 1, k, 1, 1).contiguous(), g[:, k].contiguous()
        if need[3] or need[4]:
            tot = chan_t.sum(0)
            dbet, dgt = tot[:, 0].contiguous(), tot[:, 1].contiguous()
        dx = dwconv_rows(du, wt2.flip(1).contiguous(), None, b, f, hw) if need[0] else None
        return (None if dx is None else from_rows(dx, b, f, h, w, ctx.act), dwt, dbt, dgt, dbet, dws, dbs, dgs, dbes, dscale,
                dshift, dout if has_res else None, None, None, None)


class FrameConvNormFn(torch.autograd.Function):
    """The residual branch of ResnetBlock when the width changes (video_diffusion_guidance.py:361-366): a full (k,1,1)
    convolution over frames, then GroupNorm without activation."""

    @staticmethod
    @_plain
    def forward(ctx, x, wc, bc, gamma, beta, eps, groups):
        b, c, f, h, w = x.shape
        hw = h * w
        xr = to_rows(x)
        gv, bev = _f(gamma), _f(beta)
        y = frame_conv_rows(xr, param_pack(wc, False), _f(bc), b, f, hw)            # (cout, cin, k, 1, 1): k taps
        out = groupnorm_rows(y, groups, gv, bev, eps, b, 0)
        ctx.save_for_backward(xr, y, gv, bev, wc)
        ctx.meta = (b, c, wc.shape[0], f, h, w, wc.shape[2], eps, groups)
        ctx.act = _act_dtype(x)
        return from_rows(out, b, f, h, w, ctx.act)

    @staticmethod
    @_plain
    def backward(ctx, dout):
        xr, y, gv, bev, wc = ctx.saved_tensors
        b, c, cout, f, h, w, k, eps, groups = ctx.meta
        need = ctx.needs_input_grad
        hw = h * w
        dy, chan = groupnorm_bwd_rows(y, to_rows(dout), groups, gv, bev, eps, b, 0)
        dwc = dbc = dg = dbe = None
        if need[1]:
            dwc = frame_conv_wgrad(xr, dy, b, f, hw, k)
        if need[2] or need[3] or need[4]:
            tot = chan.sum(0)
            dbe, dg = tot[:, 0].contiguous(), tot[:, 1].contiguous()
            dbc = colsum_rows(dy) if need[2] else None
        dx = None
        if need[0]:
            dx = from_rows(frame_conv_rows(dy, param_pack(wc, True), None, b, f, hw), b, f, h, w, ctx.act)
        return dx, dwc, dbc, dg, dbe, None, None


def _ts(tb, x, scale=None, shift=None, residual=None):
    tn, sn = tb.temporal_norm, tb.spatial_norm
    return TemporalSpatialFn.apply(x, tb.temporal_conv.weight, tb.temporal_conv.bias, tn.weight, tn.bias,
                                   tb.spatial_conv.weight, tb.spatial_conv.bias, sn.weight, sn.bias, scale, shift, residual,
                                   float(tn.eps), float(sn.eps), int(sn.num_groups))


def resnet_block(blk, x, temb=None):
    """ResnetBlock.forward, video_diffusion_guidance.py:370-388."""
    if isinstance(blk.res_conv, torch.nn.Identity):
        res = x
    else:
        conv, norm = blk.res_conv[0], blk.res_conv[1]
        res = FrameConvNormFn.apply(x, conv.weight, conv.bias, norm.weight, norm.bias, float(norm.eps), int(norm.num_groups))
    scale = shift = None
    if blk.mlp is not None and temb is not None:
        scale, shift = blk.mlp(temb).chunk(2, dim=1)
    return _ts(blk.ts_block2, _ts(blk.ts_block1, x, scale, shift), residual=res)


def temporal_attention(res, x, pos_bias, rot_tables, ops):
    """Residual(PreNorm(TemporalAttention)), :390-428; `ops`: the torch-op form for frame counts beyond the core."""
    att = res.fn.fn
    if att.dim_head != 32 or not attention_core_fits(x.shape[2], att.heads, pos_bias is not None):
        return torch_ops(ops, x)
    cos, sin = rot_tables if rot_tables is not None else (None, None)
    return TemporalAttnFn.apply(x, res.fn.norm.gamma, att.to_qkv.weight, att.to_out.weight, pos_bias, cos, sin, att.heads,
                                float(att.scale), float(res.fn.norm.eps))


def spatial_attention(res, x, ops):
    """Residual(PreNorm(SpatialAttention)), :431-459."""
    att = res.fn.fn
    n = x.shape[3] * x.shape[4]
    if att.dim_head != 32 or n > 256 or not attention_core_fits(n, att.heads, False):
        return torch_ops(ops, x)
    return TemporalAttnFn.apply(x, res.fn.norm.gamma, att.to_qkv.weight, att.to_out.weight, None, None, None, att.heads,
                                float(att.scale), float(res.fn.norm.eps), True)


def supports(unet) -> bool:
    """True when every convolution / GroupNorm layer of a guidance_variant.Unet3D has a shape the kernels take: widths
    64 * 2^n up to 2048, 8 groups, temporal kernels of 1, 3, 5 or 7 taps, heads of 32 features."""
    if _lib.PRECISION != "bf16":
        return False
    ok = lambda c: 64 <= c <= 2048 and (c & (c - 1)) == 0
    for mod in unet.modules():
        name = type(mod).__name__
        if name == "TemporalSpatialBlock":
            if not (ok(mod.spatial_conv.in_channels) and ok(mod.spatial_conv.out_channels)
                    and mod.temporal_conv.kernel_size[0] in (1, 3, 5, 7)):
                return False
        elif isinstance(mod, torch.nn.GroupNorm):
            if mod.num_channels % (8 * mod.num_groups) or not mod.affine:
                return False
        elif name in ("TemporalAttention", "SpatialAttention"):
            if mod.dim_head != 32:
                return False
            rot = getattr(mod, "rotary_emb", None)
            if rot is not None and getattr(rot, "convention", "half") != "half":
                return False
        elif isinstance(mod, (torch.nn.ConvTranspose3d, torch.nn.Conv3d)) and mod.kernel_size == (1, 4, 4):
            if mod.in_channels % 64:
                return False
    return True
