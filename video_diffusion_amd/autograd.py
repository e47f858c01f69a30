"""Differentiable execution of Unet3D for training.

The reference trains through PyTorch autograd (its Trainer, video_diffusion_pytorch/video_diffusion_pytorch.py:1414-1453,
calls loss.backward() on torch ops); this module runs the same arithmetic as torch ops over the module's own
parameters so that `loss.backward()`, the optimizer and DistributedDataParallel work unchanged.  It is never used for
sampling: `Unet3D.forward` only comes here when gradients are required AND the owner opted in with
`unet.train_backend` (the Trainer does); otherwise a forward that needs gradients raises.  Backends:
"torch-autograd" - every layer a torch op, fp32 like the reference; "hip-resnet" - the ResnetBlocks forward and backward on
the HIP kernels (train_hip.py); "hip" - the temporal, linear and mid-block spatial attention layers and the resampling
convolutions too, with bf16 activations between the layers (what stays a torch op: embeddings, cross-attention,
focus-masked temporal attention, the forward of init_conv / final_conv).

Line references: vdp.py = video_diffusion_pytorch/video_diffusion_pytorch.py of the reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib


def _chan_layernorm(x, norm):
    # vdp.py:380-389
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) / (var + norm.eps).sqrt() * norm.gamma


def _rmsnorm(x, norm):
    # vdp.py:391-398
    return F.normalize(x, dim=1) * norm.scale * norm.gamma


def _block(blk, x, scale_shift=None):
    # vdp.py:410-427
    x = _rmsnorm(blk.proj(x), blk.norm)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


_HIP_RESNET = False     # set by unet_forward for the duration of one forward (train_backend == "hip-resnet")


def _resnet(rb, x, temb):
    # vdp.py:429-452; the (3,1,1) temporal conv of version1diffusion.py:308,322 when the block has one
    if _HIP_RESNET and x.is_cuda:
        from .train_hip import resnet_block     # forward and backward on the HIP kernels
        return resnet_block(rb, x, temb)
    ss = None
    if rb.mlp is not None:
        e = rb.mlp[1](F.silu(temb))
        ss = e[:, :, None, None, None].chunk(2, dim=1)
    h = _block(rb.block2, _block(rb.block1, x, ss))
    if getattr(rb, "temporal_conv", None) is not None:
        h = rb.temporal_conv(h)
    return h + rb.res_conv(x)


def _rotate(t, rot):
    """t [..., n, d] with positions 0..n-1 on the second-to-last axis."""
    n, d = t.shape[-2], t.shape[-1]
    ang = torch.arange(n, device=t.device, dtype=torch.float32)[:, None] * rot.inv_freq.to(t.device)[None, :]
    if getattr(rot, "convention", "half") == "interleaved":
        ang = ang.repeat_interleave(2, dim=-1)
        x1, x2 = t[..., 0::2], t[..., 1::2]
        return t * ang.cos() + torch.stack((-x2, x1), dim=-1).reshape(t.shape) * ang.sin()
    ang = torch.cat((ang, ang), dim=-1)              # video_diffusion_guidance.py:182-198
    t1, t2 = t.chunk(2, dim=-1)
    return t * ang.cos() + torch.cat((-t2, t1), dim=-1) * ang.sin()


def _attention(att, x, pos_bias=None, focus=None):
    """vdp.py:546-612 on x [b, g, n, c] (g groups of n tokens)."""
    b, g, n, _ = x.shape
    heads = att.heads
    q, k, v = att.to_qkv(x).chunk(3, dim=-1)
    q, k, v = (t.reshape(b, g, n, heads, -1).transpose(2, 3) for t in (q, k, v))     # [b, g, heads, n, d]
    q = q * att.scale
    if getattr(att, "rotary_emb", None) is not None:
        q, k = _rotate(q, att.rotary_emb), _rotate(k, att.rotary_emb)
    if pos_bias is None and focus is None and x.is_cuda and n >= 1024:
        # long token axes (SpatioTemporalAttention over all f*h*w positions, version1diffusion.py:325-345): torch's fused
        # attention op, which never materialises the n x n matrices (q is already scaled)
        q4, k4, v4 = (t.reshape(b * g, heads, n, -1) for t in (q, k, v))       # the fused kernels take 4-D operands
        if _HIP_ATTN:       # the "hip" backend's arithmetic: bf16 q, k, v rows, float accumulation (as its HIP attention cores)
            q4, k4, v4 = q4.to(torch.bfloat16), k4.to(torch.bfloat16), v4.to(torch.bfloat16)
        out = F.scaled_dot_product_attention(q4, k4, v4, scale=1.0).to(q.dtype).reshape(b, g, heads, n, -1)
        return att.to_out(out.transpose(2, 3).reshape(b, g, n, -1))
    sim = q @ k.transpose(-1, -2)
    if pos_bias is not None:
        sim = sim + pos_bias
    if focus is not None:
        eye = torch.eye(n, device=x.device, dtype=torch.bool)
        keep = eye.view(1, 1, 1, n, n) | ~focus.view(-1, 1, 1, 1, 1)
        sim = sim.masked_fill(~keep, -torch.finfo(sim.dtype).max)
    attn = (sim - sim.amax(dim=-1, keepdim=True).detach()).softmax(dim=-1)
    out = (attn @ v).transpose(2, 3).reshape(b, g, n, -1)
    return att.to_out(out)


_HIP_ATTN = False       # train_backend == "hip": the temporal attention layers too


def _torch_ops(fn, x):
    # a layer that runs as torch ops; with the "hip" backend x may belong to the bf16 activation stream of the HIP layers
    if x.dtype != torch.bfloat16:
        return fn(x)
    from .train_hip import torch_ops
    return torch_ops(fn, x)


def _rot_tables(rot, nf, device):
    if rot is None:
        return None
    if getattr(rot, "convention", "half") != "half":
        raise NotImplementedError("the HIP temporal attention backward takes the half-split rotary convention")
    # one table per (frame count, device) and state of the frequencies: every temporal layer of a forward shares the module
    key = (nf, str(device), rot.inv_freq.data_ptr(), rot.inv_freq._version, torch.is_inference_mode_enabled())
    hit = getattr(rot, "_vdmi_tables", None)
    if hit is None or hit[0] != key:
        ang = torch.arange(nf, device=device, dtype=torch.float32)[:, None] * rot.inv_freq.to(device).float()[None, :]
        hit = (key, (ang.cos().contiguous(), ang.sin().contiguous()))
        object.__setattr__(rot, "_vdmi_tables", hit)
    return hit[1]


def _core_fits(n_tok, heads, with_bias):
    """The HIP attention cores keep a site's tiles (and the backward's bias-gradient table) in LDS: long token axes
    (from about 63 frames with 8 heads and a relative position bias) stay on torch ops."""
    from .train_hip import attention_core_fits
    return attention_core_fits(int(n_tok), int(heads), bool(with_bias))


def _temporal(res, x, pos_bias, focus=None):
    # Residual(PreNorm(EinopsToAndFrom('b c f h w', 'b (h w) f c', Attention))), vdp.py:636
    att = res.fn.fn.fn
    if _HIP_ATTN and x.is_cuda and _core_fits(x.shape[2], att.heads, pos_bias is not None):
        from .train_hip import temporal_attention     # forward and backward on the HIP kernels
        return temporal_attention(res, x, pos_bias, _rot_tables(getattr(att, "rotary_emb", None), x.shape[2], x.device),
                                  focus)
    return _torch_ops(lambda x: _temporal_ops(res, x, pos_bias, focus), x)


def _temporal_ops(res, x, pos_bias, focus):
    b, c, f, h, w = x.shape
    y = _chan_layernorm(x, res.fn.norm).permute(0, 3, 4, 2, 1).reshape(b, h * w, f, c)
    o = _attention(res.fn.fn.fn, y, pos_bias, focus)
    return o.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2) + x


def _spatial(res, x):
    # mid_spatial_attn: 'b c f h w -> b f (h w) c', vdp.py:708-710
    if _HIP_ATTN and x.is_cuda and res.fn.fn.fn.dim_head == 32:
        from .train_hip import spatial_attention      # forward and backward on the HIP kernels
        return spatial_attention(res, x)
    return _torch_ops(lambda x: _spatial_ops(res, x), x)


def _spatial_ops(res, x):
    b, c, f, h, w = x.shape
    y = _chan_layernorm(x, res.fn.norm).permute(0, 2, 3, 4, 1).reshape(b, f, h * w, c)
    o = _attention(res.fn.fn.fn, y)
    return o.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3) + x


def _resample(mod, x):
    # Downsample / Upsample, vdp.py:374-378 (nn.Identity on the last level)
    if _HIP_ATTN and x.is_cuda and not isinstance(mod, torch.nn.Identity):
        from .train_hip import resample
        return resample(mod, x)
    return x if isinstance(mod, torch.nn.Identity) else _torch_ops(mod, x)


def _linear_attention(res, x):
    sl = res.fn.fn
    b, c, f, h, w = x.shape
    if _HIP_ATTN and x.is_cuda and isinstance(sl.to_qkv, torch.nn.Conv2d):
        from .train_hip import spatial_linear_attention     # forward and backward on the HIP kernels
        return spatial_linear_attention(res, x)
    if (_HIP_ATTN and x.is_cuda and isinstance(sl.to_qkv, torch.nn.Linear) and getattr(sl, "dim_head", 32) == 32
            and sl.to_qkv.bias is None and sl.to_out.bias is None):
        from .train_hip import spatio_temporal_attention    # SpatioTemporalAttention: the cross-attention kernels, context = x
        return spatio_temporal_attention(res, x)
    return _torch_ops(lambda x: _linear_attention_ops(res, x), x)


def _linear_attention_ops(res, x):
    sl = res.fn.fn
    b, c, f, h, w = x.shape
    y = _chan_layernorm(x, res.fn.norm)
    if isinstance(sl.to_qkv, torch.nn.Linear):
        # SpatioTemporalAttention, version1diffusion.py:325-345
        tok = y.permute(0, 2, 3, 4, 1).reshape(b, 1, f * h * w, c)
        return _attention(sl, tok).reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3) + x
    # SpatialLinearAttention, vdp.py:454-479
    yf = y.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    q, k, v = sl.to_qkv(yf).reshape(b * f, 3, sl.heads, -1, h * w).unbind(dim=1)
    q = q.softmax(dim=-2) * sl.scale
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b * f, -1, h, w)
    out = sl.to_out(out)
    return out.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4) + x


def _cross_attention(res, x, ctx):
    if _HIP_ATTN and x.is_cuda:
        from .train_hip import cross_attention, cross_attention_fits      # forward and backward on the HIP kernels
        if cross_attention_fits(res.fn.fn, ctx):
            return cross_attention(res, x, ctx)
    return _torch_ops(lambda x: _cross_attention_ops(res, x, ctx), x)


def _cross_attention_ops(res, x, ctx):
    # Residual(PreNorm(CrossAttention)), vdp.py:480-528
    ca = res.fn.fn
    b, c, f, h, w = x.shape
    tok = _chan_layernorm(x, res.fn.norm).permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c)
    q, k, v = ca.to_q(tok), ca.to_k(ctx), ca.to_v(ctx)
    q, k, v = (t.reshape(b, t.shape[1], ca.heads, -1).transpose(1, 2) for t in (q, k, v))
    attn = ((q @ k.transpose(-1, -2)) * ca.scale).softmax(dim=-1)
    out = ca.to_out((attn @ v).transpose(1, 2).reshape(b, f * h * w, -1))
    return out.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3) + x


def _pos_bias(unet, nf, device):
    # vdp.py:274-312: integer buckets from the library's host routine, gathered from the LIVE embedding weight
    rpb = unet.time_rel_pos_bias
    key = (nf, str(device), rpb.num_buckets, rpb.max_distance, torch.is_inference_mode_enabled())
    hit = getattr(rpb, "_vdmi_buckets", None)
    if hit is None or hit[0] != key:          # the bucket indices depend on the frame count alone: one host pass and copy
        buckets = np.empty((nf, nf), dtype=np.int32)
        _lib.check(_lib.lib().vdmi_relpos_buckets(nf, rpb.num_buckets, rpb.max_distance, buckets.ctypes.data),
                   "vdmi_relpos_buckets")
        hit = (key, torch.from_numpy(buckets.astype(np.int64)).to(device))
        object.__setattr__(rpb, "_vdmi_buckets", hit)
    return rpb.relative_attention_bias.weight[hit[1]].permute(2, 0, 1)


def _mask(shape, prob, device):
    # prob_mask_like, vdp.py:68-74
    if prob == 1:
        return torch.ones(shape, device=device, dtype=torch.bool)
    if prob == 0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    return torch.zeros(shape, device=device).float().uniform_(0, 1) < prob


def unet_forward(unet, x, time, cond=None, null_cond_prob=0., frame_cond_features=None, null_frame_cond_prob=0.,
                 focus_present_mask=None, prob_focus_present=0.):
    """Unet3D.forward (vdp.py:756-831) as differentiable torch ops over `unet`'s parameters; with
    `unet.train_backend == "hip-resnet"` the ResnetBlocks run forward and backward on the HIP kernels (train_hip.py)."""
    global _HIP_RESNET, _HIP_ATTN
    backend = getattr(unet, "train_backend", None)
    prev = (_HIP_RESNET, _HIP_ATTN)
    _HIP_RESNET, _HIP_ATTN = backend in ("hip-resnet", "hip"), backend == "hip"
    try:
        return _unet_forward(unet, x, time, cond, null_cond_prob, frame_cond_features, null_frame_cond_prob,
                             focus_present_mask, prob_focus_present)
    finally:
        _HIP_RESNET, _HIP_ATTN = prev


def _unet_forward(unet, x, time, cond, null_cond_prob, frame_cond_features, null_frame_cond_prob, focus_present_mask,
                  prob_focus_present):
    batch, device = x.shape[0], x.device
    focus = focus_present_mask if focus_present_mask is not None else _mask((batch,), prob_focus_present, device)
    focus = focus.to(device=device, dtype=torch.bool)
    if not bool(focus.any()):
        focus = None                                   # vdp.py:590: the mask is skipped when nothing is flagged
    bias = _pos_bias(unet, x.shape[2], device)

    if _HIP_RESNET and x.is_cuda:
        from .train_hip import init_conv        # forward and weight gradient on the HIP kernels
        # with the attention layers on the library too, the activations stay bf16 from here to final_conv: every HIP
        # layer hands a view of its output rows to the next
        x = init_conv(unet.init_conv, x, torch.bfloat16 if _HIP_ATTN else torch.float32)
    else:
        x = unet.init_conv(x)
    x = _temporal(unet.init_temporal_attn, x, bias)
    r = x
    half = unet.dim // 2
    freq = torch.exp(torch.arange(half, device=device) * -(math.log(10000) / (half - 1)))
    ang = time[:, None] * freq[None, :]
    t = torch.cat((ang.sin(), ang.cos()), dim=-1)
    t = unet.time_mlp[3](F.gelu(unet.time_mlp[1](t)))
    if unet.has_cond:
        m = _mask((batch,), null_cond_prob, device)
        t = torch.cat((t, torch.where(m[:, None], unet.null_cond_emb, cond)), dim=-1)
    ctx = None
    if unet.has_frame_cross_attn and frame_cond_features is not None:
        ctx = frame_cond_features
        if null_frame_cond_prob > 0:
            ctx = ctx * (1 - _mask((batch,), null_frame_cond_prob, device)[:, None, None].float())

    if _HIP_RESNET and x.is_cuda:
        from .train_hip import time_tables      # the blocks' time MLPs as one Linear
        with time_tables(unet, t):
            return _unet_body(unet, x, r, t, ctx, bias, focus)
    return _unet_body(unet, x, r, t, ctx, bias, focus)


def _unet_body(unet, x, r, t, ctx, bias, focus):
    skips = []
    for i, (b1, b2, sattn, tattn, down) in enumerate(unet.downs):
        x = _resnet(b2, _resnet(b1, x, t), t)
        if unet.use_sparse_linear_attn:
            x = _linear_attention(sattn, x)
        x = _temporal(tattn, x, bias, focus)
        if ctx is not None:
            x = _cross_attention(unet.frame_cross_attns_down[i], x, ctx)
        skips.append(x)
        x = _resample(down, x)
    x = _resnet(unet.mid_block1, x, t)
    x = _spatial(unet.mid_spatial_attn, x)
    x = _temporal(unet.mid_temporal_attn, x, bias, focus)
    x = _resnet(unet.mid_block2, x, t)
    for i, (b1, b2, sattn, tattn, up) in enumerate(unet.ups):
        x = torch.cat((x, skips.pop()), dim=1)
        x = _resnet(b2, _resnet(b1, x, t), t)
        if unet.use_sparse_linear_attn:
            x = _linear_attention(sattn, x)
        x = _temporal(tattn, x, bias, focus)
        if ctx is not None:
            x = _cross_attention(unet.frame_cross_attns_up[i], x, ctx)
        x = _resample(up, x)
    x = torch.cat((x, r), dim=1)
    x = _resnet(unet.final_conv[0], x, None)
    if _HIP_RESNET and x.is_cuda:
        from .train_hip import pointwise_conv   # weight gradient on the HIP kernels
        return pointwise_conv(unet.final_conv[1], x)
    return unet.final_conv[1](x)


def guidance_variant_forward(unet, x, time):
    """Unet3D.forward of the reference's video_diffusion_guidance.py (:596-635) as a differentiable function of x and of
    the parameters of a guidance_variant.Unet3D: the reference's guidance sampler differentiates the reconstruction loss
    through this denoiser (video_diffusion_guidance.py:725-761, 860-900).  Used only when a gradient is asked for;
    sampling runs on the HIP plan.  `unet.train_backend == "hip"`: the layers run forward and backward on the HIP kernels
    (train_hip_variant.py) over a bf16 activation stream; otherwise torch ops in float."""
    global _HIP_RESNET, _HIP_ATTN
    hip = getattr(unet, "train_backend", None) == "hip" and x.is_cuda
    prev = (_HIP_RESNET, _HIP_ATTN)
    _HIP_RESNET, _HIP_ATTN = hip, hip
    try:
        return _guidance_variant_forward(unet, x, time, hip)
    finally:
        _HIP_RESNET, _HIP_ATTN = prev


def _guidance_variant_forward(unet, x, time, hip):
    bias = _pos_bias(unet, x.shape[2], x.device)
    if hip:
        from . import train_hip_variant as HV
        from .train_hip import init_conv, pointwise_conv

    def tattn_ops(res, x):
        # Residual(PreNorm(TemporalAttention)), :390-428: attention over the frames of every position
        b, c, f, h, w = x.shape
        y = _chan_layernorm(x, res.fn.norm).permute(0, 3, 4, 2, 1).reshape(b, h * w, f, c)
        return _attention(res.fn.fn, y, bias).reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2) + x

    def sattn_ops(res, x):
        # Residual(PreNorm(SpatialAttention)), :431-459: attention over the positions of every frame
        b, c, f, h, w = x.shape
        y = _chan_layernorm(x, res.fn.norm).permute(0, 2, 3, 4, 1).reshape(b, f, h * w, c)
        return _attention(res.fn.fn, y).reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3) + x

    def tattn(res, x):
        if hip:
            rot = _rot_tables(getattr(res.fn.fn, "rotary_emb", None), x.shape[2], x.device)
            return HV.temporal_attention(res, x, bias, rot, lambda y: tattn_ops(res, y))
        return tattn_ops(res, x)

    def sattn(res, x):
        return HV.spatial_attention(res, x, lambda y: sattn_ops(res, y)) if hip else sattn_ops(res, x)

    def ts(tb, x):
        # TemporalSpatialBlock, :334-344
        h = F.silu(tb.temporal_norm(tb.temporal_conv(x)))
        return F.silu(tb.spatial_norm(tb.spatial_conv(h)))

    def rb(blk, x, temb=None):
        # ResnetBlock, :370-388: the time embedding acts between the two blocks
        if hip:
            return HV.resnet_block(blk, x, temb)
        res = blk.res_conv(x)
        h = ts(blk.ts_block1, x)
        if blk.mlp is not None and temb is not None:
            scale, shift = blk.mlp(temb)[:, :, None, None, None].chunk(2, dim=1)
            h = h * (scale + 1) + shift
        return ts(blk.ts_block2, h) + res

    x = init_conv(unet.init_conv, x, torch.bfloat16) if hip else unet.init_conv(x)
    x = tattn(unet.init_temporal_attn, x)
    r = x
    half = unet.dim // 2
    freq = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
    ang = time[:, None] * freq[None, :]
    t = unet.time_mlp[3](F.gelu(unet.time_mlp[1](torch.cat((ang.sin(), ang.cos()), dim=-1))))
    skips = []
    for b1, b2, ta, sa, down in unet.downs:
        x = sattn(sa, tattn(ta, rb(b2, rb(b1, x, t), t)))
        skips.append(x)
        x = _resample(down, x)
    x = rb(unet.mid_block1, x, t)
    x = rb(unet.mid_block2, sattn(unet.mid_spatial_attn, tattn(unet.mid_temporal_attn, x)), t)
    for b1, b2, ta, sa, up in unet.ups:
        x = torch.cat((x, skips.pop()), dim=1)
        x = _resample(up, sattn(sa, tattn(ta, rb(b2, rb(b1, x, t), t))))
    x = torch.cat((x, r), dim=1)
    x = rb(unet.final_conv[0], x)                                 # :591-594, 635: the final block gets no embedding
    return pointwise_conv(unet.final_conv[1], x) if hip else unet.final_conv[1](x)
