"""CPU oracle: a from-scratch fp32 restatement of the reference denoising path.

TEST INFRASTRUCTURE.  Pinned against the real reference by the fixtures under
``tests/golden`` (made by ``oracle/make_golden.py`` from the reference imported
in the build container).  The one third-party piece of arithmetic on the path,
``rotary_embedding_torch.RotaryEmbedding`` (version unpinned, not installed),
is restated from the reference tree's own definition
(``video_diffusion_guidance.py:182-198``, the "half split" convention); against
the PyPI library itself this boundary is *parity unpinned*.

All citations are to files under ``/root/reference``; ``vdp.py`` abbreviates
``video_diffusion_pytorch/video_diffusion_pytorch.py``.

Everything is functional: weights come in as a flat ``state_dict`` (the
reference's own keys), activations are channels-first ``(B, C, F, H, W)``
float32 tensors exactly as in the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# rounding emulation (oracle/rounding_study.py): None = plain fp32, the pinned restatement.  Otherwise an object
# with two callables: EMU.w(t) is applied to every weight of a convolution / projection of the denoiser, EMU.a(t)
# to every activation where a reduced-precision implementation stores or feeds one (layer outputs, matrix operands).
# --------------------------------------------------------------------------
EMU = None


def _rw(t: Tensor) -> Tensor:
    return t if EMU is None else EMU.w(t)


def _ra(t: Tensor) -> Tensor:
    return t if EMU is None else EMU.a(t)


# --------------------------------------------------------------------------
# configuration (mirrors Unet3D.__init__ arguments, vdp.py:616-631)
# --------------------------------------------------------------------------
@dataclass
class UnetCfg:
    dim: int
    dim_mults: Tuple[int, ...] = (1, 2, 4, 8)
    channels: int = 4
    out_dim: Optional[int] = None
    attn_heads: int = 8
    attn_dim_head: int = 32
    cond_dim: Optional[int] = None
    init_dim: Optional[int] = None
    init_kernel_size: int = 7
    use_sparse_linear_attn: bool = True
    num_buckets: int = 32          # RelativePositionBias default, vdp.py:278
    max_distance: int = 32         # instance at vdp.py:638
    rotary: str = "half"           # "half": the reference tree's own definition; "interleaved": see rotary_interleaved

    @property
    def dims(self):
        init_dim = self.init_dim or self.dim
        return [init_dim] + [self.dim * m for m in self.dim_mults]   # vdp.py:647

    @property
    def in_out(self):
        d = self.dims
        return list(zip(d[:-1], d[1:]))                              # vdp.py:648


# --------------------------------------------------------------------------
# A5  relative position bias (vdp.py:274-312)
# --------------------------------------------------------------------------
def relpos_bucket(n: int, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """int64 [n, n] bucket ids; entry (i, j) is for query i, key j.

    vdp.py:286-304 with ``relative_position = k_pos - q_pos`` (vdp.py:309).
    """
    q = torch.arange(n).view(n, 1)
    k = torch.arange(n).view(1, n)
    neg = -(k - q)                       # "n = -relative_position"
    half = num_buckets // 2
    out = (neg < 0).long() * half
    dist = neg.abs()
    max_exact = half // 2
    small = dist < max_exact
    large = max_exact + (
        torch.log(dist.float() / max_exact) / math.log(max_distance / max_exact) * (half - max_exact)
    ).long()
    large = torch.clamp(large, max=half - 1)
    return out + torch.where(small, dist, large)


def relpos_bias(emb_weight: Tensor, n: int, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """[heads, n, n] bias table (vdp.py:306-312)."""
    b = relpos_bucket(n, num_buckets, max_distance)
    return emb_weight[b].permute(2, 0, 1).contiguous()


# --------------------------------------------------------------------------
# A4  time embedding (vdp.py:339-351, 651-656)
# --------------------------------------------------------------------------
def sinusoidal_emb(t: Tensor, dim: int) -> Tensor:
    half = dim // 2
    step = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half) * -step)
    ang = t[:, None] * freq[None, :]          # integer t promotes to float32 here
    return torch.cat((ang.sin(), ang.cos()), dim=-1)


def time_mlp(sd: SD, t: Tensor, dim: int) -> Tensor:
    e = sinusoidal_emb(t, dim)
    e = F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    e = F.gelu(e)                                  # nn.GELU() default = exact erf
    return F.linear(e, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


# --------------------------------------------------------------------------
# A2/A3  Block: (1,3,3) conv + RMSNorm + scale/shift + SiLU (vdp.py:391-427)
# --------------------------------------------------------------------------
def _per_frame(x: Tensor) -> Tuple[Tensor, Tuple[int, int]]:
    b, c, f, h, w = x.shape
    return x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w), (b, f)


def _from_frames(y: Tensor, bf: Tuple[int, int]) -> Tensor:
    b, f = bf
    n, c, h, w = y.shape
    return y.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)


def conv_1kk(x: Tensor, w: Tensor, bias: Optional[Tensor], stride: int = 1, pad: int = 1, emu: bool = True) -> Tensor:
    """Conv3d with kernel (1,k,k): a 2-D conv applied to every frame.  emu=False: never rounded by the rounding
    emulation (the first and the last convolution of the denoiser run in fp32 arithmetic in every build)."""
    if not emu:
        xf, bf = _per_frame(x)
        return _from_frames(F.conv2d(xf, w[:, :, 0], bias, stride=stride, padding=pad), bf)
    xf, bf = _per_frame(_ra(x))
    y = F.conv2d(xf, _rw(w[:, :, 0]), bias, stride=stride, padding=pad)
    return _from_frames(y, bf)


def rmsnorm_c(x: Tensor, gamma: Tensor) -> Tensor:
    """vdp.py:391-398: F.normalize over C (eps 1e-12 on the norm) * sqrt(C) * gamma."""
    c = x.shape[1]
    nrm = x.pow(2).sum(dim=1, keepdim=True).sqrt().clamp_min(1e-12)
    return x / nrm * (c ** 0.5) * gamma.view(1, c, 1, 1, 1)


def block(sd: SD, p: str, x: Tensor, scale_shift: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    y = conv_1kk(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    y = rmsnorm_c(y, sd[p + ".norm.gamma"])
    if scale_shift is not None:
        sc, sh = scale_shift
        y = y * (sc + 1) + sh
    return _ra(F.silu(y)) if EMU is not None and getattr(EMU, "block_out", True) else F.silu(y)


def resnet_block(sd: SD, p: str, x: Tensor, temb: Optional[Tensor]) -> Tensor:
    """vdp.py:429-452."""
    ss = None
    if (p + ".mlp.1.weight") in sd:
        assert temb is not None
        e = F.linear(F.silu(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])
        e = e[:, :, None, None, None]
        ss = e.chunk(2, dim=1)
    h = block(sd, p + ".block1", x, ss)
    h = block(sd, p + ".block2", h)
    if (p + ".temporal_conv.weight") in sd:      # variant of version1diffusion.py:308,322: (3,1,1) conv over frames
        h = F.conv3d(h, sd[p + ".temporal_conv.weight"], sd[p + ".temporal_conv.bias"], padding=(1, 0, 0))
    if (p + ".res_conv.weight") in sd:
        res = conv_1kk(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"], pad=0)
    else:
        res = x
    return _ra(h + res)


# --------------------------------------------------------------------------
# channel LayerNorm (vdp.py:380-389): biased variance, eps inside the sqrt, gamma only
# --------------------------------------------------------------------------
def layernorm_c(x: Tensor, gamma: Tensor, eps: float = 1e-5) -> Tensor:
    mean = x.mean(dim=1, keepdim=True)
    var = (x - mean).pow(2).mean(dim=1, keepdim=True)
    return (x - mean) / (var + eps).sqrt() * gamma.view(1, -1, 1, 1, 1)


# --------------------------------------------------------------------------
# rotary, in-tree "half split" convention (video_diffusion_guidance.py:182-198)
# --------------------------------------------------------------------------
def rotary_half(t: Tensor, rot_dim: int) -> Tensor:
    """t: [..., n, d] with d == rot_dim.  Positions are 0..n-1."""
    n, d = t.shape[-2], t.shape[-1]
    assert d == rot_dim, "half-split rotary needs rot_dim == head dim"
    inv_freq = 1.0 / (10000 ** (torch.arange(0, rot_dim, 2).float() / rot_dim))
    ang = torch.arange(n).float()[:, None] * inv_freq[None, :]
    ang = torch.cat((ang, ang), dim=-1)
    t1, t2 = t.chunk(2, dim=-1)
    return t * ang.cos() + torch.cat((-t2, t1), dim=-1) * ang.sin()


def rotary_interleaved(t: Tensor, rot_dim: int) -> Tensor:
    """The pairing of the PyPI ``rotary_embedding_torch`` library as recalled (PARITY UNPINNED: the
    library is not installed and the reference pins no version): same frequencies as rotary_half, but
    feature 2i rotates with feature 2i+1 instead of i with i + d/2.  t: [..., n, d], d == rot_dim."""
    n, d = t.shape[-2], t.shape[-1]
    assert d == rot_dim
    inv_freq = 1.0 / (10000 ** (torch.arange(0, rot_dim, 2).float() / rot_dim))
    ang = torch.arange(n).float()[:, None] * inv_freq[None, :]
    ang = ang.repeat_interleave(2, dim=-1)                       # f0 f0 f1 f1 ...
    x1, x2 = t[..., 0::2], t[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).reshape(t.shape)        # (-x1, x0, -x3, x2, ...)
    return t * ang.cos() + rot * ang.sin()


# --------------------------------------------------------------------------
# A6/A9  softmax attention over a token axis (vdp.py:546-612)
# --------------------------------------------------------------------------
def attention_tokens(x: Tensor, w_qkv: Tensor, w_out: Tensor, heads: int, dim_head: int,
                     rotary: bool, pos_bias: Optional[Tensor], focus: Optional[Tensor] = None) -> Tensor:
    """x: [..., n, C] -> [..., n, C].  focus: bool [b] (x is [b, ., n, C]) — the focus-present mask of
    vdp.py:590-600: in flagged samples a token attends only to itself.  All-False on the sampling path."""
    lead = x.shape[:-2]
    n = x.shape[-2]
    qkv = F.linear(_ra(x), _rw(w_qkv))
    q, k, v = qkv.chunk(3, dim=-1)

    def split(t):
        return t.reshape(*lead, n, heads, dim_head).transpose(-2, -3)   # [..., h, n, d]

    q, k, v = split(q), split(k), split(v)
    q = q * dim_head ** -0.5
    if rotary:
        rot = rotary_interleaved if rotary == "interleaved" else rotary_half
        q = rot(q, min(32, dim_head))
        k = rot(k, min(32, dim_head))
    sim = _ra(q) @ _ra(k).transpose(-1, -2)
    if pos_bias is not None:
        sim = sim + pos_bias
    if focus is not None and bool(focus.any()):
        eye = torch.eye(n, dtype=torch.bool)
        keep = torch.where(focus.view(-1, 1, 1, 1, 1), eye.view(1, 1, 1, n, n), torch.ones(1, 1, 1, n, n, dtype=torch.bool))
        sim = sim.masked_fill(~keep, -torch.finfo(sim.dtype).max)
    sim = sim - sim.amax(dim=-1, keepdim=True)
    attn = sim.softmax(dim=-1)
    out = _ra(attn) @ _ra(v)
    out = out.transpose(-2, -3).reshape(*lead, n, heads * dim_head)
    return F.linear(_ra(out), _rw(w_out))


def temporal_attn_residual(sd: SD, p: str, x: Tensor, cfg: UnetCfg, pos_bias: Tensor,
                           focus: Optional[Tensor] = None) -> Tensor:
    """Residual(PreNorm(EinopsToAndFrom('b c f h w','b (h w) f c', Attention))) — vdp.py:636,645."""
    b, c, f, h, w = x.shape
    y = layernorm_c(x, sd[p + ".fn.norm.gamma"])
    tok = y.permute(0, 3, 4, 2, 1).reshape(b, h * w, f, c)
    o = attention_tokens(tok, sd[p + ".fn.fn.fn.to_qkv.weight"], sd[p + ".fn.fn.fn.to_out.weight"],
                         cfg.attn_heads, cfg.attn_dim_head, cfg.rotary, pos_bias, focus)
    o = o.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2)
    return _ra(o + x)


def spatial_attn_residual(sd: SD, p: str, x: Tensor, cfg: UnetCfg) -> Tensor:
    """mid_spatial_attn: tokens = h*w per frame, heads = attn_heads, dim_head 32, no rotary/bias (vdp.py:708-710)."""
    b, c, f, h, w = x.shape
    y = layernorm_c(x, sd[p + ".fn.norm.gamma"])
    tok = y.permute(0, 2, 3, 4, 1).reshape(b, f, h * w, c)
    o = attention_tokens(tok, sd[p + ".fn.fn.fn.to_qkv.weight"], sd[p + ".fn.fn.fn.to_out.weight"],
                         cfg.attn_heads, 32, False, None)
    o = o.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)
    return _ra(o + x)


# --------------------------------------------------------------------------
# A7  spatial linear attention (vdp.py:454-479), heads = attn_heads, dim_head 32
# --------------------------------------------------------------------------
def sta_residual(sd: SD, p: str, x: Tensor, cfg: UnetCfg) -> Tensor:
    """Residual(PreNorm(SpatioTemporalAttention)) of version1diffusion.py:325-345, 431: full attention over the
    f*h*w tokens of a sample, heads = attn_heads, dim_head 32."""
    b, c, f, h, w = x.shape
    y = layernorm_c(x, sd[p + ".fn.norm.gamma"])
    tok = y.permute(0, 2, 3, 4, 1).reshape(b, 1, f * h * w, c)
    o = attention_tokens(tok, sd[p + ".fn.fn.to_qkv.weight"], sd[p + ".fn.fn.to_out.weight"], cfg.attn_heads, 32, False, None)
    return o.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3) + x


def sla_residual(sd: SD, p: str, x: Tensor, cfg: UnetCfg) -> Tensor:
    if sd[p + ".fn.fn.to_qkv.weight"].dim() == 2:      # the variant's Linear projection
        return sta_residual(sd, p, x, cfg)
    b, c, f, h, w = x.shape
    heads, dh = cfg.attn_heads, 32
    y = layernorm_c(x, sd[p + ".fn.norm.gamma"])
    yf, bf = _per_frame(y)
    qkv = F.conv2d(_ra(yf), _rw(sd[p + ".fn.fn.to_qkv.weight"]))
    q, k, v = qkv.reshape(b * f, 3, heads, dh, h * w).unbind(dim=1)    # each [bf, heads, d, n]
    q = q.softmax(dim=-2) * dh ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", _ra(k), _ra(v))
    out = torch.einsum("bhde,bhdn->bhen", _ra(ctx), _ra(q)).reshape(b * f, heads * dh, h, w)
    out = F.conv2d(_ra(out), _rw(sd[p + ".fn.fn.to_out.weight"]), sd[p + ".fn.fn.to_out.bias"])
    return _ra(_from_frames(out, bf) + x)


# --------------------------------------------------------------------------
# N2  cross-attention to previous-frame features (vdp.py:480-528, wiring :664-682, :803-805, :824-826)
# --------------------------------------------------------------------------
def cross_attn_residual(sd: SD, p: str, x: Tensor, ctx: Tensor, heads: int, dim_head: int) -> Tensor:
    """Residual(PreNorm(CrossAttention)): queries are all f*h*w positions of a sample, keys/values the
    sample's context tokens ctx [B, L, context_dim]; plain softmax, to_out has a bias."""
    b, c, f, h, w = x.shape
    y = layernorm_c(x, sd[p + ".fn.norm.gamma"])
    tok = y.permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c)
    q = F.linear(tok, sd[p + ".fn.fn.to_q.weight"])
    k = F.linear(ctx, sd[p + ".fn.fn.to_k.weight"])
    v = F.linear(ctx, sd[p + ".fn.fn.to_v.weight"])

    def split(t):
        return t.reshape(b, t.shape[1], heads, dim_head).transpose(1, 2)      # [b, heads, n, d]

    q, k, v = split(q), split(k), split(v)
    attn = ((q @ k.transpose(-1, -2)) * dim_head ** -0.5).softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(b, f * h * w, heads * dim_head)
    out = F.linear(out, sd[p + ".fn.fn.to_out.weight"], sd[p + ".fn.fn.to_out.bias"])
    return out.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3) + x


def frame_encoder(sd: SD, prev_latents: Tensor) -> Tensor:
    """FrameConditioningEncoder.forward (vdp.py:239-269): three (1,3,3) convs (strides 1, 2, 2) with SiLU,
    then 'b c f h w -> b (f h w) c'.  sd keys are those of the module ('feature_extractor.N.*')."""
    h = prev_latents
    for i, stride in ((0, 1), (2, 2), (4, 2)):
        h = F.silu(F.conv3d(h, sd[f"feature_extractor.{i}.weight"], sd[f"feature_extractor.{i}.bias"],
                            stride=(1, stride, stride), padding=(0, 1, 1)))
    b, c, f, hh, ww = h.shape
    return h.permute(0, 2, 3, 4, 1).reshape(b, f * hh * ww, c)


# --------------------------------------------------------------------------
# A8  resamplers (vdp.py:374-378)
# --------------------------------------------------------------------------
def downsample(sd: SD, p: str, x: Tensor) -> Tensor:
    return _ra(conv_1kk(x, sd[p + ".weight"], sd[p + ".bias"], stride=2, pad=1))


def upsample(sd: SD, p: str, x: Tensor) -> Tensor:
    xf, bf = _per_frame(_ra(x))
    y = F.conv_transpose2d(xf, _rw(sd[p + ".weight"][:, :, 0]), sd[p + ".bias"], stride=2, padding=1)
    return _ra(_from_frames(y, bf))


# --------------------------------------------------------------------------
# Unet3D.forward on the sampling path (vdp.py:756-831)
# --------------------------------------------------------------------------
def unet3d_forward(sd: SD, cfg: UnetCfg, x: Tensor, time: Tensor,
                   cond: Optional[Tensor] = None, null_cond: bool = False,
                   frame_cond: Optional[Tensor] = None, frame_keep: Optional[Tensor] = None,
                   focus: Optional[Tensor] = None) -> Tensor:
    """``focus`` bool [B] is the focus-present mask (all-False is what p_sample passes, vdp.py:978-983); it
    reaches every temporal attention except init_temporal_attn (vdp.py:773, 800, 812, 821).

    ``null_cond`` selects null_cond_prob = 1 (True) or 0 (False) — the two values
    forward_with_cond_scale uses (vdp.py:739,747).  ``frame_cond`` [B, L, frame_cond_dim] switches on the
    cross-attention layers (vdp.py:803-805, 824-826); ``frame_keep`` [B] of 0/1 is one minus the
    null-conditioning mask of vdp.py:782-789 (the random draw is the caller's).
    """
    if frame_cond is not None and frame_keep is not None:
        frame_cond = frame_cond * frame_keep.to(frame_cond.dtype)[:, None, None]
    nf = x.shape[2]
    bias = relpos_bias(sd["time_rel_pos_bias.relative_attention_bias.weight"], nf,
                       cfg.num_buckets, cfg.max_distance)

    x = _ra(conv_1kk(x, sd["init_conv.weight"], sd["init_conv.bias"], pad=cfg.init_kernel_size // 2, emu=False))
    x = temporal_attn_residual(sd, "init_temporal_attn", x, cfg, bias)
    r = x
    t = time_mlp(sd, time, cfg.dim)
    if cfg.cond_dim is not None:
        assert cond is not None, "cond must be passed in if cond_dim specified"
        c = sd["null_cond_emb"].expand(x.shape[0], -1) if null_cond else cond
        t = torch.cat((t, c), dim=-1)

    skips = []
    nres = len(cfg.in_out)
    for i in range(nres):
        p = f"downs.{i}"
        x = resnet_block(sd, p + ".0", x, t)
        x = resnet_block(sd, p + ".1", x, t)
        if cfg.use_sparse_linear_attn:
            x = sla_residual(sd, p + ".2", x, cfg)
        x = temporal_attn_residual(sd, p + ".3", x, cfg, bias, focus)
        if frame_cond is not None:
            x = cross_attn_residual(sd, f"frame_cross_attns_down.{i}", x, frame_cond, cfg.attn_heads, cfg.attn_dim_head)
        skips.append(x)
        if i < nres - 1:
            x = downsample(sd, p + ".4", x)

    x = resnet_block(sd, "mid_block1", x, t)
    x = spatial_attn_residual(sd, "mid_spatial_attn", x, cfg)
    x = temporal_attn_residual(sd, "mid_temporal_attn", x, cfg, bias, focus)
    x = resnet_block(sd, "mid_block2", x, t)

    for i in range(nres):
        p = f"ups.{i}"
        x = torch.cat((x, skips.pop()), dim=1)
        x = resnet_block(sd, p + ".0", x, t)
        x = resnet_block(sd, p + ".1", x, t)
        if cfg.use_sparse_linear_attn:
            x = sla_residual(sd, p + ".2", x, cfg)
        x = temporal_attn_residual(sd, p + ".3", x, cfg, bias, focus)
        if frame_cond is not None:
            x = cross_attn_residual(sd, f"frame_cross_attns_up.{i}", x, frame_cond, cfg.attn_heads, cfg.attn_dim_head)
        if i < nres - 1:
            x = upsample(sd, p + ".4", x)

    x = torch.cat((x, r), dim=1)
    x = resnet_block(sd, "final_conv.0", x, None)
    return conv_1kk(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"], pad=0, emu=False)


# --------------------------------------------------------------------------
# N3  the video_diffusion_guidance.py variant of the denoiser (:319-388, :390-459, :515-635): factorised
# temporal (depthwise) -> spatial blocks with GroupNorm(8), temporal then per-frame spatial softmax attention
# --------------------------------------------------------------------------
def _g_ts_block(sd: SD, p: str, x: Tensor) -> Tensor:
    k = sd[p + ".temporal_conv.weight"].shape[2]
    h = F.conv3d(x, sd[p + ".temporal_conv.weight"], sd[p + ".temporal_conv.bias"], padding=(k // 2, 0, 0), groups=x.shape[1])
    h = F.silu(F.group_norm(h, 8, sd[p + ".temporal_norm.weight"], sd[p + ".temporal_norm.bias"], eps=1e-5))
    h = conv_1kk(h, sd[p + ".spatial_conv.weight"], sd[p + ".spatial_conv.bias"])
    return F.silu(F.group_norm(h, 8, sd[p + ".spatial_norm.weight"], sd[p + ".spatial_norm.bias"], eps=1e-5))


def _g_resnet(sd: SD, p: str, x: Tensor, temb: Optional[Tensor]) -> Tensor:
    res = x
    if (p + ".res_conv.0.weight") in sd:
        w = sd[p + ".res_conv.0.weight"]
        res = F.conv3d(x, w, sd[p + ".res_conv.0.bias"], padding=(w.shape[2] // 2, 0, 0))
        res = F.group_norm(res, 8, sd[p + ".res_conv.1.weight"], sd[p + ".res_conv.1.bias"], eps=1e-5)
    h = _g_ts_block(sd, p + ".ts_block1", x)
    if (p + ".mlp.1.weight") in sd and temb is not None:
        e = F.linear(F.silu(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])[:, :, None, None, None]
        scale, shift = e.chunk(2, dim=1)
        h = h * (scale + 1) + shift
    return _g_ts_block(sd, p + ".ts_block2", h) + res


def _g_temporal(sd: SD, p: str, x: Tensor, cfg: UnetCfg, pos_bias: Tensor) -> Tensor:
    b, c, f, h, w = x.shape
    y = layernorm_c(x, sd[p + ".fn.norm.gamma"])
    tok = y.permute(0, 3, 4, 2, 1).reshape(b, h * w, f, c)
    o = attention_tokens(tok, sd[p + ".fn.fn.to_qkv.weight"], sd[p + ".fn.fn.to_out.weight"],
                         cfg.attn_heads, cfg.attn_dim_head, cfg.rotary, pos_bias)
    return o.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2) + x


def _g_spatial(sd: SD, p: str, x: Tensor, cfg: UnetCfg) -> Tensor:
    b, c, f, h, w = x.shape
    y = layernorm_c(x, sd[p + ".fn.norm.gamma"])
    tok = y.permute(0, 2, 3, 4, 1).reshape(b, f, h * w, c)
    o = attention_tokens(tok, sd[p + ".fn.fn.to_qkv.weight"], sd[p + ".fn.fn.to_out.weight"],
                         cfg.attn_heads, cfg.attn_dim_head, False, None)
    return o.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3) + x


def unet3d_guidance_forward(sd: SD, cfg: UnetCfg, x: Tensor, time: Tensor) -> Tensor:
    """Unet3D.forward of video_diffusion_guidance.py:596-635 (cfg.max_distance must be 300, :526)."""
    bias = relpos_bias(sd["time_rel_pos_bias.relative_attention_bias.weight"], x.shape[2], cfg.num_buckets, cfg.max_distance)
    x = conv_1kk(x, sd["init_conv.weight"], sd["init_conv.bias"], pad=cfg.init_kernel_size // 2)
    x = _g_temporal(sd, "init_temporal_attn", x, cfg, bias)
    r = x
    t = time_mlp(sd, time, cfg.dim)
    skips = []
    nres = len(cfg.in_out)
    for i in range(nres):
        p = f"downs.{i}"
        x = _g_resnet(sd, p + ".0", x, t)
        x = _g_resnet(sd, p + ".1", x, t)
        x = _g_temporal(sd, p + ".2", x, cfg, bias)
        x = _g_spatial(sd, p + ".3", x, cfg)
        skips.append(x)
        if i < nres - 1:
            x = downsample(sd, p + ".4", x)
    x = _g_resnet(sd, "mid_block1", x, t)
    x = _g_temporal(sd, "mid_temporal_attn", x, cfg, bias)
    x = _g_spatial(sd, "mid_spatial_attn", x, cfg)
    x = _g_resnet(sd, "mid_block2", x, t)
    for i in range(nres):
        p = f"ups.{i}"
        x = torch.cat((x, skips.pop()), dim=1)
        x = _g_resnet(sd, p + ".0", x, t)
        x = _g_resnet(sd, p + ".1", x, t)
        x = _g_temporal(sd, p + ".2", x, cfg, bias)
        x = _g_spatial(sd, p + ".3", x, cfg)
        if i < nres - 1:
            x = upsample(sd, p + ".4", x)
    x = torch.cat((x, r), dim=1)
    x = _g_resnet(sd, "final_conv.0", x, None)      # nn.Sequential calls it without the time embedding (:635)
    return conv_1kk(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"], pad=0)


def unet3d_forward_cfg(sd: SD, cfg: UnetCfg, x: Tensor, time: Tensor, cond: Tensor,
                       cond_scale: float = 2.0) -> Tensor:
    """forward_with_cond_scale, text-CFG branch only (vdp.py:731-754)."""
    logits = unet3d_forward(sd, cfg, x, time, cond, null_cond=False)
    if cond_scale == 1:
        return logits
    null_logits = unet3d_forward(sd, cfg, x, time, cond, null_cond=True)
    return null_logits + (logits - null_logits) * cond_scale


# --------------------------------------------------------------------------
# A11  schedule + posterior step (vdp.py:835-838, 866-872, 901-928, 939-1002)
# --------------------------------------------------------------------------
def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> Tensor:
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.9999)


def diffusion_buffers(timesteps: int) -> Dict[str, Tensor]:
    """The 12 float32 buffers GaussianDiffusion registers (vdp.py:913-928), built in float64."""
    betas = cosine_beta_schedule(timesteps)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    bufs = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": ac.sqrt(),
        "sqrt_one_minus_alphas_cumprod": (1.0 - ac).sqrt(),
        "log_one_minus_alphas_cumprod": (1.0 - ac).log(),
        "sqrt_recip_alphas_cumprod": (1.0 / ac).sqrt(),
        "sqrt_recipm1_alphas_cumprod": (1.0 / ac - 1).sqrt(),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": pv.clamp(min=1e-20).log(),
        "posterior_mean_coef1": betas * ac_prev.sqrt() / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * alphas.sqrt() / (1.0 - ac),
    }
    return {k: v.to(torch.float32) for k, v in bufs.items()}


def _bcast(a: Tensor, t: Tensor, x: Tensor) -> Tensor:
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (x.dim() - 1)))


def p_sample_step(bufs: Dict[str, Tensor], x: Tensor, t: Tensor, eps: Tensor, z: Tensor,
                  clip_denoised: bool = True) -> Tensor:
    """One p_sample given the model output eps and the noise draw z (vdp.py:985-1002, static clip)."""
    x0 = _bcast(bufs["sqrt_recip_alphas_cumprod"], t, x) * x - _bcast(bufs["sqrt_recipm1_alphas_cumprod"], t, x) * eps
    if clip_denoised:
        x0 = x0.clamp(-1.0, 1.0) / 1.0
    mean = _bcast(bufs["posterior_mean_coef1"], t, x) * x0 + _bcast(bufs["posterior_mean_coef2"], t, x) * x
    logvar = _bcast(bufs["posterior_log_variance_clipped"], t, x)
    nonzero = (1 - (t == 0).float()).reshape(x.shape[0], *((1,) * (x.dim() - 1)))
    return mean + nonzero * (0.5 * logvar).exp() * z


def frame_cfg_step(sd: SD, cfg: UnetCfg, bufs: Dict[str, Tensor], x: Tensor, t: Tensor, frame_cond: Tensor,
                   scale: float, z: Tensor) -> Tensor:
    """One step of the frame-conditioned loop in GaussianDiffusion.sample (vdp.py:1055-1080): two
    evaluations (conditioning kept / zeroed), guidance, clamp, posterior; the noise draw z is used for
    i > 0 and replaced by zeros at i == 0, which p_sample_step's nonzero mask reproduces."""
    if scale == 1.0:
        return p_sample_step(bufs, x, t, unet3d_forward(sd, cfg, x, t, frame_cond=frame_cond), z)
    ones = torch.ones(x.shape[0])
    e_c = unet3d_forward(sd, cfg, x, t, frame_cond=frame_cond, frame_keep=ones)
    e_u = unet3d_forward(sd, cfg, x, t, frame_cond=frame_cond, frame_keep=0 * ones)
    return p_sample_step(bufs, x, t, e_u + scale * (e_c - e_u), z)


def p_sample_loop(sd: SD, cfg: UnetCfg, bufs: Dict[str, Tensor], x_T: Tensor, noises: Sequence[Tensor],
                  steps: Optional[int] = None) -> Tensor:
    """p_sample_loop with injected randomness (vdp.py:1004-1014): ``noises[k]`` is the
    randn_like draw of the k-th executed step (drawn at every step, also t == 0)."""
    T = bufs["betas"].shape[0]
    img = x_T
    b = x_T.shape[0]
    order = list(reversed(range(T)))
    if steps is not None:
        order = order[:steps]
    for k, i in enumerate(order):
        t = torch.full((b,), i, dtype=torch.long)
        eps = unet3d_forward(sd, cfg, img, t)
        img = p_sample_step(bufs, img, t, eps, noises[k])
    return img


# --------------------------------------------------------------------------
# N1  VAE encoder / decoder (vdp.py:83-150): (1,4,4) stride-2 convs / transposed convs,
# GroupNorm + SiLU, 1x1 heads.  sd keys are those of the reference's VAE module.
# --------------------------------------------------------------------------
def _gn_silu(sd: SD, p: str, x: Tensor, groups: int) -> Tensor:
    return F.silu(F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5))


def vae_decode(sd: SD, z: Tensor) -> Tensor:
    """VAEDecoder.forward (vdp.py:122-150): z (B, latent, F, h, w) -> (B, channels, F, 8h, 8w)."""
    d = "decoder.decoder."
    h = F.conv3d(z, sd[d + "0.weight"], sd[d + "0.bias"])
    h = _gn_silu(sd, d + "1", h, 32)
    h = F.conv3d(h, sd[d + "3.weight"], sd[d + "3.bias"], padding=(0, 1, 1))
    h = _gn_silu(sd, d + "4", h, 32)
    h = F.conv_transpose3d(h, sd[d + "6.weight"], sd[d + "6.bias"], stride=(1, 2, 2), padding=(0, 1, 1))
    h = _gn_silu(sd, d + "7", h, 16)
    h = F.conv_transpose3d(h, sd[d + "9.weight"], sd[d + "9.bias"], stride=(1, 2, 2), padding=(0, 1, 1))
    h = _gn_silu(sd, d + "10", h, 8)
    return F.conv_transpose3d(h, sd[d + "12.weight"], sd[d + "12.bias"], stride=(1, 2, 2), padding=(0, 1, 1))


def vae_encode(sd: SD, x: Tensor):
    """VAEEncoder.forward (vdp.py:83-119): x (B, channels, F, H, W) -> mean, clamped logvar at (H/8, W/8)."""
    e = "encoder.encoder."
    h = x
    for i, g in ((0, 8), (3, 16), (6, 32)):
        h = F.conv3d(h, sd[e + f"{i}.weight"], sd[e + f"{i}.bias"], stride=(1, 2, 2), padding=(0, 1, 1))
        h = _gn_silu(sd, e + f"{i + 1}", h, g)
    h = F.conv3d(h, sd[e + "9.weight"], sd[e + "9.bias"], padding=(0, 1, 1))
    h = _gn_silu(sd, e + "10", h, 32)
    mean = F.conv3d(h, sd["encoder.to_mean.weight"], sd["encoder.to_mean.bias"])
    logvar = F.conv3d(h, sd["encoder.to_logvar.weight"], sd["encoder.to_logvar.bias"]).clamp(-30.0, 20.0)
    return mean, logvar
