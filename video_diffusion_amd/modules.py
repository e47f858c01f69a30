"""Parameter containers with the reference's class names, constructor signatures and
state-dict keys (``video_diffusion_pytorch/video_diffusion_pytorch.py:274-729`` of the
reference).  They own ``nn.Parameter``s; the arithmetic of the denoising path lives
in the HIP library and is driven by :class:`video_diffusion_amd.engine.UnetEngine`.  A few of them can also be
called on their own like the reference's modules (``ResnetBlock``, the ``Residual`` attention layers of a level,
``RelativePositionBias``, ``SinusoidalPosEmb``); the rest raise when called outside ``Unet3D``.

``Unet3D.forward`` is the drop-in for the reference's ``Unet3D.forward`` on the sampling
path (no autograd, no frame cross-attention features, focus-present mask all False).
"""
from __future__ import annotations

from functools import partial

import math

import torch
from torch import nn


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def is_odd(n):
    return (n % 2) == 1


BERT_MODEL_DIM = 768   # text.py:9 of the reference


class _HipOnly(nn.Module):
    """Sub-blocks are parameter holders; they are executed fused, through Unet3D."""

    def forward(self, *a, **k):   # pragma: no cover - guard
        raise NotImplementedError(
            f"{type(self).__name__} is executed inside the fused HIP pipeline; call Unet3D.forward")


class RelativePositionBias(_HipOnly):
    # reference: vdp.py:274-312
    def __init__(self, heads=8, num_buckets=32, max_distance=128):
        super().__init__()
        self.num_buckets = num_buckets
        self.max_distance = max_distance
        self.relative_attention_bias = nn.Embedding(num_buckets, heads)

    def forward(self, n, device):
        """The module on its own (reference vdp.py:306-312): bias [heads, n, n]; the integer buckets come from the
        library's host routine (bit-identical to the reference's), the gather is differentiable."""
        import numpy as np
        from . import _lib
        buckets = np.empty((n, n), dtype=np.int32)
        _lib.check(_lib.lib().vdmi_relpos_buckets(n, self.num_buckets, self.max_distance, buckets.ctypes.data),
                   "vdmi_relpos_buckets")
        idx = torch.from_numpy(buckets.astype(np.int64)).to(device)
        return self.relative_attention_bias(idx).permute(2, 0, 1)


class RotaryEmbedding(_HipOnly):
    """Half-split rotary (the definition in the reference tree, video_diffusion_guidance.py:182-198).

    The buffer name follows that definition; checkpoints written with the PyPI
    ``rotary_embedding_torch`` carry a ``freqs`` parameter instead and need ``strict=False``.
    """

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)
        # "half": feature i pairs with i + dim/2 (the tree's definition).  "interleaved": feature 2i pairs with
        # 2i + 1 (the PyPI library's pairing as recalled; not verifiable offline).  The scores q.k are invariant
        # under a common permutation of the head features, so "interleaved" is served by the same kernels with
        # the q / k projection rows de-interleaved at packing time.
        self.convention = "half"


class Residual(_HipOnly):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, *args, **kwargs):
        """The two attention layers of a UNet level called on their own (reference vdp.py:331-337 around PreNorm):
        Residual(PreNorm(SpatialLinearAttention)) and the temporal Residual(PreNorm(EinopsToAndFrom(Attention))) run
        forward and backward on the HIP kernels (train_hip.py); x float (B, C, F, H, W) on the GPU.  The temporal layer
        takes `pos_bias=` and `focus_present_mask=` (bool [B]) as in the reference."""
        pre = self.fn
        core = getattr(pre, "fn", None)
        if isinstance(pre, PreNorm) and isinstance(core, SpatialLinearAttention):
            from .train_hip import spatial_linear_attention
            return spatial_linear_attention(self, x)
        if (isinstance(pre, PreNorm) and isinstance(core, EinopsToAndFrom) and isinstance(core.fn, Attention)
                and core.to_einops == 'b (h w) f c'):
            mask = kwargs.get("focus_present_mask")
            if mask is not None and not bool(torch.as_tensor(mask).any()):
                mask = None                     # vdp.py:590: the mask is skipped when nothing is flagged
            from . import autograd
            from .train_hip import temporal_attention
            rot = autograd._rot_tables(core.fn.rotary_emb, x.shape[2], x.device)
            return temporal_attention(self, x, kwargs.get("pos_bias"), rot,
                                      None if mask is None else torch.as_tensor(mask, device=x.device))
        if (isinstance(pre, PreNorm) and isinstance(core, EinopsToAndFrom) and isinstance(core.fn, Attention)
                and core.to_einops == 'b f (h w) c' and core.fn.rotary_emb is None):
            from .train_hip import spatial_attention          # the mid-block attention over the positions of a frame
            return spatial_attention(self, x)
        return super().forward(x, *args, **kwargs)


class SinusoidalPosEmb(_HipOnly):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        # reference vdp.py:339-351 (a handful of scalar operations per sample: host-side arithmetic, no kernel)
        half = self.dim // 2
        freq = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
        ang = x[:, None] * freq[None, :]
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


def Upsample(dim):
    return nn.ConvTranspose3d(dim, dim, (1, 4, 4), (1, 2, 2), (0, 1, 1))


def Downsample(dim):
    return nn.Conv3d(dim, dim, (1, 4, 4), (1, 2, 2), (0, 1, 1))


class LayerNorm(_HipOnly):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(1, dim, 1, 1, 1))


class RMSNorm(_HipOnly):
    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(dim, 1, 1, 1))


class PreNorm(_HipOnly):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = LayerNorm(dim)


class Block(_HipOnly):
    def __init__(self, dim, dim_out):
        super().__init__()
        self.proj = nn.Conv3d(dim, dim_out, (1, 3, 3), padding=(0, 1, 1))
        self.norm = RMSNorm(dim_out)
        self.act = nn.SiLU()


class ResnetBlock(_HipOnly):
    def __init__(self, dim, dim_out, *, time_emb_dim=None):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, dim_out * 2)) if exists(time_emb_dim) else None
        self.block1 = Block(dim, dim_out)
        self.block2 = Block(dim_out, dim_out)
        self.res_conv = nn.Conv3d(dim, dim_out, 1) if dim != dim_out else nn.Identity()

    def forward(self, x, time_emb=None):
        """The block on its own (reference vdp.py:442-452), differentiable: forward and backward run on the HIP kernels
        (train_hip.py).  x float (B, dim, F, H, W) on the GPU; channel widths powers of two from 64 to 2048."""
        if exists(self.mlp):
            assert exists(time_emb), 'time emb must be passed in'
        from .train_hip import resnet_block
        return resnet_block(self, x, time_emb if exists(self.mlp) else None)


class SpatialLinearAttention(_HipOnly):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        hidden_dim = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden_dim, dim, 1)


class CrossAttention(_HipOnly):
    # reference: vdp.py:480-528.  Runs when frame_cond_features is passed (Plan._cross_attention in engine.py).
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = default(context_dim, query_dim)
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Linear(inner_dim, query_dim)


class EinopsToAndFrom(_HipOnly):
    def __init__(self, from_einops, to_einops, fn):
        super().__init__()
        self.from_einops = from_einops
        self.to_einops = to_einops
        self.fn = fn


class Attention(_HipOnly):
    def __init__(self, dim, heads=4, dim_head=32, rotary_emb=None):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        hidden_dim = dim_head * heads
        self.rotary_emb = rotary_emb
        self.to_qkv = nn.Linear(dim, hidden_dim * 3, bias=False)
        self.to_out = nn.Linear(hidden_dim, dim, bias=False)


class Unet3D(nn.Module):
    """Same constructor and attribute tree as the reference (vdp.py:615-729)."""

    def __init__(
        self,
        dim,
        cond_dim=None,
        out_dim=None,
        dim_mults=(1, 2, 4, 8),
        channels=4,
        attn_heads=8,
        attn_dim_head=32,
        use_bert_text_cond=False,
        init_dim=None,
        init_kernel_size=7,
        use_sparse_linear_attn=True,
        block_type='resnet',
        frame_cond_dim=512,
    ):
        super().__init__()
        self.channels = channels
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.attn_heads = attn_heads
        self.attn_dim_head = attn_dim_head
        self.init_kernel_size = init_kernel_size
        self.use_sparse_linear_attn = use_sparse_linear_attn

        rotary_emb = RotaryEmbedding(min(32, attn_dim_head))
        temporal_attn = lambda dim: EinopsToAndFrom(
            'b c f h w', 'b (h w) f c',
            Attention(dim, heads=attn_heads, dim_head=attn_dim_head, rotary_emb=rotary_emb))

        self.time_rel_pos_bias = RelativePositionBias(heads=attn_heads, max_distance=32)

        init_dim = default(init_dim, dim)
        assert is_odd(init_kernel_size)
        init_padding = init_kernel_size // 2
        self.init_conv = nn.Conv3d(channels, init_dim, (1, init_kernel_size, init_kernel_size),
                                   padding=(0, init_padding, init_padding))
        self.init_temporal_attn = Residual(PreNorm(init_dim, temporal_attn(init_dim)))

        dims = [init_dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.dims = dims

        time_dim = dim * 8
        self.time_mlp = nn.Sequential(
            SinusoidalPosEmb(dim),
            nn.Linear(dim, time_dim),
            nn.GELU(),
            nn.Linear(time_dim, time_dim),
        )

        self.has_cond = exists(cond_dim) or use_bert_text_cond
        cond_dim = BERT_MODEL_DIM if use_bert_text_cond else cond_dim
        self.cond_dim = cond_dim if self.has_cond else None
        self.null_cond_emb = nn.Parameter(torch.randn(1, cond_dim)) if self.has_cond else None

        self.has_frame_cross_attn = exists(frame_cond_dim)
        if self.has_frame_cross_attn:
            self.frame_cross_attns_down = nn.ModuleList([])
            self.frame_cross_attns_up = nn.ModuleList([])
            for dim_out in dims[1:]:
                self.frame_cross_attns_down.append(
                    Residual(PreNorm(dim_out, CrossAttention(dim_out, context_dim=frame_cond_dim,
                                                             heads=attn_heads, dim_head=attn_dim_head))))
            for dim_in in reversed(dims[:-1]):
                self.frame_cross_attns_up.append(
                    Residual(PreNorm(dim_in, CrossAttention(dim_in, context_dim=frame_cond_dim,
                                                            heads=attn_heads, dim_head=attn_dim_head))))

        cond_dim = time_dim + int(cond_dim or 0)

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        num_resolutions = len(in_out)

        block_klass = ResnetBlock
        block_klass_cond = partial(block_klass, time_emb_dim=cond_dim)

        for ind, (dim_in, dim_out) in enumerate(in_out):
            is_last = ind >= (num_resolutions - 1)
            self.downs.append(nn.ModuleList([
                block_klass_cond(dim_in, dim_out),
                block_klass_cond(dim_out, dim_out),
                Residual(PreNorm(dim_out, SpatialLinearAttention(dim_out, heads=attn_heads)))
                if use_sparse_linear_attn else nn.Identity(),
                Residual(PreNorm(dim_out, temporal_attn(dim_out))),
                Downsample(dim_out) if not is_last else nn.Identity(),
            ]))

        mid_dim = dims[-1]
        self.mid_block1 = block_klass_cond(mid_dim, mid_dim)
        spatial_attn = EinopsToAndFrom('b c f h w', 'b f (h w) c', Attention(mid_dim, heads=attn_heads))
        self.mid_spatial_attn = Residual(PreNorm(mid_dim, spatial_attn))
        self.mid_temporal_attn = Residual(PreNorm(mid_dim, temporal_attn(mid_dim)))
        self.mid_block2 = block_klass_cond(mid_dim, mid_dim)

        for ind, (dim_in, dim_out) in enumerate(reversed(in_out)):
            is_last = ind >= (num_resolutions - 1)
            self.ups.append(nn.ModuleList([
                block_klass_cond(dim_out * 2, dim_in),
                block_klass_cond(dim_in, dim_in),
                Residual(PreNorm(dim_in, SpatialLinearAttention(dim_in, heads=attn_heads)))
                if use_sparse_linear_attn else nn.Identity(),
                Residual(PreNorm(dim_in, temporal_attn(dim_in))),
                Upsample(dim_in) if not is_last else nn.Identity(),
            ]))

        out_dim = default(out_dim, channels)
        self.out_dim = out_dim
        self.final_conv = nn.Sequential(block_klass(dim * 2, dim), nn.Conv3d(dim, out_dim, 1))

        self._engine = None
        # None: every forward runs on the HIP kernels (no gradients).  "torch-autograd": a forward that needs
        # gradients runs as differentiable torch ops instead (training only; see autograd.py).
        self.train_backend = None

    def __deepcopy__(self, memo):
        # copies (the Trainer's EMA model) start without the packed-weight cache of the original
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_engine" else copy.deepcopy(v, memo)
        return new

    # -- execution ---------------------------------------------------------------
    def engine(self):
        """The HIP execution plan bound to this module's parameters (built lazily)."""
        if self._engine is None:
            from .engine import UnetEngine
            object.__setattr__(self, "_engine", UnetEngine(self))
        return self._engine

    def weights_changed(self):
        """Tell the HIP engine that parameters were rewritten through `.data` (invisible to autograd's version
        counters): the packed bf16 copy is rebuilt at the next forward.  Optimizer steps, `load_state_dict` and other
        in-place operations on the parameters themselves are noticed without this call, and the sampling loops of
        GaussianDiffusion compare a fingerprint of the values once per loop."""
        if self._engine is not None:
            self._engine.invalidate()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # The rotary pairing is not stored in a checkpoint, but the key the checkpoint uses for the frequencies
        # tells which implementation wrote it: `inv_freq` (buffer) = the half-split class of the reference tree
        # (video_diffusion_guidance.py:182-198); `freqs` (parameter) = the PyPI rotary_embedding_torch the
        # reference imports (vdp.py:19), which pairs features (2i, 2i+1).
        rot_prefix = prefix + "init_temporal_attn.fn.fn.fn.rotary_emb."
        conv = None
        if rot_prefix + "freqs" in state_dict:
            conv = "interleaved"
        elif rot_prefix + "inv_freq" in state_dict:
            conv = "half"
        if conv is not None:
            for mod in self.modules():
                if isinstance(mod, RotaryEmbedding):
                    mod.convention = conv
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward_with_cond_scale(self, *args, cond_scale=2., frame_cond_scale=1., **kwargs):
        # reference: vdp.py:731-754
        if cond_scale == 1 and frame_cond_scale == 1:
            return self.forward(*args, null_cond_prob=0., null_frame_cond_prob=0., **kwargs)
        text_only = self.has_cond and cond_scale != 1 and not (self.has_frame_cross_attn and frame_cond_scale != 1)
        if text_only and not torch.is_grad_enabled() and len(args) == 2 and args[0].is_cuda and kwargs.get("cond") is not None:
            # the conditional and the null-conditioned evaluation as ONE pass over a doubled batch: samples never
            # interact, so each half is what its own pass would give
            x, time = args[0], args[1]
            B = x.shape[0]
            rep = lambda v: None if v is None else torch.cat([v, v], dim=0)
            mask = torch.zeros(2 * B, dtype=torch.uint8, device=x.device)
            mask[B:] = 1
            kw = dict(kwargs)
            kw["cond"] = rep(kwargs["cond"])
            for k in ("frame_cond_features", "focus_present_mask"):
                if kw.get(k) is not None:
                    kw[k] = rep(kw[k])
            both = self.engine().forward(rep(x), rep(time), null_cond_prob=0., null_frame_cond_prob=0.,
                                         null_cond_mask=mask, **kw)
            logits, null_logits = both[:B], both[B:]
            return null_logits + (logits - null_logits) * cond_scale
        logits = self.forward(*args, null_cond_prob=0., null_frame_cond_prob=0., **kwargs)
        if self.has_cond and cond_scale != 1:
            null_logits = self.forward(*args, null_cond_prob=1., null_frame_cond_prob=0., **kwargs)
            logits = null_logits + (logits - null_logits) * cond_scale
        if self.has_frame_cross_attn and frame_cond_scale != 1:
            null_frame_logits = self.forward(*args, null_cond_prob=0., null_frame_cond_prob=1., **kwargs)
            logits = null_frame_logits + (logits - null_frame_logits) * frame_cond_scale
        return logits

    def forward(
        self,
        x,
        time,
        cond=None,
        null_cond_prob=0.,
        frame_cond_features=None,
        null_frame_cond_prob=0.,
        focus_present_mask=None,
        prob_focus_present=0.,
    ):
        # reference: vdp.py:756-831
        assert not (self.has_cond and not exists(cond)), 'cond must be passed in if cond_dim specified'
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if self.train_backend in ("torch-autograd", "hip-resnet", "hip"):
                # training: the same arithmetic as differentiable torch ops (autograd.py), as the reference trains
                from . import autograd
                return autograd.unet_forward(self, x, time, cond=cond, null_cond_prob=null_cond_prob,
                                             frame_cond_features=frame_cond_features,
                                             null_frame_cond_prob=null_frame_cond_prob,
                                             focus_present_mask=focus_present_mask, prob_focus_present=prob_focus_present)
            if x.requires_grad:
                raise NotImplementedError(
                    "the HIP denoiser is forward-only; set unet.train_backend = 'torch-autograd' (the Trainer does) "
                    "to run a differentiable forward through PyTorch ops")
        return self.engine().forward(x, time, cond=cond, null_cond_prob=null_cond_prob,
                                     frame_cond_features=frame_cond_features,
                                     null_frame_cond_prob=null_frame_cond_prob,
                                     focus_present_mask=focus_present_mask, prob_focus_present=prob_focus_present)
