"""ResnetBlock forward AND backward on the HIP kernels, as a torch.autograd.Function.

The reference trains through PyTorch autograd (video_diffusion_pytorch/video_diffusion_pytorch.py:1451,
`loss.backward()` over the torch ops of ResnetBlock, vdp.py:391-452).  With `unet.train_backend = "hip-resnet"` the
differentiable forward of autograd.py hands every ResnetBlock (where most of a training step's arithmetic is: two
(1,3,3) convolutions and a 1x1) to this module; the attention layers, resampling convolutions and embeddings stay
torch ops.  Arithmetic: bf16 activations and activation gradients, fp32 accumulation, fp32 parameter gradients.

  forward   conv (vdmi_tapgemm, PLAIN) -> y1;  vdmi_rmsnorm_silu -> h1;  conv -> y2;  vdmi_rmsnorm_silu (+ residual)
  backward  vdmi_rmsnorm_silu_bwd  (dy, dgamma, dbias, dscale, dshift)
            vdmi_tapgemm over dy with the transposed, tap-reversed weights (vdmi_pack_taps_device)   -> input gradient
            vdmi_rows_transpose + vdmi_conv_wgrad                                                    -> weight gradient

There is no fallback: a shape the kernels do not take raises.
"""
from __future__ import annotations

import ctypes as C
import functools
import os

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import EPI_PLAIN, TapGemmArgs, check

_TAPS_3x3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
_TAPS_1x1 = [(0, 0)]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def to_rows(x: torch.Tensor) -> torch.Tensor:
    """(B, C, F, H, W) float -> channels-last bf16 rows [B*F*H*W][C].  The cast comes first: a tensor that is already
    channels-last in memory (what from_rows returns) then needs no second pass, and one in the standard layout is
    transposed at two bytes per element instead of four."""
    return x.to(torch.bfloat16).permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous()


def from_rows(r: torch.Tensor, b: int, f: int, h: int, w: int, dtype=torch.float32) -> torch.Tensor:
    """bf16 rows -> (B, C, F, H, W) of `dtype`; for bfloat16 a view of the rows themselves (no pass over the data)."""
    return r.view(b, f, h, w, r.shape[1]).permute(0, 4, 1, 2, 3).to(dtype)


def _act_dtype(x: torch.Tensor):
    """The layers keep the activation dtype of their input: a bf16 activation (a view of the previous layer's rows) gives
    a bf16 view of this layer's rows, and between two such layers no cast or copy runs in either direction; anything
    else is answered in float."""
    return torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float32


def _plain(fn):
    """forward / backward of a Function outside autocast, with its inputs left as they come (torch.amp.custom_fwd with
    cast_inputs would widen a bf16 activation again); parameters are float whatever the autocast state."""
    @functools.wraps(fn)
    def inner(*args):
        with torch.autocast("cuda", enabled=False):
            return fn(*args)
    return inner


def torch_ops(fn, *xs):
    """fn(*xs) for a layer that runs as torch ops beside the HIP layers: activations of a bf16 stream are widened first
    (outside autocast the layer's float parameters would not take them), the result goes back to the stream's dtype."""
    if xs[0].dtype != torch.bfloat16:
        return fn(*xs)
    # widened whatever the autocast state: under autocast the layer's matmuls still run in bf16, but its normalisations and
    # softmax then see float inputs, as they do in the all-torch path
    return fn(*(x.float() for x in xs)).to(torch.bfloat16)


class DevPack:
    """A weight set in the tap-GEMM's fragment order, packed on the device (weights change every optimizer step)."""

    def __init__(self, weight: torch.Tensor, transposed: bool, tile_bn=None, in_scale=None, fill=True, gather=None,
                 shape=None, data=None):
        # weight: a Conv3d weight (cout, cin, 1, kh, kw), or the same as [cout][cin][taps].  transposed: the set of the
        # input-gradient pass.  tile_bn: 64 forces 64-channel output tiles, 0 the automatic tile, None picks by width.
        # in_scale: float [cin] factors of the weight's input channels, multiplied in while packing (a folded gamma).
        # gather: int32 [cout][cin][taps] with shape = (cout, cin, taps): the set's source is the re-arrangement
        # weight.flatten()[gather] of a parameter (never built as a tensor).  data: a bf16 slice to pack into.
        cout, cin, ntaps = shape if shape is not None else (weight.shape[0], weight.shape[1], weight[0, 0].numel())
        self.src_shape = (cout, cin, ntaps)
        self.gather = gather
        self.transposed = bool(transposed)
        self.n, self.k = (cin, cout) if transposed else (cout, cin)
        self.ntaps = ntaps
        self.tile_bn = (64 if (self.n >= 256 and self.n % 64 == 0) else 0) if tile_bn is None else tile_bn
        if self.tile_bn == 64:
            self.bn, self.bkc = 64, 64
        else:
            bn, bkc = C.c_int(), C.c_int()
            check(_lib.lib().vdmi_tapgemm_tile(EPI_PLAIN, self.n, C.byref(bn), C.byref(bkc)), "vdmi_tapgemm_tile")
            self.bn, self.bkc = bn.value, bkc.value
        self.npad = (self.n + self.bn - 1) // self.bn * self.bn
        self.elems = _lib.lib().vdmi_pack_taps_elems(self.n, self.npad, self.k, 0, ntaps, self.bkc)
        self.data = data
        self._keep = None
        if fill:
            self.fill(weight, in_scale)

    def job(self, weight: torch.Tensor, in_scale=None) -> "_lib.PackJob":
        """The vdmi_pack_job that fills this set from `weight`; the float32 tensors it reads stay referenced by the set."""
        if self.data is None:
            self.data = torch.empty(self.elems, dtype=torch.bfloat16, device=weight.device)
        cout, cin, ntaps = self.src_shape
        w3 = weight.detach().to(torch.float32).contiguous()
        assert self.gather is not None or w3.numel() == cout * cin * ntaps
        sc = None if in_scale is None else in_scale.detach().reshape(-1).to(torch.float32).contiguous()
        assert sc is None or sc.numel() == cin
        self._keep = (w3, sc)
        return _lib.PackJob(w3.data_ptr(), None if sc is None else sc.data_ptr(),
                            None if self.gather is None else self.gather.data_ptr(), self.data.data_ptr(), self.n,
                            self.npad, self.k, 0, self.ntaps, self.bkc, 1 if self.transposed else 0)

    def fill(self, weight: torch.Tensor, in_scale=None):
        j = self.job(weight, in_scale)
        if in_scale is None and self.gather is None:
            check(_lib.lib().vdmi_pack_taps_device(j.w, j.n, j.npad, j.cin0, j.cin1, j.ntaps, j.bkc, j.transposed, j.out,
                                                   _stream()), "vdmi_pack_taps_device")
        else:
            pack_batch([j], weight.device)


def pack_batch(jobs, device):
    """All of `jobs` (a list of _lib.PackJob) in one launch."""
    lib = _lib.lib()
    arr = (_lib.PackJob * len(jobs))(*jobs)
    table = torch.empty(lib.vdmi_pack_taps_batch_table_bytes(len(jobs)), device=device, dtype=torch.uint8)
    check(lib.vdmi_pack_taps_batch(C.cast(arr, C.c_void_p), len(jobs), table.data_ptr(), _stream()), "vdmi_pack_taps_batch")
    return table


class PhasePacks:
    """Several weight sets re-arranged from ONE parameter (the four output parities of the transposed convolution), back to
    back in one buffer: `data`, sets `stride` elements apart."""

    def __init__(self, weight: torch.Tensor, gathers, shape, tile_bn):
        self.packs = [DevPack(weight, False, tile_bn, fill=False, gather=g, shape=shape) for g in gathers]
        self.stride = self.packs[0].elems
        self.data = torch.empty(self.stride * len(gathers), dtype=torch.bfloat16, device=weight.device)
        for i, p in enumerate(self.packs):
            p.data = self.data[i * self.stride:(i + 1) * self.stride]

    def jobs(self, weight):
        return [p.job(weight) for p in self.packs]


class PackStore:
    """Packed weight sets that outlive a forward / backward, for a caller that knows when the weights change (the Trainer
    around its optimizer step, a sampler whose weights are frozen).  A set is found again by the weight's memory and
    shape and trusted while the version counters of the weight (and of its scale) stand where they stood when it was
    packed.  In-place updates under no_grad (load_state_dict, the single-tensor and foreach optimizers) move them; a
    FUSED optimizer step (torch._fused_adam_) and writes through `.data` do not - whoever runs those calls invalidate()
    afterwards, as the Trainer does after every optimizer step.  refresh() repacks every set that went stale in ONE
    launch (a training step otherwise spends ~200 launches of a few microseconds of work each on it)."""

    class _Entry:
        __slots__ = ("pack", "weight", "scale", "wv", "sv", "age")

    def __init__(self):
        self.entries = {}
        self._table = None

    def __len__(self):
        return len(self.entries)

    def invalidate(self):
        for e in self.entries.values():
            e.wv = -1

    @staticmethod
    def _stale(e):
        return e.wv != e.weight._version or (e.scale is not None and e.sv != e.scale._version)

    @staticmethod
    def _stamp(e):
        e.wv = e.weight._version
        e.sv = None if e.scale is None else e.scale._version

    @staticmethod
    def _jobs(e):
        return e.pack.jobs(e.weight) if isinstance(e.pack, PhasePacks) else [e.pack.job(e.weight, e.scale)]

    def _entry(self, key, weight, in_scale, make):
        e = self.entries.get(key)
        if e is None:
            e = PackStore._Entry()
            e.pack = make()
            # aliases of the caller's tensors: they share the version counters and keep the memory (and with it the key) theirs
            e.weight, e.scale = weight.detach(), None if in_scale is None else in_scale.detach()
            e.wv = -1
            self.entries[key] = e
        e.age = 0
        if self._stale(e):
            jobs = self._jobs(e)
            if len(jobs) == 1 and e.scale is None and e.pack.gather is None:
                e.pack.fill(e.weight)
            else:
                pack_batch(jobs, e.weight.device)
            self._stamp(e)
        return e.pack

    def get(self, weight: torch.Tensor, transposed: bool, tile_bn=None, in_scale=None, gather=None, shape=None) -> DevPack:
        key = (weight.data_ptr(), tuple(weight.shape), bool(transposed), tile_bn,
               None if in_scale is None else in_scale.data_ptr(), None if gather is None else gather.data_ptr())
        return self._entry(key, weight, in_scale,
                           lambda: DevPack(weight, transposed, tile_bn, fill=False, gather=gather, shape=shape))

    def get_phases(self, weight: torch.Tensor, gathers, shape, tile_bn) -> PhasePacks:
        key = (weight.data_ptr(), tuple(weight.shape), "phases", tile_bn, None, gathers[0].data_ptr())
        return self._entry(key, weight, None, lambda: PhasePacks(weight, gathers, shape, tile_bn))

    def refresh(self):
        """Repack every stale set in one launch; sets no caller asked for since the refresh before the last are dropped."""
        for k in [k for k, e in self.entries.items() if e.age >= 2]:
            del self.entries[k]
        stale = [e for e in self.entries.values() if self._stale(e)]
        for e in self.entries.values():
            e.age += 1
        if not stale:
            return
        # the forward and the input-gradient set of a weight next to each other: the second reads the weight from cache
        stale.sort(key=lambda e: e.weight.data_ptr())
        self._table = pack_batch([j for e in stale for j in self._jobs(e)], stale[0].weight.device)
        for e in stale:
            self._stamp(e)


_PACKS = None        # the PackStore of the enclosing fixed_weights block


class fixed_weights:
    """`with fixed_weights(store):` - the caller guarantees that the parameters change only through version-counted in-place
    updates between the forwards and backwards run inside this and later blocks with the same `store` (a PackStore it
    owns): the packed weight sets of the HIP layers are then kept in the store, and entering the block repacks the stale
    ones in one launch.  The Trainer wraps its micro-steps in it, the guided sampler its loop (weights frozen)."""

    def __init__(self, store: PackStore):
        self.store = store

    def __enter__(self):
        global _PACKS
        self.prev, _PACKS = _PACKS, self.store
        self.store.refresh()
        return self

    def __exit__(self, *exc):
        global _PACKS
        _PACKS = self.prev
        return False


def param_pack(weight: torch.Tensor, transposed: bool, tile_bn=None, in_scale=None, gather=None, shape=None) -> DevPack:
    """DevPack of a parameter, through the store of an enclosing fixed_weights block when there is one."""
    if _PACKS is None:
        return DevPack(weight, transposed, tile_bn, in_scale, gather=gather, shape=shape)
    return _PACKS.get(weight, transposed, tile_bn, in_scale, gather, shape)


def phase_packs(weight: torch.Tensor, gathers, shape, tile_bn) -> PhasePacks:
    if _PACKS is not None:
        return _PACKS.get_phases(weight, gathers, shape, tile_bn)
    ph = PhasePacks(weight, gathers, shape, tile_bn)
    pack_batch(ph.jobs(weight), weight.device)
    return ph


def conv_rows(src: torch.Tensor, pack: DevPack, bias, frames: int, fpb: int, h: int, w: int, residual=None) -> torch.Tensor:
    """Stride-1 'same' convolution of channels-last rows through vdmi_tapgemm (PLAIN epilogue, + bias, + residual rows)."""
    assert src.shape[1] == pack.k and src.dtype == torch.bfloat16 and src.is_contiguous()
    out = torch.empty(frames * h * w, pack.n, device=src.device, dtype=torch.bfloat16)
    a = TapGemmArgs()
    a.frames, a.frames_per_batch = frames, fpb
    a.grid_h, a.grid_w, a.out_h, a.out_w = h, w, h, w
    a.out_mul = 1
    a.out, a.ldo, a.n, a.epilogue = out.data_ptr(), pack.n, pack.n, EPI_PLAIN
    taps = _TAPS_3x3 if pack.ntaps == 9 else _TAPS_1x1
    p = a.main
    p.src0, p.c0, p.c1 = src.data_ptr(), pack.k, 0
    p.in_h, p.in_w, p.stride, p.ntaps = h, w, 1, len(taps)
    for i, (dy, dx) in enumerate(taps):
        p.dy[i], p.dx[i] = dy, dx
    p.w = pack.data.data_ptr()
    a.bias = None if bias is None else bias.data_ptr()
    a.tile_bn = pack.tile_bn
    if residual is not None:
        a.has_res, a.res_identity, a.ld_res = 2, residual.data_ptr(), residual.shape[1]
    check(_lib.lib().vdmi_tapgemm(C.byref(a), _stream()), "vdmi_tapgemm")
    return out


def frame_conv_rows(src, pack: "DevPack", bias, b, f, hw, residual=None):
    """Conv3d(cin, n, (k,1,1), padding (k/2,0,0)) over rows (+ bias, + residual rows): each sample is an f x hw 'image'
    whose k taps step along its first axis."""
    k = pack.ntaps
    out = torch.empty(b * f * hw, pack.n, device=src.device, dtype=torch.bfloat16)
    a = TapGemmArgs()
    a.frames, a.frames_per_batch = b, 1
    a.grid_h, a.grid_w, a.out_h, a.out_w = f, hw, f, hw
    a.out_mul = 1
    a.out, a.ldo, a.n, a.epilogue = out.data_ptr(), pack.n, pack.n, EPI_PLAIN
    p = a.main
    p.src0, p.c0, p.c1 = src.data_ptr(), pack.k, 0
    p.in_h, p.in_w, p.stride, p.ntaps = f, hw, 1, k
    for i in range(k):
        p.dy[i], p.dx[i] = i - k // 2, 0
    p.w = pack.data.data_ptr()
    a.bias = None if bias is None else bias.data_ptr()
    a.tile_bn = pack.tile_bn
    if residual is not None:
        a.has_res, a.res_identity, a.ld_res = 2, residual.data_ptr(), residual.shape[1]
    check(_lib.lib().vdmi_tapgemm(C.byref(a), _stream()), "vdmi_tapgemm")
    return out


def frame_conv_wgrad(xr, dy, b, f, hw, k):
    """float32 (cout, cin, k, 1, 1) weight gradient of that convolution: tap j pairs dy[b, f'] with x[b, f' + j - k/2], so
    each tap is a row GEMM over a frame-shifted, zero-filled copy of x."""
    c, cout = xr.shape[1], dy.shape[1]
    x5 = xr.view(b, f, hw, c)
    taps = []
    for j in range(k):
        s = j - k // 2
        xs = torch.zeros_like(x5)
        lo, hi = max(0, -s), min(f, f - s)
        if hi > lo:
            xs[:, lo:hi] = x5[:, lo + s:hi + s]
        taps.append(rows_wgrad(xs.view(-1, c), dy))
    return torch.stack(taps, dim=-1).reshape(cout, c, k, 1, 1)


def rmsnorm_silu_rows(y, gain, shift, bstride, rows_per_batch, residual=None):
    rows, c = y.shape
    out = torch.empty_like(y)
    check(_lib.lib().vdmi_rmsnorm_silu(
        y.data_ptr(), c, out.data_ptr(), c, rows, c, gain.data_ptr(), bstride,
        None if shift is None else shift.data_ptr(), bstride, rows_per_batch, float(c) ** 0.5,
        None if residual is None else residual.data_ptr(), 0 if residual is None else residual.shape[1], _stream()),
        "vdmi_rmsnorm_silu")
    return out


def rows_transposed(rows: torch.Tensor, nf: int, hw: int) -> torch.Tensor:
    c = rows.shape[1]
    lib = _lib.lib()
    out = torch.empty(lib.vdmi_rows_transpose_elems(c, nf, hw), device=rows.device, dtype=torch.bfloat16)
    check(lib.vdmi_rows_transpose(rows.data_ptr(), c, 0, c, nf, hw, out.data_ptr(), _stream()), "vdmi_rows_transpose")
    return out


def conv_wgrad(x_t: torch.Tensor, dy_t: torch.Tensor, cin: int, cout: int, nf: int, h: int, w: int, ntaps: int,
               tap=(0, 0)) -> torch.Tensor:
    """float32 (cout, cin, 1, k, k) weight gradient from the transposed operands; `tap`: the pixel offset of the
    one-tap form (dw[co][ci] = sum_p dy[p][co] x[p + tap][ci])."""
    k = 3 if ntaps == 9 else 1
    dw = torch.empty(cout, cin, 1, k, k, device=x_t.device, dtype=torch.float32)
    taps = _TAPS_3x3 if ntaps == 9 else [tuple(tap)]
    dy = (C.c_int8 * ntaps)(*[t[0] for t in taps])
    dx = (C.c_int8 * ntaps)(*[t[1] for t in taps])
    lib = _lib.lib()
    ws = torch.empty(lib.vdmi_conv_wgrad_workspace_bytes(cin, cout, nf, h, w, ntaps), device=x_t.device, dtype=torch.uint8)
    check(lib.vdmi_conv_wgrad(x_t.data_ptr(), dy_t.data_ptr(), dw.data_ptr(), cin, cout, cin, 0, nf, h, w, ntaps,
                              dy, dx, ws.data_ptr(), _stream()), "vdmi_conv_wgrad")
    return dw


# VDMI_WGRAD3_ROWS=0: the (1,3,3) weight gradient from transposed copies of its operands (A/B switch; the default
# contracts over the pixels of the channels-last rows as they lie)
_WGRAD3_ROWS = os.environ.get("VDMI_WGRAD3_ROWS", "1") != "0"


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, nf: int, h: int, w: int) -> torch.Tensor:
    """float32 (cout, cin, 1, 3, 3) weight gradient of a (1,3,3) "same" convolution from its input rows x [nf*h*w][cin] and
    output-gradient rows dy [nf*h*w][cout] (bf16, channels-last)."""
    cin, cout = x.shape[1], dy.shape[1]
    if not _WGRAD3_ROWS:
        return conv_wgrad(rows_transposed(x, nf, h * w), rows_transposed(dy, nf, h * w), cin, cout, nf, h, w, 9)
    assert x.shape[0] == dy.shape[0] == nf * h * w and x.is_contiguous() and dy.is_contiguous()
    lib = _lib.lib()
    dw = torch.empty(cout, cin, 1, 3, 3, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.vdmi_conv_wgrad_rows_workspace_bytes(cin, cout, nf, h, w), device=x.device, dtype=torch.uint8)
    check(lib.vdmi_conv_wgrad_rows(x.data_ptr(), cin, dy.data_ptr(), cout, dw.data_ptr(), cin, cout, cin, 0, nf, h, w,
                                   ws.data_ptr(), _stream()), "vdmi_conv_wgrad_rows")
    return dw


def rows_wgrad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """float32 [cout][cin] weight gradient of a 1x1 convolution / Linear from its input rows x [rows][cin] and output-gradient
    rows dy [rows][cout] (bf16, channels-last) - no transposed copies."""
    rows, cin = x.shape
    cout = dy.shape[1]
    assert dy.shape[0] == rows and x.is_contiguous() and dy.is_contiguous()
    lib = _lib.lib()
    dw = torch.empty(cout, cin, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.vdmi_rows_wgrad_workspace_bytes(cin, cout, rows), device=x.device, dtype=torch.uint8)
    check(lib.vdmi_rows_wgrad(x.data_ptr(), cin, dy.data_ptr(), cout, dw.data_ptr(), cin, cout, cin, 0, rows, ws.data_ptr(),
                              _stream()), "vdmi_rows_wgrad")
    return dw


def conv_wgrad_gather(x_t, dy_t, cin, cout, nf, h, w, kh, kw) -> torch.Tensor:
    """float32 (cout, cin, 1, kh, kw) weight gradient of a small-cin convolution (init_conv)."""
    lib = _lib.lib()
    dw = torch.empty(cout, cin, 1, kh, kw, device=x_t.device, dtype=torch.float32)
    ws = torch.empty(lib.vdmi_conv_wgrad_gather_workspace_bytes(cin, cout, nf, h, w, kh, kw), device=x_t.device,
                     dtype=torch.uint8)
    check(lib.vdmi_conv_wgrad_gather(x_t.data_ptr(), dy_t.data_ptr(), dw.data_ptr(), cin, cout, nf, h, w, kh, kw,
                                     ws.data_ptr(), _stream()), "vdmi_conv_wgrad_gather")
    return dw


# VDMI_NORM_BWD_ATOMIC=1: the gradient sums of Block's norm through float atomics (A/B switch; the default adds
# per-workgroup partials in a fixed order)
_NORM_BWD_FIXED_ORDER = os.environ.get("VDMI_NORM_BWD_ATOMIC", "0") != "1"


def rmsnorm_silu_bwd_rows(y, dh, gamma, scale, shift, batch, rows_per_batch, want_bias=True):
    rows, c = y.shape
    dev = y.device
    dy = torch.empty_like(y)
    nb = batch if scale is not None else 0
    lib = _lib.lib()
    ptr = lambda t: None if t is None else t.data_ptr()
    args = lambda: (y.data_ptr(), c, dh.data_ptr(), dh.shape[1], dy.data_ptr(), c, batch, rows_per_batch, c,
                    gamma.data_ptr(), ptr(scale), ptr(shift), c, float(c) ** 0.5, dgamma.data_ptr(), ptr(dbias),
                    ptr(dscale), ptr(dshift))
    ws_bytes = lib.vdmi_rmsnorm_silu_bwd_workspace_bytes(batch, rows_per_batch, c) if _NORM_BWD_FIXED_ORDER else 0
    # fixed-order form: the sums are written, from per-workgroup partials in a workspace; the other form accumulates
    # into all four with float atomics: one zero fill, four views
    acc = (torch.empty if ws_bytes else torch.zeros)(2 + 2 * nb, c, device=dev, dtype=torch.float32)
    dgamma = acc[0]
    dbias = acc[1] if want_bias else None
    dscale = acc[2:2 + nb] if scale is not None else None
    dshift = acc[2 + nb:] if scale is not None else None
    if ws_bytes:
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        check(lib.vdmi_rmsnorm_silu_bwd_ws(*args(), ws.data_ptr(), _stream()), "vdmi_rmsnorm_silu_bwd_ws")
    else:
        check(lib.vdmi_rmsnorm_silu_bwd(*args(), _stream()), "vdmi_rmsnorm_silu_bwd")
    return dy, dgamma, dbias, dscale, dshift


def colsum_rows(x: torch.Tensor) -> torch.Tensor:
    """float32 [c] column sums of bf16 rows (a bias gradient); fixed-order partial sums unless VDMI_NORM_BWD_ATOMIC=1."""
    lib = _lib.lib()
    rows, c = x.shape
    if _NORM_BWD_FIXED_ORDER:
        out = torch.empty(c, device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.vdmi_colsum_workspace_bytes(rows, c), device=x.device, dtype=torch.uint8)
        check(lib.vdmi_colsum_ws(x.data_ptr(), c, rows, c, out.data_ptr(), ws.data_ptr(), _stream()), "vdmi_colsum_ws")
        return out
    out = torch.zeros(c, device=x.device, dtype=torch.float32)
    check(lib.vdmi_colsum(x.data_ptr(), c, rows, c, out.data_ptr(), _stream()), "vdmi_colsum")
    return out


class ResnetBlockFn(torch.autograd.Function):
    """ResnetBlock.forward (vdp.py:442-452) with its backward; x float (B, Cin, F, H, W) -> float (B, Cout, F, H, W).
    scale / shift: float [B][Cout] (the chunks of the block's time MLP output) or None; wres / bres None = identity."""

    @staticmethod
    @_plain
    def forward(ctx, x, scale, shift, w1, b1, g1, w2, b2, g2, wres, bres, with_res=True):
        # with_res = False: block2(block1(x)) alone, for callers that put a layer between the blocks and the residual
        if not x.is_cuda:
            raise _lib.VdmiError("the hip-resnet training backend needs CUDA (HIP) tensors; there is no CPU path")
        b, cin, f, h, w = x.shape
        cout = w1.shape[0]
        frames, hw = b * f, h * w
        xr = to_rows(x)
        g1v, g2v = g1.detach().reshape(-1).float().contiguous(), g2.detach().reshape(-1).float().contiguous()
        if scale is not None:
            scale_c, shift_c = scale.detach().float().contiguous(), shift.detach().float().contiguous()
            gain1, bstride = (g1v[None, :] * (scale_c + 1)).contiguous(), cout
        else:
            scale_c = shift_c = None
            gain1, bstride = g1v, 0
        y1 = conv_rows(xr, param_pack(w1, False), b1.detach().float().contiguous(), frames, f, h, w)
        h1 = rmsnorm_silu_rows(y1, gain1, shift_c, bstride, f * hw)
        y2 = conv_rows(h1, param_pack(w2, False), b2.detach().float().contiguous(), frames, f, h, w)
        if not with_res:
            wres = None
            out = rmsnorm_silu_rows(y2, g2v, None, 0, f * hw)
        elif wres is None:
            out = rmsnorm_silu_rows(y2, g2v, None, 0, f * hw, residual=xr)
        else:
            h2 = rmsnorm_silu_rows(y2, g2v, None, 0, f * hw)
            out = conv_rows(xr, param_pack(wres, False), bres.detach().float().contiguous(), frames, f, h, w, residual=h2)
        ctx.save_for_backward(xr, y1, h1, y2, g1v, g2v, scale_c, shift_c, w1, w2, wres)
        ctx.dims = (b, cin, cout, f, h, w)
        ctx.with_res = with_res
        ctx.act = _act_dtype(x)
        return from_rows(out, b, f, h, w, ctx.act)

    @staticmethod
    @_plain
    def backward(ctx, dout):
        xr, y1, h1, y2, g1v, g2v, scale_c, shift_c, w1, w2, wres = ctx.saved_tensors
        b, cin, cout, f, h, w = ctx.dims
        frames, hw = b * f, h * w
        do = to_rows(dout)
        # block2: SiLU / RMSNorm, then the conv
        # weight gradients only where asked for (a gradient with respect to the input alone - guidance at sampling time,
        # frozen weights - needs neither the transposes nor the contraction over pixels)
        need = ctx.needs_input_grad
        dy2, dg2, db2, _, _ = rmsnorm_silu_bwd_rows(y2, do, g2v, None, None, b, f * hw)
        dw2 = None
        if need[6]:
            dw2 = conv3x3_wgrad(h1, dy2, frames, h, w)
        dh1 = conv_rows(dy2, param_pack(w2, True), None, frames, f, h, w)
        # block1
        dy1, dg1, db1, dscale, dshift = rmsnorm_silu_bwd_rows(y1, dh1, g1v, scale_c, shift_c, b, f * hw)
        dw1 = conv3x3_wgrad(xr, dy1, frames, h, w) if need[3] else None
        # residual branch, then the input gradient: dgrad of conv1 + (res_conv's dgrad | the identity)
        if not ctx.with_res:
            dwres = dbres = dx_res = None
        elif wres is None:
            dwres = dbres = None
            dx_res = do
        else:
            dwres = rows_wgrad(xr, do).view(cout, cin, 1, 1, 1) if need[9] else None
            dbres = colsum_rows(do) if need[10] else None
            dx_res = conv_rows(do, param_pack(wres, True), None, frames, f, h, w)
        dx = conv_rows(dy1, param_pack(w1, True), None, frames, f, h, w, residual=dx_res)
        return (from_rows(dx, b, f, h, w, ctx.act), dscale, dshift, dw1, db1, dg1.view(cout, 1, 1, 1), dw2, db2,
                dg2.view(cout, 1, 1, 1), dwres, dbres, None)


class InitConvFn(torch.autograd.Function):
    """Unet3D.init_conv (vdp.py:644, 772).  Forward on vdmi_init_conv (the sampling plan's kernel: float channels-first
    in, bf16 rows out, products in split-bf16 = float precision); the weight gradient on vdmi_conv_wgrad_gather (the
    library convolutions torch falls back to for this 3-channel (1,7,7) layer take 20 ms forward and 40 ms for the weight
    gradient at 16 x 128 x 128).  `act`: dtype of the result (bfloat16: a view of the output rows)."""

    @staticmethod
    @_plain
    def forward(ctx, x, weight, bias, pad, act):
        ctx.save_for_backward(x, weight)
        ctx.pad = pad
        b, cin, f, h, w = x.shape
        cout, k = weight.shape[0], weight.shape[3]
        if not (x.is_cuda and cout % 16 == 0 and k == 2 * pad + 1):
            return F.conv3d(x.float(), weight, bias, padding=(0, pad, pad)).to(act)
        rows = torch.empty(b * f * h * w, cout, device=x.device, dtype=torch.bfloat16)
        w2 = weight.detach()[:, :, 0].float().contiguous()
        bz = bias.detach().float().contiguous() if bias is not None else torch.zeros(cout, device=x.device)
        check(_lib.lib().vdmi_init_conv(x.detach().float().contiguous().data_ptr(), w2.data_ptr(), bz.data_ptr(),
                                        rows.data_ptr(), b, cin, f, h, w, cout, k, _stream()), "vdmi_init_conv")
        return from_rows(rows, b, f, h, w, act)

    @staticmethod
    @_plain
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors
        b, cin, f, h, w = x.shape
        cout, kh, kw = weight.shape[0], weight.shape[3], weight.shape[4]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.nn.grad.conv3d_input(x.shape, weight, dout.float(), padding=(0, ctx.pad, ctx.pad)).to(x.dtype)
        cpad = (cin + 7) // 8 * 8                      # rows need a multiple of 8 channels
        xr = torch.zeros(b * f * h * w, cpad, device=x.device, dtype=torch.bfloat16)
        xr[:, :cin] = x.permute(0, 2, 3, 4, 1).reshape(-1, cin)
        # the transposed layout is channel-major: the padding channels are simply never addressed
        do_rows = to_rows(dout)
        dw = conv_wgrad_gather(rows_transposed(xr, b * f, h * w), rows_transposed(do_rows, b * f, h * w),
                               cin, cout, b * f, h, w, kh, kw)
        return dx, dw, dout.float().sum(dim=(0, 2, 3, 4)), None, None


def init_conv(conv: torch.nn.Conv3d, x: torch.Tensor, act=torch.float32) -> torch.Tensor:
    kh, kw = conv.kernel_size[1], conv.kernel_size[2]
    assert conv.kernel_size[0] == 1 and kh == kw and conv.padding == (0, kh // 2, kw // 2)
    return InitConvFn.apply(x, conv.weight, conv.bias, kh // 2, act)


class PointwiseConvFn(torch.autograd.Function):
    """A 1x1x1 Conv3d with few output channels (final_conv[1], vdp.py:728: dim -> 3): a channel matmul forward and
    towards the input, weight gradient on vdmi_conv_wgrad (the library kernel torch picks for this weight gradient under
    bf16 autocast takes 40 ms at 16 x 128 x 128)."""

    @staticmethod
    @_plain
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        w2 = weight.reshape(weight.shape[0], weight.shape[1])
        b, cin, f, h, w = x.shape
        cout = w2.shape[0]
        if x.is_cuda and x.dtype == torch.bfloat16 and cout <= 8 and cin % 8 == 0:
            # a bf16 activation stream ("hip" backend): the sampling plan's kernel, bf16 rows in (a view of the previous
            # layer's output), float channels-first out; a float input keeps the float matmul (nothing is rounded)
            out = torch.empty(b, cout, f, h, w, device=x.device, dtype=torch.float32)
            check(_lib.lib().vdmi_final_conv(to_rows(x).data_ptr(), w2.detach().float().contiguous().data_ptr(),
                                             bias.detach().float().contiguous().data_ptr(), out.data_ptr(), b, cin, f, h, w,
                                             cout, _stream()), "vdmi_final_conv")
            return out
        return torch.einsum("oc,bcfhw->bofhw", w2, x.float()) + bias.view(1, -1, 1, 1, 1)

    @staticmethod
    @_plain
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors
        b, cin, f, h, w = x.shape
        cout = weight.shape[0]
        dout = dout.float()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.einsum("oc,bofhw->bcfhw", weight.reshape(cout, cin), dout).to(x.dtype)
        cpad = (cout + 7) // 8 * 8
        dr = torch.zeros(b * f * h * w, cpad, device=x.device, dtype=torch.bfloat16)
        dr[:, :cout] = dout.permute(0, 2, 3, 4, 1).reshape(-1, cout)
        dw = rows_wgrad(to_rows(x), dr)[:cout].reshape(cout, cin, 1, 1, 1).contiguous()
        return dx, dw, dout.sum(dim=(0, 2, 3, 4))


def pointwise_conv(conv: torch.nn.Conv3d, x: torch.Tensor) -> torch.Tensor:
    assert conv.kernel_size == (1, 1, 1) and conv.stride == (1, 1, 1) and conv.bias is not None
    return PointwiseConvFn.apply(x, conv.weight, conv.bias)


class TemporalConvResFn(torch.autograd.Function):
    """The tail of the version-1 ResnetBlock (version1diffusion.py:308, 320-323): temporal_conv(h) + res_conv(x), a
    (k,1,1) convolution over frames plus the block's residual branch (1x1 convolution or identity)."""

    @staticmethod
    @_plain
    def forward(ctx, h, x, wt, bt, wres, bres):
        b, c, f, hh, ww = h.shape
        hw = hh * ww
        hr, xr = to_rows(h), to_rows(x)
        fz = lambda t: None if t is None else t.detach().float().contiguous()
        r = xr if wres is None else conv_rows(xr, param_pack(wres, False), fz(bres), b * f, f, hh, ww)
        out = frame_conv_rows(hr, param_pack(wt, False), fz(bt), b, f, hw, residual=r)
        ctx.save_for_backward(hr, xr, wt, wres)
        ctx.meta = (b, c, x.shape[1], f, hh, ww, wt.shape[2])
        ctx.acts = (_act_dtype(h), _act_dtype(x))
        return from_rows(out, b, f, hh, ww, ctx.acts[0])

    @staticmethod
    @_plain
    def backward(ctx, dout):
        hr, xr, wt, wres = ctx.saved_tensors
        b, c, cin, f, hh, ww, k = ctx.meta
        need = ctx.needs_input_grad
        hw, frames = hh * ww, b * f
        do = to_rows(dout)
        dh = from_rows(frame_conv_rows(do, param_pack(wt, True), None, b, f, hw), b, f, hh, ww, ctx.acts[0]) if need[0] else None
        dwt = frame_conv_wgrad(hr, do, b, f, hw, k) if need[2] else None
        dbt = colsum_rows(do) if need[3] else None
        dwres = dbres = None
        if wres is None:
            dx = from_rows(do, b, f, hh, ww, ctx.acts[1]) if need[1] else None
        else:
            dx = from_rows(conv_rows(do, param_pack(wres, True), None, frames, f, hh, ww), b, f, hh, ww, ctx.acts[1]) if need[1] else None
            if need[4]:
                dwres = rows_wgrad(xr, do).view(c, cin, 1, 1, 1)
            if need[5]:
                dbres = colsum_rows(do)
        return dh, dx, dwt, dbt, dwres, dbres


# VDMI_TIME_TABLES=0: every ResnetBlock runs its own time MLP (A/B switch)
_TIME_TABLES_ON = os.environ.get("VDMI_TIME_TABLES", "1") != "0"
_TIME_TABLES = None     # {id(ResnetBlock): its [B][2 Cout] slice} for the forward in progress, or None


class time_tables:
    """`with time_tables(unet, temb):` - the time MLPs of all ResnetBlocks of `unet` (ResnetBlock.mlp = SiLU + Linear,
    vdp.py:432-436, 445-447) as ONE Linear over the concatenated weights; resnet_block() then takes its block's slice.
    The blocks' small matmuls, their SiLUs and, in the backward, the per-block products and the additions into the
    embedding's gradient (~9 launches per block) become a handful of launches for the model; under autocast the
    weights are narrowed once.  Same function of the same parameters: autograd splits the gradient back."""

    def __init__(self, unet, temb):
        self.tables = None
        if not _TIME_TABLES_ON or temb is None:
            return
        blocks = [m for m in unet.modules() if getattr(m, "mlp", None) is not None and hasattr(m, "block1")
                  and isinstance(m.mlp, torch.nn.Sequential) and isinstance(m.mlp[1], torch.nn.Linear)
                  and m.mlp[1].in_features == temb.shape[1]]
        if len(blocks) < 2:
            return
        w = torch.cat([m.mlp[1].weight for m in blocks], dim=0)
        b = torch.cat([m.mlp[1].bias for m in blocks], dim=0)
        e = F.linear(F.silu(temb), w, b)
        parts = e.split([m.mlp[1].out_features for m in blocks], dim=1)
        self.tables = {id(m): p for m, p in zip(blocks, parts)}

    def __enter__(self):
        global _TIME_TABLES
        self.prev, _TIME_TABLES = _TIME_TABLES, self.tables
        return self

    def __exit__(self, *exc):
        global _TIME_TABLES
        _TIME_TABLES = self.prev
        return False


def resnet_block(rb, x: torch.Tensor, temb) -> torch.Tensor:
    """autograd.py's `_resnet` on the HIP kernels.  The variant's block (version1diffusion.py:308, 320-323: a (3,1,1)
    convolution over frames between block2 and the residual) runs its two Blocks here and the rest as torch ops."""
    scale = shift = None
    if rb.mlp is not None:
        e = _TIME_TABLES.get(id(rb)) if _TIME_TABLES is not None else None
        if e is None:
            e = rb.mlp[1](F.silu(temb))
        scale, shift = e.chunk(2, dim=1)
    rc = rb.res_conv
    wres, bres = (rc.weight, rc.bias) if isinstance(rc, torch.nn.Conv3d) else (None, None)
    args = (x, scale, shift, rb.block1.proj.weight, rb.block1.proj.bias, rb.block1.norm.gamma,
            rb.block2.proj.weight, rb.block2.proj.bias, rb.block2.norm.gamma)
    tconv = getattr(rb, "temporal_conv", None)
    if tconv is not None:
        h = ResnetBlockFn.apply(*args, None, None, False)
        if tconv.kernel_size[1:] == (1, 1) and tconv.kernel_size[0] % 2 == 1 and tconv.padding == (tconv.kernel_size[0] // 2, 0, 0):
            return TemporalConvResFn.apply(h, x, tconv.weight, tconv.bias, wres, bres)
        return torch_ops(lambda y, xin: tconv(y) + rc(xin), h, x)
    return ResnetBlockFn.apply(*args, wres, bres)


def supports(unet) -> bool:
    """True when every ResnetBlock, attention layer and resampling convolution of `unet` has a shape the training kernels
    take (channel widths powers of two from 64 to 2048, heads of 32 features): what `train_backend = "hip"` needs."""
    if _lib.PRECISION != "bf16":     # the training layers exchange bfloat16 rows; the IEEE-half build serves sampling only
        return False
    ok = lambda c: 64 <= c <= 2048 and (c & (c - 1)) == 0
    try:
        return _supports(unet, ok)
    except AttributeError:          # a module tree that is not the reference's Unet3D layout
        return False


def _supports(unet, ok) -> bool:
    for mod in unet.modules():
        name = type(mod).__name__
        if name == "ResnetBlock":
            if not (ok(mod.block1.proj.out_channels) and mod.block1.proj.in_channels % 8 == 0):
                return False
        elif name in ("Attention", "SpatialLinearAttention", "SpatioTemporalAttention"):
            if getattr(mod, "dim_head", 32) != 32:
                return False
            rot = getattr(mod, "rotary_emb", None)
            if rot is not None and getattr(rot, "convention", "half") != "half":
                return False            # the attention backward kernel rotates half-split pairs only
        elif name == "PreNorm":
            if not ok(mod.norm.gamma.numel()):
                return False
        elif isinstance(mod, (torch.nn.ConvTranspose3d, torch.nn.Conv3d)) and mod.kernel_size == (1, 4, 4):
            if mod.in_channels % 64:
                return False
    return True


# ---------------------------------------------------------------------------------------------------------------------
# temporal attention layer: Residual(PreNorm(EinopsToAndFrom('b c f h w', 'b (h w) f c', Attention)))
# ---------------------------------------------------------------------------------------------------------------------
def _linear_pack(w2: torch.Tensor, transposed: bool, in_scale=None) -> DevPack:
    """A Linear weight [n][k] (or a 1x1x1 convolution's) as a one-tap weight set; in_scale: factors of its k input channels."""
    # the library's own tile choice (128 output channels per workgroup where the width allows) instead of the 64-channel
    # tiles the 3x3 convolutions of the wide levels take: a row GEMM is bound by its rows' traffic, and every channel
    # tile of a workgroup column reads the input rows again (measured over the training step: 31.1 -> 30.6 ms;
    # VDMI_LINEAR_TILE_BELOW=0 brings the 64-channel tiles back for an A/B)
    thr = int(os.environ.get("VDMI_LINEAR_TILE_BELOW", str(1 << 30)))
    tile = 0 if min(w2.shape[0], w2.shape[1]) <= thr else None
    return param_pack(w2.reshape(w2.shape[0], w2.shape[1], 1, 1, 1), transposed, tile, in_scale)


class TemporalAttnFn(torch.autograd.Function):
    """vdp.py:380-389 (LayerNorm), 531-612 (Attention with rotary and relative-position bias), 331-337 (Residual) over
    x float (B, C, F, H, W): forward and backward on the HIP kernels.  gamma (1, C, 1, 1, 1); wqkv (3*heads*32, C);
    wout (C, heads*32); bias float [heads][F][F] (may require grad); cos / sin float [F][16] or None."""

    @staticmethod
    @_plain
    def forward(ctx, x, gamma, wqkv, wout, bias, cos, sin, heads, scale, eps, spatial=False, focus=None):
        if not x.is_cuda:
            raise _lib.VdmiError("the HIP training backend needs CUDA (HIP) tensors; there is no CPU path")
        lib = _lib.lib()
        b, c, f, h, w = x.shape
        hw, rows, hid = h * w, b * f * h * w, heads * 32
        assert wqkv.shape == (3 * hid, c) and wout.shape == (c, hid)
        xr = to_rows(x)
        xh = torch.empty_like(xr)
        check(lib.vdmi_layernorm_rows(xr.data_ptr(), xh.data_ptr(), rows, c, eps, _stream()), "vdmi_layernorm_rows")
        # LayerNorm gamma folded into to_qkv while its weight set is packed
        qkv = conv_rows(xh, _linear_pack(wqkv.detach(), False, gamma), None, b * f, f, h, w)
        o = torch.empty(rows, hid, device=x.device, dtype=torch.bfloat16)
        lse = torch.empty(rows, heads, device=x.device, dtype=torch.float32)
        bias_c = None if bias is None else bias.detach().float().contiguous()
        ptr = lambda t: None if t is None else t.data_ptr()
        # groups of the core: the f frames of a site (rows hw apart), or - spatial - the hw positions of a frame (consecutive rows)
        core = (b * f, hw, 1) if spatial else (b, f, hw)
        # focus: bool / uint8 [b], samples whose frames attend to themselves only (the focus-present mask, vdp.py:590-600)
        foc = None if (focus is None or spatial) else focus.detach().to(device=x.device, dtype=torch.uint8).contiguous()
        check(lib.vdmi_tattn_core_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), ptr(bias_c), ptr(cos), ptr(sin), ptr(foc),
                                      *core, heads, scale, _stream()), "vdmi_tattn_core_fwd")
        y = conv_rows(o, _linear_pack(wout.detach().contiguous(), False), None, b * f, f, h, w, residual=xr)
        ctx.save_for_backward(xr, xh, qkv, o, lse, wqkv, wout, gamma, bias_c, cos, sin, foc)
        ctx.meta = (b, c, f, h, w, heads, scale, eps, core)
        ctx.act = _act_dtype(x)
        return from_rows(y, b, f, h, w, ctx.act)

    @staticmethod
    @_plain
    def backward(ctx, dy):
        xr, xh, qkv, o, lse, wqkv, wout, gamma, bias_c, cos, sin, foc = ctx.saved_tensors
        b, c, f, h, w, heads, scale, eps, core = ctx.meta
        lib = _lib.lib()
        hw, rows, hid, frames = h * w, b * f * h * w, heads * 32, b * f
        ptr = lambda t: None if t is None else t.data_ptr()
        dyr = to_rows(dy)
        # to_out: gradient of its input and of its weight
        need = ctx.needs_input_grad
        do = conv_rows(dyr, _linear_pack(wout.detach().contiguous(), True), None, frames, f, h, w)
        dwout = None
        if need[3]:
            dwout = rows_wgrad(o, dyr)
        # attention core
        dqkv = torch.empty_like(qkv)
        dbias = torch.zeros_like(bias_c) if (bias_c is not None and need[4]) else None
        ws_bytes = lib.vdmi_tattn_core_bwd_workspace_bytes(*core, heads) if dbias is not None else 0
        ws = torch.empty(ws_bytes, device=qkv.device, dtype=torch.uint8) if ws_bytes else None
        check(lib.vdmi_tattn_core_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                      ptr(dbias), ptr(bias_c), ptr(cos), ptr(sin), ptr(foc), *core, heads, scale, ptr(ws),
                                      _stream()), "vdmi_tattn_core_bwd")
        # to_qkv (with the folded gamma), LayerNorm, residual
        dxh = conv_rows(dqkv, _linear_pack(wqkv.detach(), True, gamma), None, frames, f, h, w)
        dwqkv = dgamma = None
        if need[1] or need[2]:
            dwp = rows_wgrad(xh, dqkv)
            dwqkv = dwp * gamma.detach().reshape(1, c)
            dgamma = (dwp * wqkv.detach()).sum(0).reshape(gamma.shape)
        dx = torch.empty_like(xr)
        check(lib.vdmi_layernorm_rows_bwd(xr.data_ptr(), dxh.data_ptr(), dyr.data_ptr(), dx.data_ptr(), rows, c, eps,
                                          _stream()), "vdmi_layernorm_rows_bwd")
        return from_rows(dx, b, f, h, w, ctx.act), dgamma, dwqkv, dwout, dbias, None, None, None, None, None, None, None


_LOG2E = 1.4426950408889634


def _stored_feature_order(hid: int, device) -> torch.Tensor:
    """Column 32*head + 16*hh + r of the cross-attention core's output holds feature 32*head + (r&3) + 8*(r>>2) + 4*hh."""
    return torch.tensor([32 * hd + (r & 3) + 8 * (r >> 2) + 4 * hh for hd in range(hid // 32) for hh in (0, 1) for r in range(16)],
                        device=device)


class CrossAttnFn(torch.autograd.Function):
    """Residual(PreNorm(CrossAttention)) over x (B, C, F, H, W) and a context (B, L, D): vdp.py:380-389 (LayerNorm),
    480-528 (CrossAttention.forward with heads of 32 features), 331-337 (Residual), forward and backward on the HIP
    kernels.  gamma (1, C, 1, 1, 1); wq (heads*32, C); wk, wv (heads*32, D); wout (C, heads*32); bout (C)."""

    @staticmethod
    @_plain
    def forward(ctx, x, context, gamma, wq, wk, wv, wout, bout, heads, scale, eps):
        if not x.is_cuda:
            raise _lib.VdmiError("the HIP training backend needs CUDA (HIP) tensors; there is no CPU path")
        lib = _lib.lib()
        b, c, f, h, w = x.shape
        n, hid, L, cd = f * h * w, heads * 32, context.shape[1], context.shape[2]
        rows = b * n
        assert wq.shape == (hid, c) and wk.shape == (hid, cd) and wv.shape == (hid, cd) and wout.shape == (c, hid) and cd % 8 == 0
        xr = to_rows(x)
        xh = torch.empty_like(xr)
        check(lib.vdmi_layernorm_rows(xr.data_ptr(), xh.data_ptr(), rows, c, eps, _stream()), "vdmi_layernorm_rows")
        cs = scale * _LOG2E                     # the core exponentiates in the log2 domain
        wq_eff = (wq.detach() * gamma.detach().reshape(1, c) * cs).contiguous()
        q = conv_rows(xh, _linear_pack(wq_eff, False), None, b * f, f, h, w)
        crow = context.detach().to(torch.bfloat16).reshape(b * L, cd).contiguous()
        wkv = torch.cat((wk.detach(), wv.detach()), dim=0).contiguous()
        kvrows = conv_rows(crow, _linear_pack(wkv, False), None, b, 1, 1, L)          # [b*L][K columns | V columns]
        pack = torch.empty(2 * lib.vdmi_xattn_kv_elems(1, b, L, heads), device=x.device, dtype=torch.bfloat16)
        check(lib.vdmi_xattn_pack_kv_train(kvrows.data_ptr(), 2 * hid, pack.data_ptr(), b, L, heads, _stream()),
              "vdmi_xattn_pack_kv_train")
        npad = (n + 31) // 32 * 32
        o = torch.empty(rows, hid, device=x.device, dtype=torch.bfloat16)
        lse = torch.zeros(b, heads, npad, device=x.device, dtype=torch.float32)
        check(lib.vdmi_xattn_core_train_fwd(q.data_ptr(), hid, o.data_ptr(), pack.data_ptr(), lse.data_ptr(), npad, b, n, L,
                                            heads, _stream()), "vdmi_xattn_core_train_fwd")
        perm = _stored_feature_order(hid, x.device)
        wout_p = wout.detach()[:, perm].contiguous()
        y = conv_rows(o, _linear_pack(wout_p, False), None if bout is None else bout.detach().float().contiguous(), b * f, f, h,
                      w, residual=xr)
        ctx.save_for_backward(xr, xh, q, crow, kvrows, pack, o, lse, wq_eff, wkv, wout_p, perm, wq, gamma)
        ctx.meta = (b, c, f, h, w, heads, cs, eps, L, cd, npad, context.dtype)
        ctx.act = _act_dtype(x)
        return from_rows(y, b, f, h, w, ctx.act)

    @staticmethod
    @_plain
    def backward(ctx, dy):
        xr, xh, q, crow, kvrows, pack, o, lse, wq_eff, wkv, wout_p, perm, wq, gamma = ctx.saved_tensors
        b, c, f, h, w, heads, cs, eps, L, cd, npad, ctx_dtype = ctx.meta
        lib = _lib.lib()
        n, hid, frames = f * h * w, heads * 32, b * f
        rows = b * n
        need = ctx.needs_input_grad
        dyr = to_rows(dy)
        do = conv_rows(dyr, _linear_pack(wout_p, True), None, frames, f, h, w)          # stored feature order, like o
        dwout = dbout = None
        if need[6]:
            dwout = torch.empty(c, hid, device=dy.device, dtype=torch.float32)
            dwout[:, perm] = rows_wgrad(o, dyr)
        if need[7]:
            dbout = colsum_rows(dyr)
        dq = torch.empty(rows, hid, device=dy.device, dtype=torch.bfloat16)
        dkv = torch.empty(b * L, 2 * hid, device=dy.device, dtype=torch.bfloat16)
        delta = torch.zeros(b, heads, npad, device=dy.device, dtype=torch.float32)
        ws = torch.empty(lib.vdmi_xattn_bwd_workspace_bytes(b, n, L, heads), device=dy.device, dtype=torch.uint8)
        check(lib.vdmi_xattn_core_train_bwd(q.data_ptr(), hid, o.data_ptr(), do.data_ptr(), pack.data_ptr(), kvrows.data_ptr(),
                                            2 * hid, lse.data_ptr(), delta.data_ptr(), npad, dq.data_ptr(), dkv.data_ptr(),
                                            2 * hid, b, n, L, heads, ws.data_ptr(), _stream()), "vdmi_xattn_core_train_bwd")
        # to_q (with the folded gamma and scale), LayerNorm, residual
        dxh = conv_rows(dq, _linear_pack(wq_eff, True), None, frames, f, h, w)
        dwq = dgamma = None
        if need[2] or need[3]:
            dwe = rows_wgrad(xh, dq)
            dwq = dwe * (gamma.detach().reshape(1, c) * cs)
            dgamma = (dwe * wq.detach() * cs).sum(0).reshape(gamma.shape)
        dx = torch.empty_like(xr)
        check(lib.vdmi_layernorm_rows_bwd(xr.data_ptr(), dxh.data_ptr(), dyr.data_ptr(), dx.data_ptr(), rows, c, eps,
                                          _stream()), "vdmi_layernorm_rows_bwd")
        # to_k / to_v and the context
        dctx = dwk = dwv = None
        if need[1]:
            dctx = conv_rows(dkv, _linear_pack(wkv, True), None, b, 1, 1, L).view(b, L, cd).to(ctx_dtype)
        if need[4] or need[5]:
            dwkv = rows_wgrad(crow, dkv)
            dwk, dwv = dwkv[:hid].contiguous(), dwkv[hid:].contiguous()
        return from_rows(dx, b, f, h, w, ctx.act), dctx, dgamma, dwq, dwk, dwv, dwout, dbout, None, None, None


class SelfAttnFn(torch.autograd.Function):
    """Residual(PreNorm(SpatioTemporalAttention)) of the version-1 variant (version1diffusion.py:325-345): softmax attention
    of every token of a sample to all f*h*w tokens of that sample, no bias, no rotary - the cross-attention kernels with
    the sample's own normalised rows as the context.  gamma (1, C, 1, 1, 1); wqkv (3*heads*32, C); wout (C, heads*32)."""

    @staticmethod
    @_plain
    def forward(ctx, x, gamma, wqkv, wout, heads, scale, eps):
        if not x.is_cuda:
            raise _lib.VdmiError("the HIP training backend needs CUDA (HIP) tensors; there is no CPU path")
        lib = _lib.lib()
        b, c, f, h, w = x.shape
        n, hid = f * h * w, heads * 32
        rows = b * n
        assert wqkv.shape == (3 * hid, c) and wout.shape == (c, hid)
        xr = to_rows(x)
        xh = torch.empty_like(xr)
        check(lib.vdmi_layernorm_rows(xr.data_ptr(), xh.data_ptr(), rows, c, eps, _stream()), "vdmi_layernorm_rows")
        cs = scale * _LOG2E
        fold = torch.ones(3 * hid, 1, device=x.device)
        fold[:hid] = cs                                    # q rows also carry the softmax scale and log2 e
        w_eff = (wqkv.detach() * gamma.detach().reshape(1, c) * fold).contiguous()
        qkv = conv_rows(xh, _linear_pack(w_eff, False), None, b * f, f, h, w)        # [rows][q | k | v]
        kv_ptr = qkv.data_ptr() + hid * 2                                            # k, v columns: row stride 3 * hid
        pack = torch.empty(2 * lib.vdmi_xattn_kv_elems(1, b, n, heads), device=x.device, dtype=torch.bfloat16)
        check(lib.vdmi_xattn_pack_kv_train(kv_ptr, 3 * hid, pack.data_ptr(), b, n, heads, _stream()), "vdmi_xattn_pack_kv_train")
        npad = (n + 31) // 32 * 32
        o = torch.empty(rows, hid, device=x.device, dtype=torch.bfloat16)
        lse = torch.zeros(b, heads, npad, device=x.device, dtype=torch.float32)
        check(lib.vdmi_xattn_core_train_fwd(qkv.data_ptr(), 3 * hid, o.data_ptr(), pack.data_ptr(), lse.data_ptr(), npad, b, n, n,
                                            heads, _stream()), "vdmi_xattn_core_train_fwd")
        perm = _stored_feature_order(hid, x.device)
        wout_p = wout.detach()[:, perm].contiguous()
        y = conv_rows(o, _linear_pack(wout_p, False), None, b * f, f, h, w, residual=xr)
        ctx.save_for_backward(xr, xh, qkv, pack, o, lse, w_eff, wout_p, perm, wqkv, gamma, fold)
        ctx.meta = (b, c, f, h, w, heads, eps, npad)
        ctx.act = _act_dtype(x)
        return from_rows(y, b, f, h, w, ctx.act)

    @staticmethod
    @_plain
    def backward(ctx, dy):
        xr, xh, qkv, pack, o, lse, w_eff, wout_p, perm, wqkv, gamma, fold = ctx.saved_tensors
        b, c, f, h, w, heads, eps, npad = ctx.meta
        lib = _lib.lib()
        n, hid, frames = f * h * w, heads * 32, b * f
        rows = b * n
        need = ctx.needs_input_grad
        dyr = to_rows(dy)
        do = conv_rows(dyr, _linear_pack(wout_p, True), None, frames, f, h, w)
        dwout = None
        if need[3]:
            dwout = torch.empty(c, hid, device=dy.device, dtype=torch.float32)
            dwout[:, perm] = rows_wgrad(o, dyr)
        dq = torch.empty(rows, hid, device=dy.device, dtype=torch.bfloat16)
        dkv = torch.empty(rows, 2 * hid, device=dy.device, dtype=torch.bfloat16)
        delta = torch.zeros(b, heads, npad, device=dy.device, dtype=torch.float32)
        ws = torch.empty(lib.vdmi_xattn_bwd_workspace_bytes(b, n, n, heads), device=dy.device, dtype=torch.uint8)
        check(lib.vdmi_xattn_core_train_bwd(qkv.data_ptr(), 3 * hid, o.data_ptr(), do.data_ptr(), pack.data_ptr(),
                                            qkv.data_ptr() + hid * 2, 3 * hid, lse.data_ptr(), delta.data_ptr(), npad,
                                            dq.data_ptr(), dkv.data_ptr(), 2 * hid, b, n, n, heads, ws.data_ptr(), _stream()),
              "vdmi_xattn_core_train_bwd")
        dqkv = torch.cat((dq, dkv), dim=1)
        dxh = conv_rows(dqkv, _linear_pack(w_eff, True), None, frames, f, h, w)
        dwqkv = dgamma = None
        if need[1] or need[2]:
            dwe = rows_wgrad(xh, dqkv)
            dwqkv = dwe * gamma.detach().reshape(1, c) * fold
            dgamma = (dwe * wqkv.detach() * fold).sum(0).reshape(gamma.shape)
        dx = torch.empty_like(xr)
        check(lib.vdmi_layernorm_rows_bwd(xr.data_ptr(), dxh.data_ptr(), dyr.data_ptr(), dx.data_ptr(), rows, c, eps,
                                          _stream()), "vdmi_layernorm_rows_bwd")
        return from_rows(dx, b, f, h, w, ctx.act), dgamma, dwqkv, dwout, None, None, None


def spatio_temporal_attention(res, x: torch.Tensor) -> torch.Tensor:
    """autograd.py's `_linear_attention` (the SpatioTemporalAttention form of the version-1 variant) on the HIP kernels."""
    att = res.fn.fn
    return SelfAttnFn.apply(x, res.fn.norm.gamma, att.to_qkv.weight, att.to_out.weight, att.heads, float(att.scale),
                            float(res.fn.norm.eps))


def cross_attention_fits(ca, context) -> bool:
    """Heads of 32 features and a context width the row kernels take."""
    return ca.to_q.out_features == ca.heads * 32 and context.shape[2] % 8 == 0 and ca.to_q.in_features % 8 == 0


def cross_attention(res, x: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
    """autograd.py's `_cross_attention` on the HIP kernels."""
    ca = res.fn.fn
    return CrossAttnFn.apply(x, context, res.fn.norm.gamma, ca.to_q.weight, ca.to_k.weight, ca.to_v.weight, ca.to_out.weight,
                             ca.to_out.bias, ca.heads, float(ca.scale), float(res.fn.norm.eps))


def attention_core_fits(n_tok: int, heads: int, with_bias: bool) -> bool:
    """Whether the attention core kernels (forward AND backward, the latter with the bias-gradient table [heads][n][n] in
    LDS when the layer has a relative position bias) take a token axis of n_tok positions: 8 heads with a bias end near
    62 frames.  The dispatchers in autograd.py fall back to torch ops beyond it."""
    lib = _lib.lib()
    return bool(lib.vdmi_tattn_core_fits(0, n_tok, heads, 0)) and bool(lib.vdmi_tattn_core_fits(1, n_tok, heads, int(with_bias)))


def spatial_attention(res, x: torch.Tensor) -> torch.Tensor:
    """autograd.py's `_spatial` (the mid-block attention over the positions of each frame, vdp.py:708-710: no rotary, no
    bias) on the HIP kernels."""
    att = res.fn.fn.fn
    if att.dim_head != 32:
        raise _lib.VdmiError("the HIP spatial attention takes heads of 32 features")
    b, c, f, h, w = x.shape
    if h * w <= 32:         # a frame's positions fit one tile of the attention core over short token axes
        return TemporalAttnFn.apply(x, res.fn.norm.gamma, att.to_qkv.weight, att.to_out.weight, None, None, None, att.heads,
                                    float(att.scale), float(res.fn.norm.eps), True)
    # every frame as a sample of its own on the matrix-core self-attention (the cross-attention kernels with the frame's
    # own rows as context): views only, the rows stay where they are
    xf = x.transpose(1, 2).reshape(b * f, c, h, w).unsqueeze(2)
    y = SelfAttnFn.apply(xf, res.fn.norm.gamma, att.to_qkv.weight, att.to_out.weight, att.heads, float(att.scale),
                         float(res.fn.norm.eps))
    return y.squeeze(2).reshape(b, f, c, h, w).transpose(1, 2)


def temporal_attention(res, x: torch.Tensor, pos_bias, rot_tables, focus=None) -> torch.Tensor:
    """autograd.py's `_temporal` on the HIP kernels.  rot_tables: (cos, sin) float [F][16] or None; focus: bool [B] or None."""
    att = res.fn.fn.fn
    if att.dim_head != 32:
        raise _lib.VdmiError("the HIP temporal attention takes heads of 32 features")
    cos, sin = rot_tables if rot_tables is not None else (None, None)
    return TemporalAttnFn.apply(x, res.fn.norm.gamma, att.to_qkv.weight, att.to_out.weight, pos_bias, cos, sin, att.heads,
                                float(att.scale), float(res.fn.norm.eps), False, focus)


# ---------------------------------------------------------------------------------------------------------------------
# Downsample / Upsample (vdp.py:374-378): Conv3d (1,4,4) / stride 2 / pad 1 and its transposed twin
# ---------------------------------------------------------------------------------------------------------------------
_UP_PAIRS = {0: [(1, 0), (3, -1)], 1: [(0, 1), (2, 0)]}     # output parity -> (kernel index, input offset) pairs


_GATHERS = {}


def _resample_gathers(kind: str, weight: torch.Tensor):
    """Index tables (int32, positions in the flattened parameter) of the weight sets the resampling convolutions run on:
    "down": [n][4 cin][4 taps], the 2x2 / stride-1 kernel over the space-to-depth view; "up": four [n][cin][4 taps], one
    per output parity.  The re-arrangement is applied to an index tensor once per shape and device."""
    key = (kind, tuple(weight.shape), weight.device)
    hit = _GATHERS.get(key)
    if hit is None:
        idx = torch.arange(weight.numel(), dtype=torch.int32, device=weight.device).view(weight.shape)[:, :, 0]
        if kind == "down":
            hit = [torch.cat([torch.stack([idx[:, :, 2 * dy + a, 2 * dx + b] for dy in (0, 1) for dx in (0, 1)], dim=-1)
                              for a in (0, 1) for b in (0, 1)], dim=1).contiguous()]
        else:
            hit = [torch.stack([idx[:, :, ky, kx].t() for (ky, _) in _UP_PAIRS[pp] for (kx, _) in _UP_PAIRS[q]], dim=-1).contiguous()
                   for pp in (0, 1) for q in (0, 1)]
        _GATHERS[key] = hit
    return hit


def down_conv_rows(x: torch.Tensor, weight: torch.Tensor, bias, frames: int, fpb: int, h: int, w: int) -> torch.Tensor:
    """Conv3d(cin, n, (1,4,4), stride 2, pad 1) over rows of an h x w image -> rows of the (h/2) x (w/2) image, as a
    2x2 / stride-1 tap-GEMM over the space-to-depth view of x.  weight (n, cin, 1, 4, 4)."""
    n, cin = weight.shape[0], weight.shape[1]
    if h % 2 or w % 2 or cin % 64:
        raise _lib.VdmiError("the HIP resampling convolutions take even images and channel widths that are multiples of 64")
    # channel (2a+b)*cin + ci, tap (dY, dX)  <-  kernel element (2dY+a, 2dX+b): an index table, the packer gathers
    pack = param_pack(weight, False, 64, gather=_resample_gathers("down", weight)[0], shape=(n, 4 * cin, 4))
    oh, ow = h // 2, w // 2
    out = torch.empty(frames * oh * ow, n, device=x.device, dtype=torch.bfloat16)
    a = TapGemmArgs()
    a.frames, a.frames_per_batch = frames, fpb
    a.grid_h, a.grid_w, a.out_h, a.out_w = oh, ow, oh, ow
    a.out_mul = 1
    a.out, a.ldo, a.n, a.epilogue = out.data_ptr(), n, n, EPI_PLAIN
    p = a.main
    p.src0, p.c0, p.c1 = x.data_ptr(), 4 * cin, 0
    p.in_h, p.in_w, p.stride, p.ntaps = h, w, 1, 4
    for i, (dy, dx) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        p.dy[i], p.dx[i] = dy, dx
    p.w = pack.data.data_ptr()
    p.space_to_depth = 1
    a.bias = None if bias is None else bias.data_ptr()
    a.tile_bn = 64
    check(_lib.lib().vdmi_tapgemm(C.byref(a), _stream()), "vdmi_tapgemm")
    return out


def up_conv_rows(x: torch.Tensor, weight: torch.Tensor, bias, frames: int, fpb: int, h: int, w: int) -> torch.Tensor:
    """ConvTranspose3d(cin, n, (1,4,4), stride 2, pad 1) over rows of an h x w image -> rows of the 2h x 2w image: the four
    output parities as four 4-tap phases of one launch.  weight (cin, n, 1, 4, 4)."""
    cin, n = weight.shape[0], weight.shape[1]
    # one 4-tap weight set [n][cin][tap] per output parity (pp, q), gathered from the parameter by index tables
    ph = phase_packs(weight, _resample_gathers("up", weight), (n, cin, 4), 0)
    wall = ph.data
    oh, ow = 2 * h, 2 * w
    out = torch.empty(frames * oh * ow, n, device=x.device, dtype=torch.bfloat16)
    taps = [(dy, dx) for pp in (0, 1) for q in (0, 1) for (_, dy) in _UP_PAIRS[pp] for (_, dx) in _UP_PAIRS[q]]
    a = TapGemmArgs()
    a.frames, a.frames_per_batch = frames, fpb
    a.grid_h, a.grid_w, a.out_h, a.out_w = h, w, oh, ow
    a.out_mul, a.out_off_y, a.out_off_x = 2, 0, 0
    a.out, a.ldo, a.n, a.epilogue = out.data_ptr(), n, n, EPI_PLAIN
    p = a.main
    p.src0, p.c0, p.c1 = x.data_ptr(), cin, 0
    p.in_h, p.in_w, p.stride, p.ntaps = h, w, 1, len(taps)
    for i, (dy, dx) in enumerate(taps):
        p.dy[i], p.dx[i] = dy, dx
    p.w = wall.data_ptr()
    p.phases, p.w_phase_stride = 4, ph.stride
    a.bias = None if bias is None else bias.data_ptr()
    check(_lib.lib().vdmi_tapgemm(C.byref(a), _stream()), "vdmi_tapgemm")
    return out


# VDMI_STRIDED_WGRAD=0: the resampling convolutions' weight gradient as four one-tap gradients over a padded
# space-to-depth copy (A/B switch)
_STRIDED_WGRAD_ROWS = os.environ.get("VDMI_STRIDED_WGRAD", "1") != "0"


def strided_wgrad(big: torch.Tensor, small: torch.Tensor, frames: int, h: int, w: int) -> torch.Tensor:
    """dW[cs][cb][ky][kx] = sum over frames and small-image pixels q of small[q][cs] * big[2q + (ky-1, kx-1)][cb]: the
    weight gradient of the (1,4,4) / stride 2 / pad 1 convolution from big (h x w) to small (h/2 x w/2), and - with the
    roles of input and output gradient swapped - of its transposed twin.  Four one-tap gradients over the
    space-to-depth view of `big`."""
    cb, cs = big.shape[1], small.shape[1]
    oh, ow = h // 2, w // 2
    if _STRIDED_WGRAD_ROWS:
        # one launch over the rows as they lie (+ the partial-tile reduce); below: the same sum as four one-tap gradients
        # over a padded space-to-depth copy of `big`
        assert big.shape[0] == frames * h * w and small.shape[0] == frames * oh * ow
        assert big.is_contiguous() and small.is_contiguous()
        lib = _lib.lib()
        dw = torch.empty(cs, cb, 1, 4, 4, device=big.device, dtype=torch.float32)
        ws = torch.empty(lib.vdmi_strided_wgrad_rows_workspace_bytes(cb, cs, frames, h, w), device=big.device, dtype=torch.uint8)
        check(lib.vdmi_strided_wgrad_rows(big.data_ptr(), cb, small.data_ptr(), cs, dw.data_ptr(), cb, cs, frames, h, w,
                                          ws.data_ptr(), _stream()), "vdmi_strided_wgrad_rows")
        return dw
    bp = F.pad(big.view(frames, h, w, cb), (0, 0, 1, 1, 1, 1))                      # zero border of one pixel
    s2d = bp.view(frames, oh + 1, 2, ow + 1, 2, cb).permute(0, 1, 3, 2, 4, 5).reshape(frames * (oh + 1) * (ow + 1), 4 * cb)
    sp = F.pad(small.view(frames, oh, ow, cs), (0, 0, 0, 1, 0, 1)).reshape(frames * (oh + 1) * (ow + 1), cs)
    # `small` is zero in its padding row and column, so a tap (ty, tx) is a plain shift of the flattened rows: where the
    # shifted row wraps into the next image row or frame, the other factor is zero
    s2d, sp = s2d.contiguous(), sp.contiguous()
    total = s2d.shape[0]
    d2 = []
    for ty in (0, 1):
        for tx in (0, 1):
            shift = ty * (ow + 1) + tx
            d2.append(rows_wgrad(s2d[shift:], sp[:total - shift]))              # [cs][(a, b, cb)]
    # [ty][tx][cs][a][b][cb] -> [cs][cb][2 ty + a][2 tx + b]
    return torch.stack(d2).view(2, 2, cs, 2, 2, cb).permute(2, 5, 0, 3, 1, 4).reshape(cs, cb, 1, 4, 4)


class DownsampleFn(torch.autograd.Function):
    @staticmethod
    @_plain
    def forward(ctx, x, weight, bias):
        b, c, f, h, w = x.shape
        xr = to_rows(x)
        y = down_conv_rows(xr, weight, bias.detach().float().contiguous(), b * f, f, h, w)
        ctx.save_for_backward(xr, weight)
        ctx.dims = (b, f, h, w)
        ctx.act = _act_dtype(x)
        return from_rows(y, b, f, h // 2, w // 2, ctx.act)

    @staticmethod
    @_plain
    def backward(ctx, dy):
        xr, weight = ctx.saved_tensors
        b, f, h, w = ctx.dims
        dyr = to_rows(dy)
        # the input gradient of a strided convolution is the transposed convolution with the same weight tensor
        dx = up_conv_rows(dyr, weight, None, b * f, f, h // 2, w // 2)
        dw = strided_wgrad(xr, dyr, b * f, h, w) if ctx.needs_input_grad[1] else None
        return from_rows(dx, b, f, h, w, ctx.act), dw, colsum_rows(dyr) if ctx.needs_input_grad[2] else None


class UpsampleFn(torch.autograd.Function):
    @staticmethod
    @_plain
    def forward(ctx, x, weight, bias):
        b, c, f, h, w = x.shape
        xr = to_rows(x)
        y = up_conv_rows(xr, weight, bias.detach().float().contiguous(), b * f, f, h, w)
        ctx.save_for_backward(xr, weight)
        ctx.dims = (b, f, h, w)
        ctx.act = _act_dtype(x)
        return from_rows(y, b, f, 2 * h, 2 * w, ctx.act)

    @staticmethod
    @_plain
    def backward(ctx, dy):
        xr, weight = ctx.saved_tensors
        b, f, h, w = ctx.dims
        dyr = to_rows(dy)
        # ... and the input gradient of the transposed convolution is the strided convolution with the same weight tensor
        dx = down_conv_rows(dyr, weight, None, b * f, f, 2 * h, 2 * w)
        dw = strided_wgrad(dyr, xr, b * f, 2 * h, 2 * w) if ctx.needs_input_grad[1] else None
        return from_rows(dx, b, f, h, w, ctx.act), dw, colsum_rows(dyr) if ctx.needs_input_grad[2] else None


def resample(mod, x: torch.Tensor) -> torch.Tensor:
    """Downsample / Upsample modules of the reference (vdp.py:374-378) on the HIP kernels; anything else runs as it is."""
    k, st, pd = (1, 4, 4), (1, 2, 2), (0, 1, 1)
    if isinstance(mod, torch.nn.ConvTranspose3d) and (mod.kernel_size, mod.stride, mod.padding) == (k, st, pd):
        return UpsampleFn.apply(x, mod.weight, mod.bias)
    if isinstance(mod, torch.nn.Conv3d) and (mod.kernel_size, mod.stride, mod.padding) == (k, st, pd):
        return DownsampleFn.apply(x, mod.weight, mod.bias)
    return mod(x)


# ---------------------------------------------------------------------------------------------------------------------
# Residual(PreNorm(SpatialLinearAttention)), vdp.py:454-479
# ---------------------------------------------------------------------------------------------------------------------
def _sla_reduce(a_ptr, lda, b_ptr, ldb, comb_ctx, frames, n, heads, mode, device):
    """Sums over the positions -> one combined record per (frame, head) (float tensor)."""
    lib = _lib.lib()
    scratch = torch.empty(lib.vdmi_sla_train_partial_bytes(frames, n, heads), device=device, dtype=torch.uint8)
    comb = torch.empty(lib.vdmi_sla_train_combined_bytes(frames, heads) // 4, device=device, dtype=torch.float32)
    check(lib.vdmi_sla_train_reduce(a_ptr, lda, b_ptr, ldb, scratch.data_ptr(), None if comb_ctx is None else comb_ctx.data_ptr(),
                                    comb.data_ptr(), frames, n, heads, mode, _stream()), "vdmi_sla_train_reduce")
    return comb


class SlaFn(torch.autograd.Function):
    """x float (B, C, F, H, W); gamma (1, C, 1, 1, 1); wqkv (3*heads*32, C, 1, 1) without bias; wout (C, heads*32, 1, 1),
    bout (C).  Forward and backward on the HIP kernels."""

    @staticmethod
    @_plain
    def forward(ctx, x, gamma, wqkv, wout, bout, heads, scale, eps):
        if not x.is_cuda:
            raise _lib.VdmiError("the HIP training backend needs CUDA (HIP) tensors; there is no CPU path")
        lib = _lib.lib()
        b, c, f, h, w = x.shape
        n, frames, hid = h * w, b * f, heads * 32
        rows = frames * n
        xr = to_rows(x)
        xh = torch.empty_like(xr)
        check(lib.vdmi_layernorm_rows(xr.data_ptr(), xh.data_ptr(), rows, c, eps, _stream()), "vdmi_layernorm_rows")
        wo2 = wout.detach().reshape(c, hid)
        qkv = conv_rows(xh, _linear_pack(wqkv.detach(), False, gamma), None, frames, f, h, w)   # gamma folded while packing
        base = qkv.data_ptr()
        p_ctx = _sla_reduce(base + 2 * hid, 3 * hid, base + 4 * hid, 3 * hid, None, frames, n, heads, 0, x.device)
        out = torch.empty(rows, hid, device=x.device, dtype=torch.bfloat16)
        qt = torch.empty_like(out)
        check(lib.vdmi_sla_train_rows_fwd(qkv.data_ptr(), p_ctx.data_ptr(), out.data_ptr(), qt.data_ptr(), frames, n, heads,
                                          scale, _stream()), "vdmi_sla_train_rows_fwd")
        y = conv_rows(out, _linear_pack(wo2.contiguous(), False), bout.detach().float().contiguous(), frames, f, h, w,
                      residual=xr)
        ctx.save_for_backward(xr, xh, qkv, qt, out, p_ctx, wqkv, wout, gamma)
        ctx.meta = (b, c, f, h, w, heads, scale, eps)
        ctx.act = _act_dtype(x)
        return from_rows(y, b, f, h, w, ctx.act)

    @staticmethod
    @_plain
    def backward(ctx, dy):
        xr, xh, qkv, qt, out, p_ctx, wqkv, wout, gamma = ctx.saved_tensors
        b, c, f, h, w, heads, scale, eps = ctx.meta
        lib = _lib.lib()
        n, frames, hid = h * w, b * f, heads * 32
        rows = frames * n
        dyr = to_rows(dy)
        wo2 = wout.detach().reshape(c, hid).contiguous()
        need = ctx.needs_input_grad
        do = conv_rows(dyr, _linear_pack(wo2, True), None, frames, f, h, w)
        dwout = None
        if need[3]:
            dwout = rows_wgrad(out, dyr).reshape(wout.shape)
        p_dctx = _sla_reduce(qt.data_ptr(), hid, do.data_ptr(), hid, p_ctx, frames, n, heads, 1, dy.device)
        dqkv = torch.empty_like(qkv)
        check(lib.vdmi_sla_train_rows_bwd(qkv.data_ptr(), qt.data_ptr(), do.data_ptr(), p_ctx.data_ptr(), p_dctx.data_ptr(),
                                          dqkv.data_ptr(), frames, n, heads, scale, _stream()), "vdmi_sla_train_rows_bwd")
        dxh = conv_rows(dqkv, _linear_pack(wqkv.detach(), True, gamma), None, frames, f, h, w)
        dwqkv = dgamma = None
        if need[1] or need[2]:
            dwp = rows_wgrad(xh, dqkv)
            dwqkv = (dwp * gamma.detach().reshape(1, c)).reshape(wqkv.shape)
            dgamma = (dwp * wqkv.detach().reshape(3 * hid, c)).sum(0).reshape(gamma.shape)
        dx = torch.empty_like(xr)
        check(lib.vdmi_layernorm_rows_bwd(xr.data_ptr(), dxh.data_ptr(), dyr.data_ptr(), dx.data_ptr(), rows, c, eps,
                                          _stream()), "vdmi_layernorm_rows_bwd")
        return (from_rows(dx, b, f, h, w, ctx.act), dgamma, dwqkv, dwout, colsum_rows(dyr) if need[4] else None, None, None, None)


def spatial_linear_attention(res, x: torch.Tensor) -> torch.Tensor:
    """autograd.py's `_linear_attention` (the SpatialLinearAttention form) on the HIP kernels."""
    sl = res.fn.fn
    if sl.dim_head != 32 or sl.to_qkv.bias is not None:
        raise _lib.VdmiError("the HIP linear attention takes heads of 32 features and a to_qkv without bias")
    return SlaFn.apply(x, res.fn.norm.gamma, sl.to_qkv.weight, sl.to_out.weight, sl.to_out.bias, sl.heads, float(sl.scale),
                       float(res.fn.norm.eps))
