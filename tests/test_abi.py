"""CPU: the C-ABI library loads, exports every symbol include/vdmi.h declares, and its host-side
routines (weight packing, relative-position buckets, status strings) behave.  No GPU launches."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from tests.helpers import golden
from video_diffusion_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vdmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vdmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.EXPORTED_SYMBOLS)
    assert lib.vdmi_version() >= 100
    assert lib.vdmi_strerror(0) == b"ok"
    assert b"shape" in lib.vdmi_strerror(-2)


def test_half_precision_library_exports_the_same_symbols_and_packs_ieee_half():
    """libvdmi_f16.so is the same sources built with IEEE half in place of bfloat16 (VDMI_PRECISION=f16 selects it):
    same entry points; its host-side packer rounds weights to half, the default library's to bfloat16."""
    path = os.path.join(os.path.dirname(_lib.__file__), "libvdmi_f16.so")
    assert os.path.exists(path), "build it with make -C video_diffusion_amd/csrc"
    h16 = C.CDLL(path)
    for n in declared_symbols():
        assert hasattr(h16, n), n
    w = np.array([[[1.0 + 2.0 ** -9]]], dtype=np.float32)        # exact in half (10 fraction bits), not in bfloat16 (7)
    w = np.tile(w, (32, 16, 1)).astype(np.float32)
    outs = {}
    for name, lib in (("f16", h16), ("bf16", C.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libvdmi.so")))):
        lib.vdmi_pack_taps_elems.restype = C.c_size_t
        lib.vdmi_pack_taps_elems.argtypes = [C.c_int] * 6
        lib.vdmi_pack_taps.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]
        elems = lib.vdmi_pack_taps_elems(32, 64, 16, 0, 1, 64)
        out = np.zeros(elems, dtype=np.uint16)
        assert lib.vdmi_pack_taps(w.ctypes.data, 32, 64, 16, 0, 1, 64, out.ctypes.data, None) == 0
        outs[name] = out
    assert float(torch.from_numpy(outs["f16"][:1].view(np.int16)).view(torch.float16)[0]) == 1.0 + 2.0 ** -9
    assert float(torch.from_numpy(outs["bf16"][:1].view(np.int16)).view(torch.bfloat16)[0]) == 1.0


def test_relpos_buckets_match_reference():
    lib = _lib.lib()
    g = golden("relpos_buckets")
    for k, v in g.items():
        _, md, n = k.split("_")
        n, md = int(n[1:]), int(md[2:])
        out = np.empty((n, n), dtype=np.int32)
        assert lib.vdmi_relpos_buckets(n, 32, md, out.ctypes.data) == 0
        assert np.array_equal(out, v.numpy().astype(np.int32)), k


def _bf16(x):
    return torch.from_numpy(x).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize("n,cin0,cin1,ntaps,bkc", [(32, 32, 0, 9, 64), (64, 64, 64, 1, 64), (96, 40, 24, 4, 32)])
def test_pack_taps_layout(n, cin0, cin1, ntaps, bkc):
    lib = _lib.lib()
    rng = np.random.default_rng(0)
    cin = cin0 + cin1
    w = rng.standard_normal((n, cin, ntaps)).astype(np.float32)
    npad = (n + 63) // 64 * 64
    elems = lib.vdmi_pack_taps_elems(n, npad, cin0, cin1, ntaps, bkc)
    out = np.zeros(elems, dtype=np.uint16)
    colsum = np.zeros(n, dtype=np.float32)
    assert lib.vdmi_pack_taps(w.ctypes.data, n, npad, cin0, cin1, ntaps, bkc, out.ctypes.data, colsum.ctypes.data) == 0
    nch0, nch1 = -(-cin0 // bkc), -(-cin1 // bkc)
    assert elems == (nch0 + nch1) * ntaps * npad * bkc
    packed = torch.from_numpy(out.view(np.int16)).view(torch.bfloat16).float().numpy()
    packed = packed.reshape(nch0 + nch1, ntaps, npad // 32, bkc // 16, 64, 8)
    wb = _bf16(w)
    for c in range(nch0 + nch1):
        first = c < nch0
        cbase, clim, coff = (c * bkc, cin0, 0) if first else ((c - nch0) * bkc, cin1, cin0)
        for lane in (0, 5, 31, 32, 63):
            for ks in range(bkc // 16):
                for j in (0, 7):
                    for ct in range(npad // 32):
                        co, ci = ct * 32 + (lane & 31), cbase + ks * 16 + 8 * (lane >> 5) + j
                        for t in range(ntaps):
                            want = wb[co, coff + ci, t] if (co < n and ci < clim) else 0.0
                            assert packed[c, t, ct, ks, lane, j] == want
    assert np.allclose(colsum, wb.sum(axis=(1, 2)), rtol=1e-5, atol=1e-5)


def test_bad_arguments_are_rejected_without_launching():
    lib = _lib.lib()
    assert lib.vdmi_tapgemm(None, None) == -1
    a = _lib.TapGemmArgs()
    assert lib.vdmi_tapgemm(C.byref(a), None) == -1           # null output
    assert lib.vdmi_attn_core(None, None, 0, 8, 10, 1, None, None, 0, None) == -1
    assert lib.vdmi_ddpm_step(None, None, None, None, None, None, 1, 1, 1, 0, 0, None) == -1
    bn, bkc = C.c_int(), C.c_int()
    assert lib.vdmi_tapgemm_tile(_lib.EPI_NORM_SILU, 64, C.byref(bn), C.byref(bkc)) == 0 and bn.value == 64
    assert lib.vdmi_tapgemm_tile(_lib.EPI_NORM_SILU, 4096, C.byref(bn), C.byref(bkc)) == -2
    with pytest.raises(_lib.VdmiError):
        _lib.check(-2, "probe")


def test_fused_attention_and_threshold_entry_points_check_their_arguments():
    """The wide fused attention family, the dynamic-threshold pair and the allocation-free GroupNorm refuse shapes and
    pointers they do not serve before anything is launched (no GPU needed: the checks come first)."""
    lib = _lib.lib()
    buf = (C.c_uint8 * 4096)()
    p = C.addressof(buf)
    p = (p + 255) & ~255          # 256-byte aligned fake device pointer: never dereferenced on these paths
    # temporal: channels, heads, frames
    assert lib.vdmi_tattn_fused_wide(None, p, p, None, None, None, 1, 10, 64, 512, 8, 0.1, 1e-5, None, None) == -1
    assert lib.vdmi_tattn_fused_wide(p, p, p, None, None, None, 1, 10, 64, 384, 8, 0.1, 1e-5, None, None) == -2
    assert lib.vdmi_tattn_fused_wide(p, p, p, None, None, None, 1, 10, 64, 512, 4, 0.1, 1e-5, None, None) == -2
    assert lib.vdmi_tattn_fused_wide(p, p, p, None, None, None, 1, 33, 64, 512, 8, 0.1, 1e-5, None, None) == -2
    assert lib.vdmi_tattn_fused_wide(p, p + 2, p, None, None, None, 1, 10, 64, 512, 8, 0.1, 1e-5, None, None) == -3
    # per-frame softmax attention: at most 64 positions
    assert lib.vdmi_sattn_fused_wide(p, p, p, 4, 65, 512, 8, 0.1, 1e-5, None) == -2
    assert lib.vdmi_sattn_fused_wide(p, p, p, 0, 64, 512, 8, 0.1, 1e-5, None) == -2
    assert lib.vdmi_sattn_fused_wide(p, None, p, 4, 64, 512, 8, 0.1, 1e-5, None) == -1
    # linear attention: larger frames only at 256 channels
    assert lib.vdmi_sla_fused_wide(p, p, p, None, 4, 256, 512, 8, 0.1, 1e-5, None) == -2
    assert lib.vdmi_sla_fused_wide(p, p, p, None, 4, 64, 128, 8, 0.1, 1e-5, None) == -2
    assert lib.vdmi_sla_fused_wide(p, p, p, p + 4, 4, 64, 256, 8, 0.1, 1e-5, None) == -3
    # dynamic threshold: percentile range, sample size (float32 ranks), noise source of the step
    coefs = _lib.DdpmCoefs()
    for f in ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
              "posterior_log_variance_clipped"):
        setattr(coefs, f, p)
    assert lib.vdmi_dynamic_threshold(p, p, p, C.byref(coefs), p, 2, 100, 1.5, None) == -1
    assert lib.vdmi_dynamic_threshold(p, p, p, C.byref(coefs), p, 2, 1 << 24, 0.9, None) == -2
    assert lib.vdmi_dynamic_threshold(p, p, p, C.byref(coefs), None, 2, 100, 0.9, None) == -1
    assert lib.vdmi_ddpm_step_thres(p, p, None, p, C.byref(coefs), p, p, 2, 100, None, None) == -1      # no noise source
    assert lib.vdmi_ddpm_step_thres(p, p, p, p, C.byref(coefs), p, p, 2, 100, p, None) == -1            # two noise sources
    assert lib.vdmi_ddpm_step_thres(p, p, p, p, C.byref(coefs), None, p, 2, 100, None, None) == -1      # no thresholds
    # allocation-free GroupNorm: inputs beyond the static scratch
    assert lib.vdmi_groupnorm_silu_noalloc(p, p, 64, 1 << 22, 2048, 32, p, p, 1e-5, None) == -6


def test_batch_split_policy():
    """Host logic of the two-stream forward: which batches are run as sub-batches, and the launch flags of a plan."""
    from video_diffusion_amd import engine
    assert engine.batch_parts(16) == 2 and engine.batch_parts(32) == 2 and engine.batch_parts(4) == 2
    assert engine.batch_parts(1) == 1 and engine.batch_parts(2) == 1 and engine.batch_parts(3) == 1
    assert engine.batch_parts(7) == 2      # 4 + 3
    old = engine.BATCH_PARTS
    try:
        engine.BATCH_PARTS = 1
        assert engine.batch_parts(16) == 1
        engine.BATCH_PARTS = 4
        assert engine.batch_parts(16) == 4 and engine.batch_parts(6) == 1 and engine.batch_parts(9) == 4
    finally:
        engine.BATCH_PARTS = old
    c = engine._Call(lambda s: 0, None, "x", side=True)
    assert c.side and not c.join


def test_backward_entry_points_reject_bad_arguments():
    """Argument checks of the backward entry points (no GPU work: every call returns before a launch)."""
    import ctypes as C
    from video_diffusion_amd import _lib
    lib = _lib.lib()
    E_ARG, E_SHAPE, E_ALIGN = -1, -2, -3
    assert lib.vdmi_pack_taps_device(None, 64, 64, 64, 0, 9, 64, 0, None, None) == E_ARG
    assert lib.vdmi_pack_taps_device(16, 64, 48, 64, 0, 9, 64, 0, 16, None) == E_SHAPE          # npad < n
    assert lib.vdmi_pack_taps_device(16, 64, 64, 64, 0, 9, 64, 0, 24, None) == E_ALIGN
    assert lib.vdmi_rows_transpose_elems(64, 10, 100) == 64 * 100 * 16
    assert lib.vdmi_rows_transpose(None, 64, 0, 64, 10, 100, None, None) == E_ARG
    assert lib.vdmi_rows_transpose(16, 64, 0, 60, 10, 100, 16, None) == E_SHAPE                # c not a multiple of 8
    assert lib.vdmi_rows_transpose(16, 64, 8, 64, 10, 100, 16, None) == E_SHAPE                # window past the row
    taps = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
    dy = (C.c_int8 * 9)(*[t[0] for t in taps])
    dx = (C.c_int8 * 9)(*[t[1] for t in taps])
    assert lib.vdmi_conv_wgrad_workspace_bytes(64, 64, 10, 8, 8, 9) > 0
    assert lib.vdmi_conv_wgrad_workspace_bytes(64, 64, 10, 8, 8, 4) == 0                        # 1 or 9 taps only
    assert lib.vdmi_conv_wgrad(None, 16, 16, 64, 64, 64, 0, 10, 8, 8, 9, dy, dx, 16, None) == E_ARG
    assert lib.vdmi_conv_wgrad(16, 16, 16, 64, 64, 64, 8, 10, 8, 8, 9, dy, dx, 16, None) == E_SHAPE   # window past cin_total
    bad = (C.c_int8 * 9)(*[t[1] for t in taps])
    assert lib.vdmi_conv_wgrad(16, 16, 16, 64, 64, 64, 0, 10, 8, 8, 9, bad, dx, 16, None) == E_SHAPE  # not the 3 x 3 table
    assert lib.vdmi_conv_wgrad_gather(16, 16, 16, 3, 64, 10, 8, 8, 4, 4, 16, None) == E_SHAPE         # even kernel
    assert lib.vdmi_rows_wgrad_workspace_bytes(64, 64, 1000) > 0
    assert lib.vdmi_rows_wgrad_workspace_bytes(60, 64, 1000) == 0                                     # multiples of 8 only
    assert lib.vdmi_rows_wgrad(None, 64, 16, 64, 16, 64, 64, 64, 0, 1000, 16, None) == E_ARG
    assert lib.vdmi_rows_wgrad(16, 64, 16, 64, 16, 64, 64, 64, 8, 1000, 16, None) == E_SHAPE          # window past cin_total
    assert lib.vdmi_rows_wgrad(16, 32, 16, 64, 16, 64, 64, 64, 0, 1000, 16, None) == E_SHAPE          # row pitch below cin
    assert lib.vdmi_conv_wgrad_rows_workspace_bytes(128, 128, 16, 128, 128) == 128 * 4 * 9 * 4096 * 4
    assert lib.vdmi_conv_wgrad_rows_workspace_bytes(64, 64, 1, 4, 4) == 9 * 4096 * 4                  # one unit: one workgroup
    assert lib.vdmi_conv_wgrad_rows_workspace_bytes(60, 64, 1, 4, 4) == 0                             # cin not a multiple of 8
    assert lib.vdmi_conv_wgrad_rows(None, 64, 16, 64, 16, 64, 64, 64, 0, 2, 8, 8, 16, None) == E_ARG
    assert lib.vdmi_conv_wgrad_rows(16, 64, 16, 64, 16, 64, 64, 64, 8, 2, 8, 8, 16, None) == E_SHAPE   # window past cin_total
    assert lib.vdmi_conv_wgrad_rows(16, 32, 16, 64, 16, 64, 64, 64, 0, 2, 8, 8, 16, None) == E_SHAPE   # row pitch below cin
    assert lib.vdmi_conv_wgrad_rows(16, 64, 16, 64, 16, 64, 64, 64, 0, 2, 8, 8, 8, None) == E_ALIGN
    assert lib.vdmi_strided_wgrad_rows_workspace_bytes(128, 128, 16, 128, 128) == 64 * 2 * 4 * 16 * 2048 * 4
    assert lib.vdmi_strided_wgrad_rows_workspace_bytes(128, 128, 16, 127, 128) == 0                   # odd image
    assert lib.vdmi_strided_wgrad_rows(None, 64, 16, 64, 16, 64, 64, 2, 8, 8, 16, None) == E_ARG
    assert lib.vdmi_strided_wgrad_rows(16, 32, 16, 64, 16, 64, 64, 2, 8, 8, 16, None) == E_SHAPE      # row pitch below cb
    assert lib.vdmi_strided_wgrad_rows(16, 64, 16, 64, 16, 64, 64, 2, 8, 8, 8, None) == E_ALIGN
    assert lib.vdmi_rmsnorm_silu_bwd(None, 64, 16, 64, 16, 64, 1, 10, 64, None, None, None, 0, 8.0, None, None, None,
                                     None, None) == E_ARG
    assert lib.vdmi_rmsnorm_silu_bwd(16, 96, 16, 96, 16, 96, 1, 10, 96, 16, None, None, 0, 9.8, 16, None, None, None,
                                     None) == E_SHAPE                                                   # c not a power of two
    assert lib.vdmi_rmsnorm_silu_bwd(16, 64, 16, 64, 16, 64, 1, 10, 64, 16, 16, None, 64, 8.0, 16, None, None, None,
                                     None) == E_ARG                                                     # scale without shift
    # the form that adds per-workgroup partials in a fixed order: one partial [4][c] per 16 passes of 256 / (c / 8) rows
    assert lib.vdmi_rmsnorm_silu_bwd_workspace_bytes(2, 1000, 64) == 2 * ((1000 + 511) // 512) * 4 * 64 * 4
    assert lib.vdmi_rmsnorm_silu_bwd_workspace_bytes(1, 1, 2048) == 4 * 2048 * 4
    assert lib.vdmi_rmsnorm_silu_bwd_workspace_bytes(1, 10, 96) == 0
    assert lib.vdmi_rmsnorm_silu_bwd_workspace_bytes(0, 10, 64) == 0
    assert lib.vdmi_rmsnorm_silu_bwd_ws(16, 64, 16, 64, 16, 64, 1, 10, 64, 16, None, None, 0, 8.0, 16, None, None, None,
                                        None, None) == E_ARG                                            # no workspace
    assert lib.vdmi_rmsnorm_silu_bwd_ws(16, 64, 16, 64, 16, 64, 1, 10, 64, 16, None, None, 0, 8.0, 16, None, None, None,
                                        8, None) == E_ALIGN
    assert lib.vdmi_rmsnorm_silu_bwd_ws(16, 96, 16, 96, 16, 96, 1, 10, 96, 16, None, None, 0, 9.8, 16, None, None, None,
                                        16, None) == E_SHAPE
    assert lib.vdmi_colsum(None, 64, 10, 64, None, None) == E_ARG
    assert lib.vdmi_colsum(16, 64, 10, 4096, 16, None) == E_SHAPE
    assert lib.vdmi_colsum_workspace_bytes(5000, 64) == ((5000 + 511) // 512) * 64 * 4
    assert lib.vdmi_colsum_workspace_bytes(10, 4096) == 0
    assert lib.vdmi_colsum_ws(16, 64, 10, 64, 16, None, None) == E_ARG
    assert lib.vdmi_colsum_ws(16, 64, 10, 64, 16, 8, None) == E_ALIGN
    assert lib.vdmi_colsum_ws(16, 64, 10, 4096, 16, 16, None) == E_SHAPE


def test_attention_training_entry_points_reject_bad_arguments():
    """Argument checks of the attention training entry points (every call returns before a launch)."""
    from video_diffusion_amd import _lib
    lib = _lib.lib()
    E_ARG, E_SHAPE, E_ALIGN = -1, -2, -3
    assert lib.vdmi_layernorm_rows(None, 16, 10, 64, 1e-5, None) == E_ARG
    assert lib.vdmi_layernorm_rows(16, 16, 10, 96, 1e-5, None) == E_SHAPE                       # not a power of two
    assert lib.vdmi_layernorm_rows(16, 24, 10, 64, 1e-5, None) == E_ALIGN
    assert lib.vdmi_layernorm_rows_bwd(16, None, None, 16, 10, 64, 1e-5, None) == E_ARG
    assert lib.vdmi_tattn_core_fwd(None, 16, 16, None, None, None, None, 1, 10, 64, 8, 0.17, None) == E_ARG
    assert lib.vdmi_tattn_core_fwd(16, 16, 16, None, None, None, None, 1, 257, 64, 8, 0.17, None) == E_SHAPE   # f > 256
    assert lib.vdmi_tattn_core_fwd(16, 16, 16, None, 16, None, None, 1, 10, 64, 8, 0.17, None) == E_ARG        # cos without sin
    assert lib.vdmi_tattn_core_bwd(16, 16, 16, 16, 16, 16, None, None, None, None, 1, 10, 64, 8, 0.17, None, None) == E_ARG   # dbias, no bias
    assert lib.vdmi_tattn_core_bwd(16, 16, 16, 16, 16, None, None, None, None, None, 1, 300, 64, 8, 0.17, None, None) == E_SHAPE  # f > 256
    # matrix-core form: 22 tiles of 3 sites x 8 heads = 176 units -> 22 workgroups, rounded to a multiple of the heads
    assert lib.vdmi_tattn_core_bwd_workspace_bytes(1, 10, 64, 8) == 24 * 8 * 4096
    assert lib.vdmi_tattn_core_bwd_workspace_bytes(1, 33, 64, 8) == 0                           # vector kernel
    assert lib.vdmi_sla_train_partial_bytes(16, 4096, 4) == 16 * 4 * 4 * (1024 + 64) * 4
    assert lib.vdmi_sla_train_combined_bytes(16, 4) == 16 * 4 * (1024 + 64) * 4
    assert lib.vdmi_sla_train_partial_bytes(16, 4096, 3) == 0                                   # 256 % heads
    assert lib.vdmi_sla_train_reduce(None, 384, 16, 384, 16, None, 16, 2, 100, 4, 0, None) == E_ARG
    assert lib.vdmi_sla_train_reduce(16, 384, 16, 384, 16, None, 16, 2, 100, 4, 1, None) == E_ARG       # mode 1 needs the ctx record
    assert lib.vdmi_sla_train_reduce(16, 64, 16, 384, 16, None, 16, 2, 100, 4, 0, None) == E_SHAPE      # lda < heads * 32
    assert lib.vdmi_sla_train_rows_fwd(16, None, 16, 16, 2, 100, 4, 0.17, None) == E_ARG
    assert lib.vdmi_sla_train_rows_bwd(16, 16, 16, 16, None, 16, 2, 100, 4, 0.17, None) == E_ARG
    assert lib.vdmi_sla_train_rows_fwd(16, 16, 16, 16, 2, 100, 3, 0.17, None) == E_SHAPE


def test_attention_core_fit_query():
    """vdmi_tattn_core_fits (host): with 8 heads the backward core with a bias gradient holds token axes up to about
    62 positions (its [heads][f][f] table and four tiles must fit 160 KiB of LDS); nothing beyond 256."""
    lib = _lib.lib()
    assert lib.vdmi_tattn_core_fits(1, 10, 8, 1) == 1 and lib.vdmi_tattn_core_fits(1, 32, 8, 1) == 1
    assert lib.vdmi_tattn_core_fits(1, 64, 8, 1) == 0 and lib.vdmi_tattn_core_fits(0, 64, 8, 0) == 1
    assert lib.vdmi_tattn_core_fits(0, 257, 8, 0) == 0 and lib.vdmi_tattn_core_fits(1, 0, 8, 0) == 0


def test_batched_weight_packer_rejects_bad_arguments():
    """vdmi_pack_taps_batch checks its job table on the host before anything is copied or launched."""
    import ctypes as C
    from video_diffusion_amd import _lib
    lib = _lib.lib()
    E_ARG, E_SHAPE, E_ALIGN = -1, -2, -3
    assert lib.vdmi_pack_taps_batch_table_bytes(3) == 3 * 80 and lib.vdmi_pack_taps_batch_table_bytes(0) == 0
    ok = _lib.PackJob(16, None, None, 16, 64, 64, 64, 0, 9, 64, 0)
    arr = (_lib.PackJob * 1)(ok)
    assert lib.vdmi_pack_taps_batch(None, 1, 16, None) == E_ARG
    assert lib.vdmi_pack_taps_batch(C.cast(arr, C.c_void_p), 0, 16, None) == E_ARG
    assert lib.vdmi_pack_taps_batch(C.cast(arr, C.c_void_p), 1, None, None) == E_ARG
    for field, value, status in (("w", None, E_ARG), ("out", None, E_ARG), ("out", 24, E_ALIGN), ("npad", 48, E_SHAPE),
                                 ("bkc", 48, E_SHAPE), ("ntaps", 0, E_SHAPE), ("cin0", 0, E_SHAPE)):
        bad = _lib.PackJob(16, None, None, 16, 64, 64, 64, 0, 9, 64, 0)
        setattr(bad, field, value)
        arr = (_lib.PackJob * 2)(ok, bad)
        assert lib.vdmi_pack_taps_batch(C.cast(arr, C.c_void_p), 2, 16, None) == status, field
