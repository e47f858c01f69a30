"""GPU tests of the backward kernels (csrc/bwd.hip) and of the hip-resnet training backend (train_hip.py).

The checker is PyTorch autograd in fp32 over the same bf16-rounded operands (kernel tests) and autograd.py, the
differentiable torch restatement of the reference's forward that tests/test_training.py pins to the reference's golden
outputs (block and model tests).  Bars: fp32 sums of exact bf16 products 1e-3; anything that passes through bf16
activations or bf16 activation gradients 2e-2 to 5e-2 relative L2, stated per assertion."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import record, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bf(t):
    return t.to(torch.bfloat16)


def test_pack_taps_device_matches_host():
    from video_diffusion_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(1)
    for n, npad, cin0, cin1, ntaps, bkc in [(64, 64, 64, 0, 9, 64), (96, 128, 64, 40, 9, 64), (128, 128, 72, 0, 1, 64),
                                            (32, 32, 32, 32, 4, 32)]:
        w = torch.randn(n, cin0 + cin1, ntaps, generator=g)
        elems = lib.vdmi_pack_taps_elems(n, npad, cin0, cin1, ntaps, bkc)
        host = torch.empty(elems, dtype=torch.int16)
        _lib.check(lib.vdmi_pack_taps(w.data_ptr(), n, npad, cin0, cin1, ntaps, bkc, host.data_ptr(), None))
        wd = w.to(DEV)
        dev = torch.empty(elems, dtype=torch.int16, device=DEV)
        _lib.check(lib.vdmi_pack_taps_device(wd.data_ptr(), n, npad, cin0, cin1, ntaps, bkc, 0, dev.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(dev.cpu(), host)
        if cin1 == 0:
            # transposed: the device packs from the forward weight [k][n][taps]; the host packs the explicit transpose
            fwd = torch.randn(cin0, n, ntaps, generator=g)         # [k][n][taps]: what the forward convolution holds
            host_t = torch.empty(elems, dtype=torch.int16)
            explicit = fwd.permute(1, 0, 2).flip(-1).contiguous()  # [n][k][reversed taps]
            _lib.check(lib.vdmi_pack_taps(explicit.data_ptr(), n, npad, cin0, 0, ntaps, bkc, host_t.data_ptr(), None))
            fd = fwd.to(DEV)
            dev_t = torch.empty(elems, dtype=torch.int16, device=DEV)
            _lib.check(lib.vdmi_pack_taps_device(fd.data_ptr(), n, npad, cin0, 0, ntaps, bkc, 1, dev_t.data_ptr(), None))
            torch.cuda.synchronize()
            assert torch.equal(dev_t.cpu(), host_t)


def test_rows_transpose():
    from video_diffusion_amd import train_hip as T
    g = torch.Generator().manual_seed(2)
    for nf, hw, c in [(10, 64, 64), (20, 35, 72), (16, 9, 8)]:
        rows = _bf(torch.randn(nf * hw, c, generator=g)).to(DEV)
        out = T.rows_transposed(rows, nf, hw)
        nfb, cb = (nf + 15) // 16, (c + 31) // 32
        ref = torch.zeros(cb * 32, hw, nfb * 16, dtype=torch.bfloat16, device=DEV)
        ref[:c, :, :nf] = rows.view(nf, hw, c).permute(2, 1, 0)
        # [frame block][channel block][pixel][channel in block][frame in block]
        ref = ref.view(cb, 32, hw, nfb, 16).permute(3, 0, 2, 1, 4).contiguous()
        assert torch.equal(out.view(nfb, cb, hw, 32, 16), ref)


@pytest.mark.parametrize("nf,h,w,cin,cout,ntaps", [(10, 8, 8, 64, 64, 9), (20, 12, 10, 72, 40, 9), (3, 16, 16, 128, 64, 9),
                                                   (10, 8, 8, 128, 64, 1), (33, 5, 7, 8, 32, 9)])
def test_conv_wgrad_vs_torch(nf, h, w, cin, cout, ntaps):
    from video_diffusion_amd import train_hip as T
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(nf * h * w, cin, generator=g)).to(DEV)
    dy = _bf(torch.randn(nf * h * w, cout, generator=g)).to(DEV)
    dw = T.conv_wgrad(T.rows_transposed(x, nf, h * w), T.rows_transposed(dy, nf, h * w), cin, cout, nf, h, w, ntaps)
    k = 3 if ntaps == 9 else 1
    xi = x.float().view(nf, h, w, cin).permute(0, 3, 1, 2)
    gi = dy.float().view(nf, h, w, cout).permute(0, 3, 1, 2)
    wt = torch.zeros(cout, cin, k, k, device=DEV, requires_grad=True)
    F.conv2d(xi.double(), wt.double(), padding=k // 2).backward(gi.double())
    err = record(f"wgrad.nf{nf}.{h}x{w}.{cin}to{cout}.taps{ntaps}", rel_err(dw.view(cout, cin, k, k).cpu(), wt.grad.cpu()), 1e-3)
    assert err < 1e-3


@pytest.mark.parametrize("nf,h,w,cin,cout", [(10, 8, 8, 64, 64), (20, 12, 10, 72, 40), (3, 16, 16, 128, 64), (33, 5, 7, 8, 32),
                                             (2, 37, 45, 64, 136), (3, 64, 64, 128, 128), (16, 16, 16, 256, 192), (1, 1, 1, 8, 8),
                                             (2, 128, 128, 64, 64), (5, 4, 33, 24, 16)])
def test_conv_wgrad_rows_vs_torch(nf, h, w, cin, cout):
    """vdmi_conv_wgrad_rows (no transposed copies): both tile shapes (w <= 16 and wider), ragged widths / heights / channel
    counts, more workgroups than units, and the transposed-operand form as a second reference."""
    from video_diffusion_amd import train_hip as T
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(nf * h * w, cin, generator=g)).to(DEV)
    dy = _bf(torch.randn(nf * h * w, cout, generator=g)).to(DEV)
    assert T._WGRAD3_ROWS
    dw = T.conv3x3_wgrad(x, dy, nf, h, w)
    xi = x.float().view(nf, h, w, cin).permute(0, 3, 1, 2)
    gi = dy.float().view(nf, h, w, cout).permute(0, 3, 1, 2)
    wt = torch.zeros(cout, cin, 3, 3, device=DEV, requires_grad=True)
    F.conv2d(xi.double(), wt.double(), padding=1).backward(gi.double())
    err = record(f"wgrad_rows.nf{nf}.{h}x{w}.{cin}to{cout}", rel_err(dw.view(cout, cin, 3, 3).cpu(), wt.grad.cpu()), 1e-3)
    assert err < 1e-3
    old = T.conv_wgrad(T.rows_transposed(x, nf, h * w), T.rows_transposed(dy, nf, h * w), cin, cout, nf, h, w, 9)
    assert rel_err(dw.cpu(), old.cpu()) < 1e-4


@pytest.mark.parametrize("nf,h,w,cb,cs", [(3, 16, 16, 64, 64), (2, 32, 32, 128, 64), (5, 8, 12, 72, 40), (1, 2, 2, 8, 8),
                                          (2, 64, 64, 64, 128), (16, 26, 38, 32, 96), (3, 128, 128, 64, 64)])
def test_strided_wgrad_rows_vs_torch(nf, h, w, cb, cs, monkeypatch):
    """vdmi_strided_wgrad_rows (weight gradient of Downsample's (1,4,4) / stride 2 / pad 1 convolution, vdp.py:377-378, from
    the rows as they lie): against torch's conv2d weight gradient and against the four-one-tap form; ragged tiles (heights
    that are no multiple of 6 small rows, widths no multiple of 16 small pixels, channel counts off the tile sizes)."""
    from video_diffusion_amd import train_hip as T
    g = torch.Generator().manual_seed(11)
    big = _bf(torch.randn(nf * h * w, cb, generator=g)).to(DEV)
    small = _bf(torch.randn(nf * (h // 2) * (w // 2), cs, generator=g)).to(DEV)
    assert T._STRIDED_WGRAD_ROWS
    dw = T.strided_wgrad(big, small, nf, h, w)
    xi = big.float().view(nf, h, w, cb).permute(0, 3, 1, 2)
    gi = small.float().view(nf, h // 2, w // 2, cs).permute(0, 3, 1, 2)
    wt = torch.zeros(cs, cb, 4, 4, device=DEV, requires_grad=True)
    F.conv2d(xi.double(), wt.double(), stride=2, padding=1).backward(gi.double())
    err = record(f"strided_wgrad.nf{nf}.{h}x{w}.{cb}to{cs}", rel_err(dw.view(cs, cb, 4, 4).cpu(), wt.grad.cpu()), 1e-3)
    assert err < 1e-3
    assert torch.equal(dw, T.strided_wgrad(big, small, nf, h, w))
    monkeypatch.setattr(T, "_STRIDED_WGRAD_ROWS", False)
    assert rel_err(dw.cpu(), T.strided_wgrad(big, small, nf, h, w).cpu()) < 1e-4


@pytest.mark.parametrize("b,f,h,w,cin,cout,k", [(1, 5, 16, 16, 3, 64, 7), (2, 10, 12, 9, 1, 32, 7), (1, 3, 8, 8, 4, 128, 3)])
def test_init_conv_weight_gradient_vs_torch(b, f, h, w, cin, cout, k):
    """init_conv (vdp.py:644) through InitConvFn: torch's forward, weight / bias gradient from vdmi_conv_wgrad_gather."""
    from video_diffusion_amd import train_hip as T
    torch.manual_seed(7)
    conv = torch.nn.Conv3d(cin, cout, (1, k, k), padding=(0, k // 2, k // 2)).to(DEV)
    x = torch.randn(b, cin, f, h, w, device=DEV)
    dout = torch.randn(b, cout, f, h, w, device=DEV)
    conv(x).backward(dout)
    ref_w, ref_b = conv.weight.grad.clone(), conv.bias.grad.clone()
    conv.weight.grad = conv.bias.grad = None
    xr = x.clone().requires_grad_(True)
    T.init_conv(conv, xr).backward(dout)
    ref_x = torch.nn.grad.conv3d_input(x.shape, conv.weight, dout, padding=(0, k // 2, k // 2))
    tag = f"initconv.{cin}to{cout}.k{k}"
    assert record(tag + ".dw", rel_err(conv.weight.grad, ref_w), 1e-2) < 1e-2        # operands rounded to bf16
    assert record(tag + ".db", rel_err(conv.bias.grad, ref_b), 1e-5) < 1e-5
    assert rel_err(xr.grad, ref_x) < 1e-5


def test_pointwise_conv_vs_torch():
    """final_conv[1] (vdp.py:728) through PointwiseConvFn against torch's Conv3d, forward and all three gradients."""
    from video_diffusion_amd import train_hip as T
    torch.manual_seed(8)
    conv = torch.nn.Conv3d(64, 3, 1).to(DEV)
    x = torch.randn(2, 64, 3, 16, 16, device=DEV)
    dout = torch.randn(2, 3, 3, 16, 16, device=DEV)
    xa = x.clone().requires_grad_(True)
    ya = conv(xa)
    ya.backward(dout)
    ref = (ya.detach(), xa.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    conv.weight.grad = conv.bias.grad = None
    xb = x.clone().requires_grad_(True)
    yb = T.pointwise_conv(conv, xb)
    yb.backward(dout)
    assert rel_err(yb.detach(), ref[0]) < 1e-5 and rel_err(xb.grad, ref[1]) < 1e-5
    assert record("pointwise.dw", rel_err(conv.weight.grad, ref[2]), 1e-2) < 1e-2          # operands rounded to bf16
    assert rel_err(conv.bias.grad, ref[3]) < 1e-5


def _norm_ref(y, gamma, scale, shift):
    n = F.normalize(y, dim=1) * (y.shape[1] ** 0.5) * gamma
    if scale is not None:
        n = n * (scale + 1) + shift
    return F.silu(n)


@pytest.mark.parametrize("form", ["fixed_order", "atomic"])
@pytest.mark.parametrize("c,batch,rpb,with_ss", [(64, 2, 1000, True), (256, 3, 77, True), (1024, 2, 130, False), (128, 1, 5000, True),
                                                 (2048, 2, 37, True), (512, 1, 1, False), (128, 4, 16 * 24 * 24, True)])
def test_rmsnorm_silu_bwd_vs_torch(c, batch, rpb, with_ss, form, monkeypatch):
    from video_diffusion_amd import train_hip as T
    monkeypatch.setattr(T, "_NORM_BWD_FIXED_ORDER", form == "fixed_order")
    g = torch.Generator().manual_seed(4)
    y = _bf(torch.randn(batch * rpb, c, generator=g) * 1.5).to(DEV)
    dh = _bf(torch.randn(batch * rpb, c, generator=g)).to(DEV)
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(DEV)
    scale = (0.3 * torch.randn(batch, c, generator=g)).to(DEV) if with_ss else None
    shift = (0.3 * torch.randn(batch, c, generator=g)).to(DEV) if with_ss else None
    dy, dgamma, dbias, dscale, dshift = T.rmsnorm_silu_bwd_rows(y, dh, gamma, scale, shift, batch, rpb)
    yr = y.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    sr = scale.double().requires_grad_(True) if with_ss else None
    hr = shift.double().requires_grad_(True) if with_ss else None
    rows_scale = sr.repeat_interleave(rpb, dim=0) if with_ss else None
    rows_shift = hr.repeat_interleave(rpb, dim=0) if with_ss else None
    _norm_ref(yr, gr, rows_scale, rows_shift).backward(dh.double())
    if form == "fixed_order":
        # no float atomics in this form: a second call gives the same bits
        again = T.rmsnorm_silu_bwd_rows(y, dh, gamma, scale, shift, batch, rpb)
        for first, second in zip((dy, dgamma, dbias, dscale, dshift), again):
            assert first is None or torch.equal(first, second)
    tag = f"normbwd.{form}.c{c}"
    assert record(tag + ".dy", rel_err(dy.float().cpu(), yr.grad.cpu()), 1e-2) < 1e-2          # bf16 output
    assert record(tag + ".dgamma", rel_err(dgamma.cpu(), gr.grad.cpu()), 2e-3) < 2e-3
    assert record(tag + ".dbias", rel_err(dbias.cpu(), yr.grad.sum(0).cpu()), 2e-3) < 2e-3
    if with_ss:
        assert record(tag + ".dscale", rel_err(dscale.cpu(), sr.grad.cpu()), 2e-3) < 2e-3
        assert record(tag + ".dshift", rel_err(dshift.cpu(), hr.grad.cpu()), 2e-3) < 2e-3


@pytest.mark.parametrize("form", ["fixed_order", "atomic"])
def test_colsum(form, monkeypatch):
    from video_diffusion_amd import train_hip as T
    monkeypatch.setattr(T, "_NORM_BWD_FIXED_ORDER", form == "fixed_order")
    g = torch.Generator().manual_seed(5)
    for rows, c in [(5000, 64), (333, 1024), (70, 8), (262144, 128), (1000, 72), (3, 2048)]:
        x = _bf(torch.randn(rows, c, generator=g)).to(DEV)
        out = T.colsum_rows(x)
        assert rel_err(out.cpu(), x.double().sum(0).cpu()) < 1e-4
        if form == "fixed_order":
            assert torch.equal(out, T.colsum_rows(x))


def _rand_block(cin, cout, temb_dim, seed):
    from video_diffusion_amd.modules import ResnetBlock
    torch.manual_seed(seed)
    rb = ResnetBlock(cin, cout, time_emb_dim=temb_dim)
    with torch.no_grad():
        for p in rb.parameters():
            if p.dim() == 4 and p.shape[1:] == (1, 1, 1):          # RMSNorm gammas: around one
                p.copy_(1 + 0.2 * torch.randn_like(p))
    return rb.to(DEV)


@pytest.mark.parametrize("cin,cout,temb", [(64, 128, 32), (64, 64, None), (256, 256, 48)])
def test_resnet_block_forward_backward_vs_torch_autograd(cin, cout, temb):
    """One ResnetBlock (vdp.py:429-452): output and every gradient against the torch ops of autograd.py."""
    from video_diffusion_amd import autograd as A, train_hip as T
    rb = _rand_block(cin, cout, temb, 11)
    g = torch.Generator().manual_seed(6)
    b, f, h, w = 2, 3, 16, 16
    x = torch.randn(b, cin, f, h, w, generator=g).to(DEV)
    te = torch.randn(b, temb, generator=g).to(DEV) if temb else None
    dout = torch.randn(b, cout, f, h, w, generator=g).to(DEV)

    def run(fn):
        xr = x.clone().requires_grad_(True)
        ter = te.clone().requires_grad_(True) if te is not None else None
        for p in rb.parameters():
            p.grad = None
        out = fn(rb, xr, ter)
        out.backward(dout)
        grads = {n: p.grad.clone() for n, p in rb.named_parameters()}
        grads["x"] = xr.grad.clone()
        if ter is not None:
            grads["temb"] = ter.grad.clone()
        return out.detach(), grads

    out_ref, g_ref = run(A._resnet)
    out_hip, g_hip = run(T.resnet_block)
    with torch.no_grad():       # the module called on its own, as a user of the reference's ResnetBlock would
        assert torch.equal(rb(x, te) if te is not None else rb(x), out_hip)
    tag = f"resnet_bwd.{cin}to{cout}"
    assert record(tag + ".out", rel_err(out_hip, out_ref), 2e-2) < 2e-2
    assert set(g_hip) == set(g_ref)
    for name in sorted(g_ref):
        assert g_hip[name].shape == g_ref[name].shape
        assert record(f"{tag}.grad.{name}", rel_err(g_hip[name], g_ref[name]), 3e-2) < 3e-2, name


def test_unet_backward_hip_resnet_vs_torch_autograd():
    """loss.backward() of a whole Unet3D with the ResnetBlocks on the HIP kernels against the all-torch backend
    (vdp.py:1146-1161 loss, :1451 backward): loss, input gradient and all parameter gradients."""
    from tests.helpers import inp, make_unet
    m = make_unet(salt=2, dim=64, dim_mults=(1, 2), channels=3).to(DEV)
    x = inp("bwd.x", (2, 3, 4, 16, 16)).to(DEV)
    noise = inp("bwd.noise", (2, 3, 4, 16, 16)).to(DEV)
    t = torch.tensor([17, 203], device=DEV)

    def run(backend):
        m.train_backend = backend
        for p in m.parameters():
            p.grad = None
        xr = x.clone().requires_grad_(True)
        loss = F.l1_loss(m(xr, t), noise)
        loss.backward()
        m.train_backend = None
        return loss.detach(), xr.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    l_ref, gx_ref, g_ref = run("torch-autograd")
    l_hip, gx_hip, g_hip = run("hip-resnet")
    assert set(g_ref) == set(g_hip)
    assert record("unet_bwd.loss", abs(float(l_hip - l_ref)) / float(l_ref), 2e-2) < 2e-2
    # ... and with the temporal attention layers on the HIP kernels too
    l_all, gx_all, g_all = run("hip")
    assert set(g_ref) == set(g_all)
    assert record("unet_bwd_hip.loss", abs(float(l_all - l_ref)) / float(l_ref), 2e-2) < 2e-2
    cat_ = lambda d: torch.cat([d[n].flatten() for n in sorted(d)])
    assert record("unet_bwd_hip.all_param_grads", rel_err(cat_(g_all), cat_(g_ref)), 6e-2) < 6e-2
    assert record("unet_bwd_hip.input_grad", rel_err(gx_all, gx_ref), 1.2e-1) < 1.2e-1
    # the loss is an L1: its gradient is a sign pattern, which flips wherever the bf16 forward moves a residual across
    # zero; the parameter gradients (sums over all positions) are the stable quantity
    cat = lambda d: torch.cat([d[n].flatten() for n in sorted(d)])
    assert record("unet_bwd.all_param_grads", rel_err(cat(g_hip), cat(g_ref)), 5e-2) < 5e-2
    assert record("unet_bwd.input_grad", rel_err(gx_hip, gx_ref), 1e-1) < 1e-1
    worst = max((rel_err(g_hip[n], g_ref[n]), n) for n in g_ref if g_ref[n].norm() > 1e-6 * cat(g_ref).norm())
    record("unet_bwd.worst_param_grad", worst[0], None)


@pytest.mark.parametrize("c,rows", [(64, 1000), (256, 333), (1024, 70)])
def test_layernorm_rows_forward_backward_vs_torch(c, rows):
    from video_diffusion_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(9)
    x = _bf(torch.randn(rows, c, generator=g) * 2 + 0.5).to(DEV)
    dxh = _bf(torch.randn(rows, c, generator=g)).to(DEV)
    add = _bf(torch.randn(rows, c, generator=g)).to(DEV)
    xh, dx = torch.empty_like(x), torch.empty_like(x)
    _lib.check(lib.vdmi_layernorm_rows(x.data_ptr(), xh.data_ptr(), rows, c, 1e-5, None))
    _lib.check(lib.vdmi_layernorm_rows_bwd(x.data_ptr(), dxh.data_ptr(), add.data_ptr(), dx.data_ptr(), rows, c, 1e-5, None))
    xr = x.double().requires_grad_(True)
    ref = (xr - xr.mean(1, keepdim=True)) / (xr.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()
    ref.backward(dxh.double())
    assert record(f"layernorm.c{c}.fwd", rel_err(xh.float().cpu(), ref.detach().cpu()), 5e-3) < 5e-3
    assert record(f"layernorm.c{c}.bwd", rel_err(dx.float().cpu(), (xr.grad + add.double()).cpu()), 5e-3) < 5e-3


def _attn_core_ref(qkv, bias, cos, sin, b, f, hw, heads, scale, focus=None):
    """Attention over frames on raw q, k, v rows (vdp.py:560-606), double precision torch ops; focus: bool [b], the
    focus-present mask of vdp.py:590-600."""
    hid = heads * 32
    t = qkv.view(b, f, hw, 3, heads, 32).permute(3, 0, 2, 4, 1, 5)            # [3][b][hw][heads][f][32]
    q, k, v = t[0] * scale, t[1], t[2]

    def rot(u):
        if cos is None:
            return u
        c = torch.cat((cos, cos), -1).double()
        s = torch.cat((sin, sin), -1).double()
        u1, u2 = u.chunk(2, dim=-1)
        return u * c + torch.cat((-u2, u1), -1) * s
    sim = rot(q) @ rot(k).transpose(-1, -2)
    if bias is not None:
        sim = sim + bias
    if focus is not None:
        keep = torch.eye(f, device=sim.device, dtype=torch.bool).view(1, 1, 1, f, f) | ~focus.view(-1, 1, 1, 1, 1)
        sim = sim.masked_fill(~keep, -torch.finfo(sim.dtype).max)
    out = sim.softmax(-1) @ v                                                  # [b][hw][heads][f][32]
    return out.permute(0, 3, 1, 2, 4).reshape(b * f * hw, hid)


@pytest.mark.parametrize("form", ["mfma", "vector"])
@pytest.mark.parametrize("b,f,hw,heads,rotary,with_bias,focus", [
    (2, 10, 37, 8, True, True, None), (1, 16, 64, 8, True, True, None), (3, 5, 20, 4, False, False, None),
    (1, 32, 9, 8, True, True, None), (3, 256, 1, 8, False, False, None), (2, 64, 1, 8, False, False, None),
    (3, 10, 37, 8, True, True, (True, False, True)), (2, 6, 16, 4, False, True, (True, True)),
    (2, 7, 11, 3, True, True, None), (1, 16, 1031, 8, True, True, None), (2, 17, 5, 8, True, False, None),
    (1, 1, 40, 8, False, False, None)])
def test_tattn_core_forward_backward_vs_torch(b, f, hw, heads, rotary, with_bias, focus, form):
    """vdmi_tattn_core_fwd / _bwd against double-precision torch ops, on the matrix-core form (frame axes up to 32: tiles
    of 32 / f sites, also with a ragged last tile, 3 heads, one frame) and on the vector kernels (every length)."""
    from video_diffusion_amd import _lib
    lib = _lib.lib()
    if form == "mfma" and f > 32:
        pytest.skip("frame axes beyond 32 run on the vector kernels only")
    lib.vdmi_tattn_core_mfma_enable(1 if form == "mfma" else 0)
    try:
        _tattn_core_case(lib, b, f, hw, heads, rotary, with_bias, focus, form)
    finally:
        lib.vdmi_tattn_core_mfma_enable(1)


def _tattn_core_case(lib, b, f, hw, heads, rotary, with_bias, focus, form):
    from video_diffusion_amd import _lib
    g = torch.Generator().manual_seed(10)
    rows, hid = b * f * hw, heads * 32
    qkv = _bf(torch.randn(rows, 3 * hid, generator=g)).to(DEV)
    dout = _bf(torch.randn(rows, hid, generator=g)).to(DEV)
    bias = (0.5 * torch.randn(heads, f, f, generator=g)).to(DEV) if with_bias else None
    ang = torch.arange(f).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))[None, :]
    cos, sin = (ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()) if rotary else (None, None)
    scale = 32 ** -0.5
    ptr = lambda t: None if t is None else t.data_ptr()
    o = torch.empty(rows, hid, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(rows, heads, device=DEV)
    foc = None if focus is None else torch.tensor(focus, dtype=torch.uint8, device=DEV)
    _lib.check(lib.vdmi_tattn_core_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), ptr(bias), ptr(cos), ptr(sin), ptr(foc),
                                       b, f, hw, heads, scale, None))
    dqkv = torch.empty_like(qkv)
    dbias = torch.zeros_like(bias) if with_bias else None
    ws_bytes = lib.vdmi_tattn_core_bwd_workspace_bytes(b, f, hw, heads)
    assert (ws_bytes > 0) == (form == "mfma")
    ws = torch.empty(max(ws_bytes, 16), device=DEV, dtype=torch.uint8).fill_(0x7f)      # contents must not matter
    _lib.check(lib.vdmi_tattn_core_bwd(qkv.data_ptr(), o.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                       ptr(dbias), ptr(bias), ptr(cos), ptr(sin), ptr(foc), b, f, hw, heads, scale,
                                       ws.data_ptr(), None))
    qr = qkv.double().requires_grad_(True)
    br = bias.double().requires_grad_(True) if with_bias else None
    ref = _attn_core_ref(qr, br, cos, sin, b, f, hw, heads, scale, None if foc is None else foc.bool())
    ref.backward(dout.double())
    tag = f"tattn_core.{form}.f{f}.hw{hw}.h{heads}" + ("" if focus is None else ".focus")
    assert record(tag + ".o", rel_err(o.float().cpu(), ref.detach().cpu()), 5e-3) < 5e-3
    assert record(tag + ".dqkv", rel_err(dqkv.float().cpu(), qr.grad.cpu()), 1e-2) < 1e-2
    if with_bias:
        assert record(tag + ".dbias", rel_err(dbias.cpu(), br.grad.cpu()), 5e-3) < 5e-3


@pytest.mark.parametrize("c,f,h,w", [(64, 10, 12, 12), (256, 4, 8, 8)])
def test_temporal_attention_layer_forward_backward_vs_torch_autograd(c, f, h, w):
    """The whole layer (vdp.py:636) through TemporalAttnFn against the torch ops of autograd.py: output, input gradient
    and the gradients of gamma, to_qkv, to_out and the relative-position bias."""
    from video_diffusion_amd import autograd as A, train_hip as T
    from video_diffusion_amd.modules import Attention, EinopsToAndFrom, PreNorm, Residual, RotaryEmbedding
    torch.manual_seed(12)
    rot = RotaryEmbedding(32)
    res = Residual(PreNorm(c, EinopsToAndFrom('b c f h w', 'b (h w) f c', Attention(c, heads=8, dim_head=32, rotary_emb=rot)))).to(DEV)
    with torch.no_grad():
        res.fn.norm.gamma.copy_(1 + 0.2 * torch.randn_like(res.fn.norm.gamma))
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, c, f, h, w, generator=g).to(DEV)
    dout = torch.randn(2, c, f, h, w, generator=g).to(DEV)
    bias0 = (0.3 * torch.randn(8, f, f, generator=g)).to(DEV)

    def run(hip):
        xr = x.clone().requires_grad_(True)
        br = bias0.clone().requires_grad_(True)
        for p in res.parameters():
            p.grad = None
        out = T.temporal_attention(res, xr, br, A._rot_tables(rot, f, DEV)) if hip else A._temporal(res, xr, br)
        out.backward(dout)
        grads = {n: p.grad.clone() for n, p in res.named_parameters() if p.grad is not None}
        grads["x"], grads["pos_bias"] = xr.grad.clone(), br.grad.clone()
        return out.detach(), grads

    out_ref, g_ref = run(False)
    out_hip, g_hip = run(True)
    with torch.no_grad():       # the module called on its own, with the reference's keyword
        assert torch.equal(res(x, pos_bias=bias0), out_hip)
        assert torch.equal(res(x, pos_bias=bias0, focus_present_mask=torch.zeros(2, dtype=torch.bool, device=DEV)), out_hip)
        fm = torch.tensor([True, False], device=DEV)      # flagged sample: every frame attends to itself only
        out_f = res(x, pos_bias=bias0, focus_present_mask=fm)
        assert record(f"tattn_layer.c{c}.focus_out", rel_err(out_f, A._temporal(res, x, bias0, fm)), 2e-2) < 2e-2
        assert torch.equal(out_f[1], out_hip[1]) and not torch.equal(out_f[0], out_hip[0])
    tag = f"tattn_layer.c{c}"
    assert record(tag + ".out", rel_err(out_hip, out_ref), 2e-2) < 2e-2
    assert set(g_hip) == set(g_ref)
    for name in sorted(g_ref):
        assert record(f"{tag}.grad.{name}", rel_err(g_hip[name], g_ref[name]), 3e-2) < 3e-2, name


@pytest.mark.parametrize("c,f,h,w", [(64, 3, 16, 16), (128, 2, 8, 12)])
def test_resampling_convs_forward_backward_vs_torch(c, f, h, w):
    """Downsample / Upsample (vdp.py:374-378) through DownsampleFn / UpsampleFn against torch's Conv3d /
    ConvTranspose3d: output, input gradient, weight and bias gradients."""
    from video_diffusion_amd import train_hip as T
    from video_diffusion_amd.modules import Downsample, Upsample
    torch.manual_seed(14)
    for kind, mod in (("down", Downsample(c).to(DEV)), ("up", Upsample(c).to(DEV))):
        x = torch.randn(2, c, f, h, w, device=DEV)
        ref_out = mod(x)
        dout = torch.randn_like(ref_out)
        xa = x.clone().requires_grad_(True)
        mod(xa).backward(dout)
        ref = (ref_out.detach(), xa.grad.clone(), mod.weight.grad.clone(), mod.bias.grad.clone())
        mod.weight.grad = mod.bias.grad = None
        xb = x.clone().requires_grad_(True)
        out = T.resample(mod, xb)
        out.backward(dout)
        tag = f"resample.{kind}.c{c}"
        assert record(tag + ".out", rel_err(out.detach(), ref[0]), 1e-2) < 1e-2
        assert record(tag + ".dx", rel_err(xb.grad, ref[1]), 1e-2) < 1e-2
        assert record(tag + ".dw", rel_err(mod.weight.grad, ref[2]), 1e-2) < 1e-2
        assert record(tag + ".db", rel_err(mod.bias.grad, ref[3]), 1e-2) < 1e-2


@pytest.mark.parametrize("form", ["mfma", "vector"])
@pytest.mark.parametrize("frames,n,heads", [(3, 100, 4), (2, 2500, 4), (5, 64, 8), (16, 4096, 4), (7, 33, 8), (1, 1, 2)])
def test_sla_core_forward_backward_vs_torch(frames, n, heads, form):
    """The linear-attention core (vdp.py:467-477) on raw q, k, v rows: reduction partials + row kernels against torch, with
    the row kernels on the matrix cores (tiles of 32 rows, ragged last tiles, runs of tiles that cross frames) and on
    the vector kernels."""
    from video_diffusion_amd import _lib
    lib = _lib.lib()
    lib.vdmi_sla_rows_mfma_enable(1 if form == "mfma" else 0)
    try:
        _sla_core_case(lib, frames, n, heads, form)
    finally:
        lib.vdmi_sla_rows_mfma_enable(1)


def _sla_core_case(lib, frames, n, heads, form):
    from video_diffusion_amd import _lib
    g = torch.Generator().manual_seed(15)
    rows, hid, scale = frames * n, heads * 32, 32 ** -0.5
    qkv = _bf(torch.randn(rows, 3 * hid, generator=g) * 1.5).to(DEV)
    dout = _bf(torch.randn(rows, hid, generator=g)).to(DEV)
    from video_diffusion_amd import train_hip as T
    base = qkv.data_ptr()
    c_ctx = T._sla_reduce(base + 2 * hid, 3 * hid, base + 4 * hid, 3 * hid, None, frames, n, heads, 0, DEV)
    out, qt, dqkv = (torch.empty(rows, hid, device=DEV, dtype=torch.bfloat16), torch.empty(rows, hid, device=DEV, dtype=torch.bfloat16),
                     torch.empty_like(qkv))
    _lib.check(lib.vdmi_sla_train_rows_fwd(qkv.data_ptr(), c_ctx.data_ptr(), out.data_ptr(), qt.data_ptr(), frames, n, heads, scale, None))
    c_dctx = T._sla_reduce(qt.data_ptr(), hid, dout.data_ptr(), hid, c_ctx, frames, n, heads, 1, DEV)
    _lib.check(lib.vdmi_sla_train_rows_bwd(qkv.data_ptr(), qt.data_ptr(), dout.data_ptr(), c_ctx.data_ptr(), c_dctx.data_ptr(),
                                           dqkv.data_ptr(), frames, n, heads, scale, None))
    qr = qkv.double().requires_grad_(True)
    t = qr.view(frames, n, 3, heads, 32).permute(2, 0, 3, 4, 1)            # [3][frames][heads][32][n]
    q, k, v = t[0].softmax(dim=-2) * scale, t[1].softmax(dim=-1), t[2]
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    ref = torch.einsum("bhde,bhdn->bhen", ctx, q).permute(0, 3, 1, 2).reshape(rows, hid)
    ref.backward(dout.double())
    tag = f"sla_core.{form}.f{frames}.n{n}.h{heads}"
    assert record(tag + ".out", rel_err(out.float().cpu(), ref.detach().cpu()), 5e-3) < 5e-3
    assert record(tag + ".dqkv", rel_err(dqkv.float().cpu(), qr.grad.cpu()), 1.5e-2) < 1.5e-2


@pytest.mark.parametrize("c,f,h,w", [(64, 3, 16, 16), (256, 2, 8, 8)])
def test_linear_attention_layer_forward_backward_vs_torch_autograd(c, f, h, w):
    """Residual(PreNorm(SpatialLinearAttention)) (vdp.py:454-479) through SlaFn against the torch ops of autograd.py."""
    from video_diffusion_amd import autograd as A, train_hip as T
    from video_diffusion_amd.modules import PreNorm, Residual, SpatialLinearAttention
    torch.manual_seed(16)
    res = Residual(PreNorm(c, SpatialLinearAttention(c, heads=4))).to(DEV)
    with torch.no_grad():
        res.fn.norm.gamma.copy_(1 + 0.2 * torch.randn_like(res.fn.norm.gamma))
    g = torch.Generator().manual_seed(17)
    x = torch.randn(2, c, f, h, w, generator=g).to(DEV)
    dout = torch.randn(2, c, f, h, w, generator=g).to(DEV)

    def run(hip):
        xr = x.clone().requires_grad_(True)
        for p in res.parameters():
            p.grad = None
        out = T.spatial_linear_attention(res, xr) if hip else A._linear_attention(res, xr)
        out.backward(dout)
        grads = {n: p.grad.clone() for n, p in res.named_parameters() if p.grad is not None}
        grads["x"] = xr.grad.clone()
        return out.detach(), grads

    out_ref, g_ref = run(False)
    out_hip, g_hip = run(True)
    with torch.no_grad():
        assert torch.equal(res(x), out_hip)       # the module called on its own
    tag = f"sla_layer.c{c}"
    assert record(tag + ".out", rel_err(out_hip, out_ref), 2e-2) < 2e-2
    assert set(g_hip) == set(g_ref)
    for name in sorted(g_ref):
        assert record(f"{tag}.grad.{name}", rel_err(g_hip[name], g_ref[name]), 3e-2) < 3e-2, name


@pytest.mark.parametrize("c,f,h,w", [(256, 3, 16, 16), (512, 2, 8, 8)])
def test_spatial_attention_layer_forward_backward_vs_torch_autograd(c, f, h, w):
    """The mid-block spatial attention (vdp.py:708-710) through the same Function against the torch ops of autograd.py."""
    from video_diffusion_amd import autograd as A, train_hip as T
    from video_diffusion_amd.modules import Attention, EinopsToAndFrom, PreNorm, Residual
    torch.manual_seed(18)
    res = Residual(PreNorm(c, EinopsToAndFrom('b c f h w', 'b f (h w) c', Attention(c, heads=8, dim_head=32)))).to(DEV)
    with torch.no_grad():
        res.fn.norm.gamma.copy_(1 + 0.2 * torch.randn_like(res.fn.norm.gamma))
    g = torch.Generator().manual_seed(19)
    x = torch.randn(2, c, f, h, w, generator=g).to(DEV)
    dout = torch.randn(2, c, f, h, w, generator=g).to(DEV)

    def run(hip):
        xr = x.clone().requires_grad_(True)
        for p in res.parameters():
            p.grad = None
        out = T.spatial_attention(res, xr) if hip else A._spatial(res, xr)
        out.backward(dout)
        grads = {n: p.grad.clone() for n, p in res.named_parameters() if p.grad is not None}
        grads["x"] = xr.grad.clone()
        return out.detach(), grads

    out_ref, g_ref = run(False)
    out_hip, g_hip = run(True)
    with torch.no_grad():
        assert torch.equal(res(x), out_hip)       # the module called on its own
    tag = f"sattn_layer.c{c}"
    assert record(tag + ".out", rel_err(out_hip, out_ref), 2e-2) < 2e-2
    assert set(g_hip) == set(g_ref)
    for name in sorted(g_ref):
        assert record(f"{tag}.grad.{name}", rel_err(g_hip[name], g_ref[name]), 3e-2) < 3e-2, name


def test_input_gradient_alone_skips_the_weight_gradients():
    """torch.autograd.grad with respect to the input only (reconstruction guidance at sampling time, reference
    version1diffusion.py:707-747, or frozen weights): the same input gradient as a full backward, no parameter gradients."""
    from tests.helpers import inp, make_unet
    m = make_unet(salt=2, dim=64, dim_mults=(1, 2), channels=3).to(DEV)
    m.train_backend = "hip"
    x = inp("bwd.x", (1, 3, 4, 16, 16)).to(DEV)
    t = torch.tensor([17], device=DEV)
    xa = x.clone().requires_grad_(True)
    m(xa, t).square().mean().backward()
    xb = x.clone().requires_grad_(True)
    for p in m.parameters():
        p.grad = None
    (gx,) = torch.autograd.grad(m(xb, t).square().mean(), xb)
    m.train_backend = None
    assert all(p.grad is None for p in m.parameters())
    assert rel_err(gx, xa.grad) < 1e-6


def test_bf16_activation_stream_between_hip_layers():
    """A HIP layer answers a bf16 activation with a bf16 view of its own output rows (no cast, no copy on either side)
    and a float one in float, with the same bits; a whole model with focus-present flags (vdp.py:590-600: the flagged
    samples' temporal attention is masked to the diagonal inside the attention core) trains on the same stream."""
    from tests.helpers import inp, make_unet
    from video_diffusion_amd import train_hip
    m = make_unet(salt=4, dim=64, dim_mults=(1, 2), channels=3).to(DEV)
    rb = m.downs[0][0]
    x = inp("stream.x", (1, 64, 2, 16, 16)).to(DEV)
    temb = inp("stream.t", (1, 512)).to(DEV)
    xb = x.to(torch.bfloat16).requires_grad_(True)
    yb = train_hip.resnet_block(rb, xb, temb)
    yf = train_hip.resnet_block(rb, xb.detach().float(), temb)
    assert yb.dtype == torch.bfloat16 and yf.dtype == torch.float32 and torch.equal(yb.float(), yf)
    rows = train_hip.to_rows(yb)
    assert rows.data_ptr() == yb.data_ptr()                   # the next layer's rows ARE this layer's output
    yb.float().square().mean().backward()
    assert xb.grad.dtype == torch.bfloat16 and bool(torch.isfinite(xb.grad.float()).all())

    xi = inp("stream.img", (2, 3, 4, 16, 16)).to(DEV)
    t = torch.tensor([5, 900], device=DEV)
    focus = torch.tensor([True, False], device=DEV)

    def run(backend):
        m.train_backend = backend
        for p in m.parameters():
            p.grad = None
        out = m(xi, t, focus_present_mask=focus)
        out.square().mean().backward()
        m.train_backend = None
        return out.detach(), torch.cat([p.grad.flatten() for _, p in sorted(m.named_parameters()) if p.grad is not None])

    o_ref, g_ref = run("torch-autograd")
    o_hip, g_hip = run("hip")
    assert o_hip.dtype == torch.float32
    assert record("stream.focus.out", rel_err(o_hip, o_ref), 2e-2) < 2e-2
    assert record("stream.focus.param_grads", rel_err(g_hip, g_ref), 5e-2) < 5e-2


@pytest.mark.parametrize("rows,cin,cout", [(1000, 64, 64), (777, 72, 40), (64, 128, 384), (40000, 256, 768), (5, 8, 8),
                                           (4097, 192, 320), (33, 136, 200), (9000, 128, 128)])
def test_rows_wgrad_vs_torch(rows, cin, cout):
    """vdmi_rows_wgrad (the weight gradient of a 1x1 convolution / Linear from the rows as they lie, through the transposing
    LDS read) against dy^T x in float over the same bf16 operands; ragged row counts and channel widths, on the 64 x 64
    tiles (a width below 128) and on the 128 x 128 tiles (odd numbers of 64-channel tiles, partly empty last tiles)."""
    from tests.helpers import inp
    from video_diffusion_amd import train_hip as T
    x = inp("rw.x", (rows, cin)).to(DEV).to(torch.bfloat16)
    dy = inp("rw.dy", (rows, cout)).to(DEV).to(torch.bfloat16)
    dw = T.rows_wgrad(x, dy)
    ref = dy.float().t() @ x.float()
    assert dw.shape == (cout, cin)
    assert record(f"rows_wgrad[{rows},{cin},{cout}]", rel_err(dw, ref), 1e-5) < 1e-5
    assert torch.equal(dw, T.rows_wgrad(x, dy))           # fixed summation order: repeatable


def test_long_token_axis_attention_matches_the_explicit_softmax():
    """autograd._attention over 1024 or more tokens without bias or mask (SpatioTemporalAttention,
    version1diffusion.py:325-345) goes through torch's fused attention op: same output and gradients as the explicit
    softmax(q k^T) v (taken here by passing an all-zero bias), in float and with the "hip" backend's bf16 operands."""
    from tests.helpers import inp
    from video_diffusion_amd import autograd as A
    from video_diffusion_amd.version1 import SpatioTemporalAttention
    att = SpatioTemporalAttention(64, heads=4).to(DEV)
    n = 1280
    x = inp("sdpa.x", (2, 1, n, 64)).to(DEV)
    zero_bias = torch.zeros(n, n, device=DEV)

    def run(bias, hip):
        A._HIP_ATTN = hip
        try:
            xr = x.clone().requires_grad_(True)
            out = A._attention(att, xr, bias)
            g, = torch.autograd.grad(out.square().mean(), xr)
        finally:
            A._HIP_ATTN = False
        return out.detach(), g

    o_ref, g_ref = run(zero_bias, False)
    o_f, g_f = run(None, False)
    assert record("sdpa.float.out", rel_err(o_f, o_ref), 1e-4) < 1e-4
    assert record("sdpa.float.grad", rel_err(g_f, g_ref), 1e-4) < 1e-4
    o_b, g_b = run(None, True)
    assert record("sdpa.bf16.out", rel_err(o_b, o_ref), 1e-2) < 1e-2
    assert record("sdpa.bf16.grad", rel_err(g_b, g_ref), 2e-2) < 2e-2


def test_fixed_weights_store_follows_the_version_counter():
    """train_hip.fixed_weights / PackStore: a parameter's packed set is found again while its memory and version counter
    are the same, refilled after an in-place update (what an optimizer step or load_state_dict does) - on demand, or
    for every stale set at once when a block is entered - and bit-equal to a set packed on its own; a folded per-channel
    scale is part of the key and of the staleness check."""
    from video_diffusion_amd import train_hip as T
    w = torch.nn.Parameter(torch.randn(64, 64, 1, 3, 3, device=DEV))
    w1 = torch.nn.Parameter(torch.randn(192, 64, device=DEV))
    gam = torch.nn.Parameter(1 + 0.3 * torch.randn(1, 64, 1, 1, 1, device=DEV))
    store = T.PackStore()
    lin = lambda tr: T._linear_pack(w1.detach(), tr, gam)
    with T.fixed_weights(store):
        p0 = T.param_pack(w, False)
        assert T.param_pack(w, False) is p0 and T.param_pack(w, True) is not p0 and len(store) == 2
        d0 = p0.data.clone()
        with torch.no_grad():
            w.mul_(2.0)
        p1 = T.param_pack(w, False)
        assert p1 is p0 and torch.equal(p1.data.float(), d0.float() * 2.0)            # refilled in place
        l0, l1 = lin(False), lin(True)
        folded = (w1.detach() * gam.detach().reshape(1, 64)).contiguous()
        for tr, got in ((False, l0), (True, l1)):
            assert torch.equal(got.data, T.DevPack(folded.reshape(192, 64, 1, 1, 1), tr).data)
        assert len(store) == 4
    assert T._PACKS is None
    assert T.param_pack(w, False) is not p1                 # outside a block nothing is kept
    # the batched refill at the next block: every set is stale after "an optimizer step"
    with torch.no_grad():
        w.add_(0.5)
        w1.mul_(-1.0)
        gam.add_(0.1)
    with T.fixed_weights(store):
        assert not any(store._stale(e) for e in store.entries.values())
        folded = (w1.detach() * gam.detach().reshape(1, 64)).contiguous()
        assert torch.equal(T.param_pack(w, False).data, T.DevPack(w, False).data)
        assert torch.equal(T.param_pack(w, True).data, T.DevPack(w, True).data)
        assert torch.equal(lin(False).data, T.DevPack(folded.reshape(192, 64, 1, 1, 1), False).data)
        assert torch.equal(lin(True).data, T.DevPack(folded.reshape(192, 64, 1, 1, 1), True).data)
    # writes through .data do not move the version counter: invalidate() is the caller's tool for those
    w.data.mul_(3.0)
    with T.fixed_weights(store):
        assert not torch.equal(T.param_pack(w, False).data, T.DevPack(w, False).data)
    store.invalidate()
    with T.fixed_weights(store):
        assert torch.equal(T.param_pack(w, False).data, T.DevPack(w, False).data)
    # sets nobody asks for are dropped after two refreshes
    for _ in range(3):
        with T.fixed_weights(store):
            T.param_pack(w, False)
    assert len(store) == 1


def test_long_frame_axis_falls_back_to_torch_ops_under_the_hip_backend():
    """64 frames with 8 heads and a relative position bias: the backward attention core would need its [8][64][64]
    bias-gradient table plus four tiles in LDS (> 160 KiB), so the "hip" backend's dispatcher (autograd._temporal)
    takes the torch ops for this layer instead of failing in loss.backward() (reference layer: vdp.py:546-612, 636)."""
    from video_diffusion_amd import autograd as A, train_hip as T
    from video_diffusion_amd.modules import Attention, EinopsToAndFrom, PreNorm, Residual, RotaryEmbedding
    assert T.attention_core_fits(32, 8, True) and not T.attention_core_fits(64, 8, True)
    assert not T.attention_core_fits(300, 8, False)
    torch.manual_seed(3)
    c, f = 64, 64
    rot = RotaryEmbedding(32)
    res = Residual(PreNorm(c, EinopsToAndFrom('b c f h w', 'b (h w) f c', Attention(c, heads=8, dim_head=32, rotary_emb=rot)))).to(DEV)
    x = torch.randn(1, c, f, 4, 4, device=DEV)
    bias = (0.3 * torch.randn(8, f, f, device=DEV))

    def run(hip_flag):
        old = A._HIP_ATTN
        A._HIP_ATTN = hip_flag
        try:
            xr = x.clone().requires_grad_(True)
            br = bias.clone().requires_grad_(True)
            out = A._temporal(res, xr, br)
            out.square().mean().backward()
            return out.detach(), xr.grad.clone(), br.grad.clone()
        finally:
            A._HIP_ATTN = old
    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)          # the same torch ops ran both times


def _xattn_core_ref(q, kvrows, b, n, L, heads):
    """q [b*n][hid] already scaled by scale * log2 e, kvrows [b*L][2*hid]: softmax over the L keys per head, natural feature
    order; returns (o [b*n][hid], log2-domain lse [b][heads][n])."""
    hid = heads * 32
    qf = q.float().view(b, n, heads, 32).transpose(1, 2)
    kf = kvrows.float()[:, :hid].reshape(b, L, heads, 32).transpose(1, 2)
    vf = kvrows.float()[:, hid:].reshape(b, L, heads, 32).transpose(1, 2)
    s2 = qf @ kf.transpose(-1, -2)                                  # log2 domain
    lse2 = torch.logsumexp(s2 * 0.6931471805599453, dim=-1) / 0.6931471805599453
    p = torch.exp2(s2 - lse2[..., None])
    o = (p @ vf).transpose(1, 2).reshape(b * n, hid)
    return o, lse2


@pytest.mark.parametrize("b,n,L,heads", [(1, 64, 32, 2), (2, 100, 37, 4), (1, 1000, 300, 8), (2, 515, 1024, 8), (1, 4096, 64, 1),
                                         (1, 5, 1, 1), (3, 33, 33, 2), (1, 20000, 257, 4)])      # one key; ragged by one; many chunks
def test_cross_attention_core_forward_backward_vs_torch(b, n, L, heads):
    """vdmi_xattn_core_train_fwd / _bwd (CrossAttention.forward's softmax attention, vdp.py:511-521) against torch autograd
    over the same bf16 operands: output, log-sum-exp, and the gradients of q and of the projected key / value rows."""
    from video_diffusion_amd import _lib, train_hip as T
    lib = _lib.lib()
    hid = heads * 32
    g = torch.Generator().manual_seed(n + L)
    q = _bf(torch.randn(b * n, hid, generator=g) * 0.8).to(DEV)
    kv = _bf(torch.randn(b * L, 2 * hid, generator=g)).to(DEV)
    dout_nat = _bf(torch.randn(b * n, hid, generator=g)).to(DEV)
    perm = T._stored_feature_order(hid, DEV)
    s = torch.cuda.current_stream().cuda_stream
    pack = torch.empty(2 * lib.vdmi_xattn_kv_elems(1, b, L, heads), device=DEV, dtype=torch.bfloat16)
    assert lib.vdmi_xattn_pack_kv_train(kv.data_ptr(), 2 * hid, pack.data_ptr(), b, L, heads, s) == 0
    npad = (n + 31) // 32 * 32
    o = torch.empty(b * n, hid, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(b, heads, npad, device=DEV)
    assert lib.vdmi_xattn_core_train_fwd(q.data_ptr(), hid, o.data_ptr(), pack.data_ptr(), lse.data_ptr(), npad, b, n, L, heads, s) == 0
    # the plain core over the two-form pack gives the same output bits
    pack2 = torch.empty(lib.vdmi_xattn_kv_elems(1, b, L, heads), device=DEV, dtype=torch.bfloat16)
    o2 = torch.empty_like(o)
    assert lib.vdmi_xattn_pack_kv(kv.data_ptr(), 2 * hid, pack2.data_ptr(), 1, b, L, heads, s) == 0
    assert lib.vdmi_xattn_core(q.data_ptr(), hid, o2.data_ptr(), pack2.data_ptr(), b, n, L, heads, s) == 0
    assert torch.equal(o, o2)

    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    o_ref, lse_ref = _xattn_core_ref(qr, kvr, b, n, L, heads)
    o_nat = torch.empty_like(o)
    o_nat[:, perm] = o                                   # stored column c holds feature perm[c]
    tag = f"xattn_core.n{n}_L{L}_h{heads}"
    assert record(tag + ".o", rel_err(o_nat.float(), o_ref), 6e-3) < 6e-3
    assert record(tag + ".lse", rel_err(lse[:, :, :n], lse_ref), 1e-4) < 1e-4
    o_ref.backward(dout_nat.float())

    dout = dout_nat[:, perm].contiguous()                # what the out-projection's input gradient looks like: stored order
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    delta = torch.zeros(b, heads, npad, device=DEV)
    ws = torch.empty(lib.vdmi_xattn_bwd_workspace_bytes(b, n, L, heads), device=DEV, dtype=torch.uint8)
    st = lib.vdmi_xattn_core_train_bwd(q.data_ptr(), hid, o.data_ptr(), dout.data_ptr(), pack.data_ptr(), kv.data_ptr(), 2 * hid,
                                       lse.data_ptr(), delta.data_ptr(), npad, dq.data_ptr(), dkv.data_ptr(), 2 * hid, b, n, L,
                                       heads, ws.data_ptr(), s)
    assert st == 0
    torch.cuda.synchronize()
    assert record(tag + ".dq", rel_err(dq.float(), qr.grad), 2e-2) < 2e-2
    assert record(tag + ".dk", rel_err(dkv.float()[:, :hid], kvr.grad[:, :hid]), 2e-2) < 2e-2
    assert record(tag + ".dv", rel_err(dkv.float()[:, hid:], kvr.grad[:, hid:]), 2e-2) < 2e-2
    dkv2 = torch.empty_like(kv)
    assert lib.vdmi_xattn_core_train_bwd(q.data_ptr(), hid, o.data_ptr(), dout.data_ptr(), pack.data_ptr(), kv.data_ptr(), 2 * hid,
                                         lse.data_ptr(), delta.data_ptr(), npad, dq.data_ptr(), dkv2.data_ptr(), 2 * hid, b, n, L,
                                         heads, ws.data_ptr(), s) == 0
    assert torch.equal(dkv, dkv2)                        # partial sums are added in a fixed order


@pytest.mark.parametrize("c,f,h,w,L,cd", [(64, 3, 8, 8, 48, 512), (128, 2, 6, 5, 100, 64)])
def test_cross_attention_layer_forward_backward_vs_torch_autograd(c, f, h, w, L, cd):
    """Residual(PreNorm(CrossAttention)) (vdp.py:480-528, 803-805): output and every gradient, including the context's,
    against the torch ops of autograd.py."""
    from video_diffusion_amd import autograd as A, train_hip as T
    from video_diffusion_amd.modules import CrossAttention, PreNorm, Residual
    torch.manual_seed(3)
    layer = Residual(PreNorm(c, CrossAttention(c, context_dim=cd, heads=4, dim_head=32))).to(DEV)
    with torch.no_grad():
        layer.fn.norm.gamma.copy_(1 + 0.2 * torch.randn_like(layer.fn.norm.gamma))
    g = torch.Generator().manual_seed(8)
    b = 2
    x = torch.randn(b, c, f, h, w, generator=g).to(DEV)
    ctx = torch.randn(b, L, cd, generator=g).to(DEV)
    dout = torch.randn(b, c, f, h, w, generator=g).to(DEV)

    def run(fn):
        xr, cr = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
        for p in layer.parameters():
            p.grad = None
        out = fn(layer, xr, cr)
        out.backward(dout)
        grads = {n: p.grad.clone() for n, p in layer.named_parameters()}
        grads["x"], grads["context"] = xr.grad.clone(), cr.grad.clone()
        return out.detach().float(), grads

    out_ref, g_ref = run(A._cross_attention_ops)
    assert T.cross_attention_fits(layer.fn.fn, ctx)
    out_hip, g_hip = run(T.cross_attention)
    tag = f"xattn_layer.c{c}"
    assert record(tag + ".out", rel_err(out_hip, out_ref), 1e-2) < 1e-2
    assert set(g_hip) == set(g_ref)
    for name in sorted(g_ref):
        assert g_hip[name].shape == g_ref[name].shape, name
        assert record(f"{tag}.grad.{name}", rel_err(g_hip[name], g_ref[name]), 3e-2) < 3e-2, name


def test_unet_backward_with_frame_conditioning_on_the_hip_kernels():
    """A training step's loss.backward() with previous-frame conditioning (vdp.py:1146-1161 with frame_cond_features, the
    CrossAttention layers of vdp.py:803-805, 824-826): loss, the gradient of the conditioning features (what trains the
    FrameConditioningEncoder) and all parameter gradients, "hip" backend against the all-torch backend; the
    cross-attention layers run on the library (none falls back)."""
    from tests.helpers import inp, make_unet
    from video_diffusion_amd import train_hip as T
    m = make_unet(salt=2, dim=64, dim_mults=(1, 2), channels=3).to(DEV)
    assert m.has_frame_cross_attn
    x = inp("bwdfc.x", (2, 3, 4, 16, 16)).to(DEV)
    noise = inp("bwdfc.noise", (2, 3, 4, 16, 16)).to(DEV)
    feats = inp("bwdfc.feat", (2, 40, 512)).to(DEV)
    t = torch.tensor([17, 203], device=DEV)
    calls = []
    orig = T.CrossAttnFn.apply

    def run(backend):
        m.train_backend = backend
        for p in m.parameters():
            p.grad = None
        fr = feats.clone().requires_grad_(True)
        loss = F.mse_loss(m(x, t, frame_cond_features=fr), noise)
        loss.backward()
        m.train_backend = None
        return loss.detach(), fr.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    l_ref, gf_ref, g_ref = run("torch-autograd")
    T.cross_attention_calls = 0
    orig_ca = T.cross_attention

    def spy(*a, **k):
        calls.append(1)
        return orig_ca(*a, **k)
    T.cross_attention = spy
    try:
        l_hip, gf_hip, g_hip = run("hip")
    finally:
        T.cross_attention = orig_ca
    assert len(calls) == len(m.frame_cross_attns_down) + len(m.frame_cross_attns_up)
    assert set(g_ref) == set(g_hip)
    assert any("frame_cross_attns" in n for n in g_hip)
    cat = lambda d: torch.cat([d[n].flatten() for n in sorted(d)])
    xa = lambda d: torch.cat([d[n].flatten() for n in sorted(d) if "frame_cross_attns" in n])
    assert record("unet_bwd_framecond.loss", abs(float(l_hip - l_ref)) / float(l_ref), 2e-2) < 2e-2
    assert record("unet_bwd_framecond.features_grad", rel_err(gf_hip, gf_ref), 6e-2) < 6e-2
    assert record("unet_bwd_framecond.cross_attn_param_grads", rel_err(xa(g_hip), xa(g_ref)), 6e-2) < 6e-2
    assert record("unet_bwd_framecond.all_param_grads", rel_err(cat(g_hip), cat(g_ref)), 6e-2) < 6e-2


@pytest.mark.parametrize("c,f,h,w,heads", [(64, 3, 8, 8, 4), (128, 5, 10, 6, 8)])
def test_spatio_temporal_attention_layer_forward_backward_vs_torch_autograd(c, f, h, w, heads):
    """Residual(PreNorm(SpatioTemporalAttention)) of the version-1 variant (version1diffusion.py:325-345: every token
    attends to all f*h*w tokens of its sample) on the cross-attention kernels, against the explicit softmax of autograd.py."""
    from video_diffusion_amd import autograd as A, train_hip as T
    from video_diffusion_amd.modules import PreNorm, Residual
    from video_diffusion_amd.version1 import SpatioTemporalAttention
    torch.manual_seed(4)
    layer = Residual(PreNorm(c, SpatioTemporalAttention(c, heads=heads, dim_head=32))).to(DEV)
    with torch.no_grad():
        layer.fn.norm.gamma.copy_(1 + 0.2 * torch.randn_like(layer.fn.norm.gamma))
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, c, f, h, w, generator=g).to(DEV)
    dout = torch.randn(2, c, f, h, w, generator=g).to(DEV)

    def run(fn):
        xr = x.clone().requires_grad_(True)
        for p in layer.parameters():
            p.grad = None
        out = fn(layer, xr)
        out.backward(dout)
        grads = {n: p.grad.clone() for n, p in layer.named_parameters()}
        grads["x"] = xr.grad.clone()
        return out.detach().float(), grads

    out_ref, g_ref = run(A._linear_attention_ops)
    out_hip, g_hip = run(T.spatio_temporal_attention)
    tag = f"st_attn_layer.c{c}"
    assert record(tag + ".out", rel_err(out_hip, out_ref), 1e-2) < 1e-2
    assert set(g_hip) == set(g_ref)
    for name in sorted(g_ref):
        assert g_hip[name].shape == g_ref[name].shape, name
        assert record(f"{tag}.grad.{name}", rel_err(g_hip[name], g_ref[name]), 3e-2) < 3e-2, name


@pytest.mark.parametrize("cin,cout,temb", [(64, 128, 32), (64, 64, None)])
def test_version1_resnet_block_with_temporal_conv_vs_torch_autograd(cin, cout, temb):
    """The version-1 ResnetBlock (version1diffusion.py:301-323: the packaged block, then a (3,1,1) convolution over frames,
    then the residual branch) entirely on the library: output and every gradient against the torch ops of autograd.py."""
    from video_diffusion_amd import autograd as A, train_hip as T
    rb = _rand_block(cin, cout, temb, 13)
    rb.temporal_conv = torch.nn.Conv3d(cout, cout, (3, 1, 1), padding=(1, 0, 0)).to(DEV)
    g = torch.Generator().manual_seed(2)
    b, f, h, w = 2, 4, 8, 8
    x = torch.randn(b, cin, f, h, w, generator=g).to(DEV)
    te = torch.randn(b, temb, generator=g).to(DEV) if temb else None
    dout = torch.randn(b, cout, f, h, w, generator=g).to(DEV)
    seen = []
    orig = T.TemporalConvResFn.apply

    def run(fn):
        xr = x.clone().requires_grad_(True)
        ter = te.clone().requires_grad_(True) if te is not None else None
        for p in rb.parameters():
            p.grad = None
        out = fn(rb, xr, ter)
        out.backward(dout)
        grads = {n: p.grad.clone() for n, p in rb.named_parameters()}
        grads["x"] = xr.grad.clone()
        return out.detach().float(), grads

    out_ref, g_ref = run(A._resnet)
    out_hip, g_hip = run(T.resnet_block)
    tag = f"v1_resnet_bwd.{cin}to{cout}"
    assert record(tag + ".out", rel_err(out_hip, out_ref), 2e-2) < 2e-2
    assert set(g_hip) == set(g_ref) and "temporal_conv.weight" in g_hip
    for name in sorted(g_ref):
        assert g_hip[name].shape == g_ref[name].shape, name
        assert record(f"{tag}.grad.{name}", rel_err(g_hip[name], g_ref[name]), 3e-2) < 3e-2, name


def test_hip_backend_under_autocast_with_focus_flags_and_frame_conditioning():
    """Trainer(amp=True) wraps the loss in torch.autocast: the HIP layers ignore it (their Functions run outside autocast
    on bf16 rows), the few torch layers left (embeddings) follow it.  A step with focus-present flags (vdp.py:590-600) and
    previous-frame features (vdp.py:803-805) under bf16 autocast: loss and gradients against the all-torch backend under
    the same autocast, which normalises in float."""
    from tests.helpers import inp, make_unet
    m = make_unet(salt=5, dim=64, dim_mults=(1, 2), channels=3).to(DEV)
    x = inp("amp.x", (2, 3, 4, 16, 16)).to(DEV)
    noise = inp("amp.noise", (2, 3, 4, 16, 16)).to(DEV)
    feats = inp("amp.feat", (2, 24, 512)).to(DEV)
    t = torch.tensor([400, 30], device=DEV)
    focus = torch.tensor([False, True], device=DEV)

    def run(backend):
        m.train_backend = backend
        for p in m.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(x, t, frame_cond_features=feats, focus_present_mask=focus)
            loss = F.mse_loss(out.float(), noise)
        loss.backward()
        m.train_backend = None
        return loss.detach(), {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}

    l_ref, g_ref = run("torch-autograd")
    l_hip, g_hip = run("hip")
    assert set(g_ref) == set(g_hip)
    cat = lambda d: torch.cat([d[n].flatten() for n in sorted(d)])
    assert record("amp_focus_framecond.loss", abs(float(l_hip - l_ref)) / float(l_ref), 2e-2) < 2e-2
    assert record("amp_focus_framecond.all_param_grads", rel_err(cat(g_hip), cat(g_ref)), 4e-2) < 4e-2
