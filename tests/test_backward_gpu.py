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
        nfpad = (nf + 15) // 16 * 16
        ref = torch.zeros(c, hw, nfpad, dtype=torch.bfloat16, device=DEV)
        ref[:, :, :nf] = rows.view(nf, hw, c).permute(2, 1, 0)
        assert torch.equal(out.view(c, hw, nfpad), ref)


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


@pytest.mark.parametrize("c,batch,rpb,with_ss", [(64, 2, 1000, True), (256, 3, 77, True), (1024, 2, 130, False), (128, 1, 5000, True)])
def test_rmsnorm_silu_bwd_vs_torch(c, batch, rpb, with_ss):
    from video_diffusion_amd import train_hip as T
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
    tag = f"normbwd.c{c}"
    assert record(tag + ".dy", rel_err(dy.float().cpu(), yr.grad.cpu()), 1e-2) < 1e-2          # bf16 output
    assert record(tag + ".dgamma", rel_err(dgamma.cpu(), gr.grad.cpu()), 2e-3) < 2e-3
    assert record(tag + ".dbias", rel_err(dbias.cpu(), yr.grad.sum(0).cpu()), 2e-3) < 2e-3
    if with_ss:
        assert record(tag + ".dscale", rel_err(dscale.cpu(), sr.grad.cpu()), 2e-3) < 2e-3
        assert record(tag + ".dshift", rel_err(dshift.cpu(), hr.grad.cpu()), 2e-3) < 2e-3


def test_colsum():
    from video_diffusion_amd import train_hip as T
    g = torch.Generator().manual_seed(5)
    for rows, c in [(5000, 64), (333, 1024), (70, 8)]:
        x = _bf(torch.randn(rows, c, generator=g)).to(DEV)
        assert rel_err(T.colsum_rows(x).cpu(), x.double().sum(0).cpu()) < 1e-4


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
    # the loss is an L1: its gradient is a sign pattern, which flips wherever the bf16 forward moves a residual across
    # zero; the parameter gradients (sums over all positions) are the stable quantity
    cat = lambda d: torch.cat([d[n].flatten() for n in sorted(d)])
    assert record("unet_bwd.all_param_grads", rel_err(cat(g_hip), cat(g_ref)), 5e-2) < 5e-2
    assert record("unet_bwd.input_grad", rel_err(gx_hip, gx_ref), 1e-1) < 1e-1
    worst = max((rel_err(g_hip[n], g_ref[n]), n) for n in g_ref if g_ref[n].norm() > 1e-6 * cat(g_ref).norm())
    record("unet_bwd.worst_param_grad", worst[0], None)
