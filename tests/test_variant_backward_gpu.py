"""GPU tests of the guidance variant's training layers (train_hip_variant.py, the GroupNorm / depthwise-convolution
backward kernels of csrc/misc.hip).

The checker is PyTorch autograd in fp32 over the same bf16-rounded operands (kernel tests) and
autograd.guidance_variant_forward without the HIP backend, which tests/test_training.py pins to the reference's own
output and to the oracle's input gradient (layer and model tests).  Bars per assertion: sums of fp32 terms 1e-3; anything
that passes through bf16 activations or activation gradients 2e-2 to 6e-2 relative L2."""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import record, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("c,b,rps,act,post,ld", [(64, 2, 1000, 1, True, 64), (128, 3, 515, 1, False, 128), (512, 1, 2048, 0, False, 512),
                                                   (256, 2, 40000, 1, True, 320), (2048, 1, 70, 1, True, 2048),
                                                   (64, 4, 1, 1, True, 64), (1024, 2, 3, 0, False, 1024)])      # one to three rows per sample
def test_groupnorm_backward_vs_torch_autograd(c, b, rps, act, post, ld):
    """vdmi_groupnorm_bwd: dx and the per-(sample, channel) sums against autograd over F.group_norm -> SiLU -> affine."""
    from video_diffusion_amd import train_hip_variant as HV
    g = torch.Generator().manual_seed(c + rps)
    x = _bf(torch.randn(b * rps, c, generator=g) * 1.5 + 0.3).to(DEV)
    dfull = _bf(torch.randn(b * rps, ld, generator=g)).to(DEV)
    dout = dfull[:, :c]
    gamma = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV)
    beta = (0.2 * torch.randn(c, generator=g)).to(DEV)
    pg = (1 + 0.5 * torch.randn(b, c, generator=g)).to(DEV).contiguous() if post else None
    ps = (0.1 * torch.randn(b, c, generator=g)).to(DEV).contiguous() if post else None

    xf = x.float().view(b, rps, c).requires_grad_(True)
    gam, bet = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    pgr = pg.clone().requires_grad_(True) if post else None
    psr = ps.clone().requires_grad_(True) if post else None
    y = F.group_norm(xf.permute(0, 2, 1), 8, gam, bet, 1e-5)                # (b, c, rps)
    z = F.silu(y) if act else y
    if post:
        z = z * pgr[:, :, None] + psr[:, :, None]
    z.backward(dout.float().view(b, rps, c).permute(0, 2, 1))

    dx, chan = HV.groupnorm_bwd_rows(x, dout, 8, gamma, beta, 1e-5, b, act, pg)
    torch.cuda.synchronize()
    tag = f"gn_bwd.c{c}_r{rps}"
    assert record(tag + ".dx", rel_err(dx.float(), xf.grad.reshape(-1, c)), 1e-2) < 1e-2          # bf16 output rounding
    tot = chan.sum(0)
    assert record(tag + ".dbeta", rel_err(tot[:, 0], bet.grad), 2e-3) < 2e-3
    assert record(tag + ".dgamma", rel_err(tot[:, 1], gam.grad), 2e-3) < 2e-3
    if post:
        assert record(tag + ".dpost_gain", rel_err(chan[:, :, 2], pgr.grad), 2e-3) < 2e-3
        assert record(tag + ".dpost_shift", rel_err(chan[:, :, 3], psr.grad), 2e-3) < 2e-3
    # no atomics anywhere: a second call gives the same bits
    dx2, chan2 = HV.groupnorm_bwd_rows(x, dout, 8, gamma, beta, 1e-5, b, act, pg)
    assert torch.equal(dx, dx2) and torch.equal(chan, chan2)


@pytest.mark.parametrize("b,f,hw,c,k", [(2, 5, 64, 64, 3), (1, 10, 100, 128, 3), (2, 3, 16, 512, 5), (1, 1, 256, 64, 3), (3, 4, 9, 256, 7)])
def test_depthwise_temporal_conv_gradients_vs_torch(b, f, hw, c, k):
    """vdmi_temporal_dwconv_wgrad (taps and bias) and the input gradient (the forward kernel with reversed taps)."""
    from video_diffusion_amd import train_hip_variant as HV
    g = torch.Generator().manual_seed(b * 100 + c)
    x = _bf(torch.randn(b * f * hw, c, generator=g)).to(DEV)
    du = _bf(torch.randn(b * f * hw, c, generator=g)).to(DEV)
    w = torch.randn(c, k, generator=g).to(DEV)
    x5 = x.float().view(b, f, hw, c).permute(0, 3, 1, 2).unsqueeze(-1).requires_grad_(True)       # (b, c, f, hw, 1)
    wr = w.clone().view(c, 1, k, 1, 1).requires_grad_(True)
    bias = torch.zeros(c, device=DEV, requires_grad=True)
    y = F.conv3d(x5, wr, bias, padding=(k // 2, 0, 0), groups=c)
    y.backward(du.float().view(b, f, hw, c).permute(0, 3, 1, 2).unsqueeze(-1))
    got = HV.dwconv_wgrad_rows(x, du, b, f, hw, k)
    tag = f"dwconv_bwd.c{c}_k{k}"
    assert record(tag + ".dw", rel_err(got[:, :k], wr.grad.view(c, k)), 1e-3) < 1e-3
    assert record(tag + ".dbias", rel_err(got[:, k], bias.grad), 1e-3) < 1e-3
    dx = HV.dwconv_rows(du, w.flip(1).contiguous(), None, b, f, hw)
    ref = x5.grad.squeeze(-1).permute(0, 2, 3, 1).reshape(-1, c)
    assert record(tag + ".dx", rel_err(dx.float(), ref), 1e-2) < 1e-2


def _variant_block(cin, cout, temb, seed):
    from video_diffusion_amd import guidance_variant as GV
    torch.manual_seed(seed)
    blk = GV.ResnetBlock(cin, cout, time_emb_dim=temb)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.GroupNorm):
                m.weight.copy_(1 + 0.2 * torch.randn_like(m.weight))
                m.bias.copy_(0.1 * torch.randn_like(m.bias))
    return blk.to(DEV)


def _block_ops(blk, x, temb):
    # ResnetBlock.forward of video_diffusion_guidance.py:370-388 as torch ops (what autograd.guidance_variant_forward runs)
    def ts(tb, x):
        h = F.silu(tb.temporal_norm(tb.temporal_conv(x)))
        return F.silu(tb.spatial_norm(tb.spatial_conv(h)))
    res = blk.res_conv(x)
    h = ts(blk.ts_block1, x)
    if blk.mlp is not None and temb is not None:
        scale, shift = blk.mlp(temb)[:, :, None, None, None].chunk(2, dim=1)
        h = h * (scale + 1) + shift
    return ts(blk.ts_block2, h) + res


@pytest.mark.parametrize("cin,cout,temb,shape", [(64, 128, 32, (2, 3, 16, 16)), (64, 64, None, (1, 5, 8, 8)),
                                                 (256, 128, 48, (2, 4, 8, 12)), (128, 128, 16, (2, 1, 16, 16))])
def test_variant_resnet_block_forward_backward_vs_torch_autograd(cin, cout, temb, shape):
    """One ResnetBlock of the guidance variant (two TemporalSpatialBlocks, time scale / shift, the (3,1,1) + GroupNorm
    residual branch when the width changes): output and every gradient against torch autograd over the same module."""
    from video_diffusion_amd import train_hip_variant as HV
    blk = _variant_block(cin, cout, temb, 5)
    g = torch.Generator().manual_seed(9)
    b, f, h, w = shape
    x = torch.randn(b, cin, f, h, w, generator=g).to(DEV)
    te = torch.randn(b, temb, generator=g).to(DEV) if temb else None
    dout = torch.randn(b, cout, f, h, w, generator=g).to(DEV)

    def run(fn):
        xr = x.clone().requires_grad_(True)
        ter = te.clone().requires_grad_(True) if te is not None else None
        for p in blk.parameters():
            p.grad = None
        out = fn(blk, xr, ter)
        out.backward(dout)
        grads = {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None}
        grads["x"] = xr.grad.clone()
        if ter is not None:
            grads["temb"] = ter.grad.clone()
        return out.detach().float(), grads

    out_ref, g_ref = run(_block_ops)
    out_hip, g_hip = run(HV.resnet_block)
    tag = f"gv_block_bwd.{cin}to{cout}"
    assert record(tag + ".out", rel_err(out_hip, out_ref), 2e-2) < 2e-2
    assert set(g_hip) == set(g_ref)
    for name in sorted(g_ref):
        assert g_hip[name].shape == g_ref[name].shape, name
        if float(g_ref[name].norm()) < 1e-4 * float(g_ref["x"].norm()):
            continue        # conv biases straight in front of a GroupNorm: their gradient is rounding noise around zero
        assert record(f"{tag}.grad.{name}", rel_err(g_hip[name], g_ref[name]), 4e-2) < 4e-2, name


def _variant_model():
    from oracle import weightgen
    from video_diffusion_amd import guidance_variant as GV
    m = GV.Unet3D(dim=64, dim_mults=(1, 2), channels=4)
    m.load_state_dict(weightgen.fill_state_dict(m.state_dict(), 4))
    return m.to(DEV)


def test_variant_unet_backward_on_the_hip_layers_vs_torch_autograd():
    """The whole denoiser of video_diffusion_guidance.py (:596-635): loss, input gradient (what the guidance sampler asks
    for, :725-761) and all parameter gradients, HIP layers against the torch ops pinned to the reference."""
    from tests.helpers import inp
    m = _variant_model()
    assert m.hip_training_supported()
    x = inp("gvb.x", (2, 4, 4, 16, 16)).to(DEV)
    target = inp("gvb.n", (2, 4, 4, 16, 16)).to(DEV)
    t = torch.tensor([17, 203], device=DEV)

    def run(backend):
        m.train_backend = backend
        for p in m.parameters():
            p.grad = None
        xr = x.clone().requires_grad_(True)
        loss = F.mse_loss(m(xr, t), target)
        loss.backward()
        m.train_backend = None
        return loss.detach(), xr.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    l_ref, gx_ref, g_ref = run("torch-autograd")
    l_hip, gx_hip, g_hip = run("hip")
    assert set(g_ref) == set(g_hip)
    assert record("gv_unet_bwd.loss", abs(float(l_hip - l_ref)) / float(l_ref), 2e-2) < 2e-2
    cat = lambda d: torch.cat([d[n].flatten() for n in sorted(d)])
    assert record("gv_unet_bwd.input_grad", rel_err(gx_hip, gx_ref), 6e-2) < 6e-2
    assert record("gv_unet_bwd.all_param_grads", rel_err(cat(g_hip), cat(g_ref)), 6e-2) < 6e-2
    worst = max((rel_err(g_hip[n], g_ref[n]), n) for n in g_ref if g_ref[n].norm() > 1e-3 * cat(g_ref).norm())
    record("gv_unet_bwd.worst_param_grad", worst[0], None)
    # the input gradient alone (frozen parameters, as the sampler runs it): same numbers, no parameter gradient computed
    for p in m.parameters():
        p.requires_grad_(False)
    m.train_backend = "hip"
    xr = x.clone().requires_grad_(True)
    gx_only, = torch.autograd.grad(F.mse_loss(m(xr, t), target), xr)
    m.train_backend = None
    assert rel_err(gx_only, gx_hip) < 1e-6


def test_variant_guidance_function_runs_on_the_hip_layers():
    """temporal_consistency_guidance over the variant picks the HIP layers on the GPU (the parameters are frozen for the
    call: input gradient only) and agrees with the all-torch gradient."""
    from video_diffusion_amd import VAE, guidance_variant as GV, train_hip_variant as HV
    torch.manual_seed(6)
    unet = GV.Unet3D(dim=64, dim_mults=(1, 2), channels=4)
    vae = VAE(channels=3, latent_dim=4, base_dim=64)
    gd = GV.GaussianDiffusion(unet.eval(), vae.eval(), image_size=64, num_frames=3, channels=3, timesteps=250).to(DEV)
    z = torch.randn(2, 4, 3, 8, 8, device=DEV)
    t = torch.tensor([40, 200], device=DEV)
    prev = torch.rand(2, 3, 2, 64, 64, device=DEV)
    calls = []
    orig = HV.groupnorm_bwd_rows

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    HV.groupnorm_bwd_rows = spy
    try:
        g_hip = GV.temporal_consistency_guidance(gd, prev, 2, w_r=3.0)(z, t)
    finally:
        HV.groupnorm_bwd_rows = orig
    assert calls, "the guidance gradient did not run through the HIP layers"
    orig_sup = HV.supports
    HV.supports = lambda m: False
    try:
        g_ref = GV.temporal_consistency_guidance(gd, prev, 2, w_r=3.0)(z, t)
    finally:
        HV.supports = orig_sup
    assert gd.denoise_fn.train_backend is None
    assert bool(torch.isfinite(g_hip).all())
    assert record("gv_guidance.hip_vs_torch", rel_err(g_hip, g_ref), 8e-2) < 8e-2
