"""Generate tests/golden/*.npz from the REAL reference (build container only).

    python -m oracle.make_golden            # from the repo root

Every fixture holds inputs and the reference's outputs only; weights are not
stored — they are re-created from ``oracle.weightgen`` (same key -> same
values), which this script also loads into the reference modules.
TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, weightgen  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def build_unet(vdp, salt=0, **kw):
    m = vdp.Unet3D(**kw)
    sd = weightgen.fill_state_dict(m.state_dict(), salt)
    m.load_state_dict(sd)
    return m.eval()


TINY = dict(dim=32, dim_mults=(1, 2), channels=3)
TINY_COND = dict(dim=32, dim_mults=(1, 2), channels=3, cond_dim=16)
CFG0 = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=3)
CFG1 = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=1)
CFG3 = dict(dim=128, dim_mults=(1, 2, 4, 8), channels=3, cond_dim=768)


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    vdp = ref_import.load_reference()
    inp = weightgen.synthetic_input

    # ---- A11: schedule buffers --------------------------------------------------
    for T in (8, 250, 1000):
        gd = vdp.GaussianDiffusion(torch.nn.Identity(), torch.nn.Identity(), image_size=32, num_frames=5,
                                   channels=3, timesteps=T)
        save(f"schedule_T{T}", **{k: v for k, v in gd.state_dict().items()})

    # ---- A5: relative position buckets ---------------------------------------------
    arrs = {}
    for md in (32, 128):
        for n in (1, 5, 10, 16, 40, 64, 200):
            q = torch.arange(n)
            rel = q[None, :] - q[:, None]
            arrs[f"bucket_md{md}_n{n}"] = vdp.RelativePositionBias._relative_position_bucket(
                rel, num_buckets=32, max_distance=md)
    save("relpos_buckets", **arrs)

    # ---- tiny model: per-submodule I/O ------------------------------------------------
    m = build_unet(vdp, **TINY)
    B, Fr, H = 2, 4, 12
    t = torch.tensor([3, 977])
    x_in = inp("tiny.x", (B, 3, Fr, H, H))
    bias = m.time_rel_pos_bias(Fr, device=x_in.device)
    temb = m.time_mlp(t)
    a32 = inp("tiny.a32", (B, 32, Fr, H, H))
    a32h = inp("tiny.a32h", (B, 32, Fr, H // 2, H // 2))
    a64h = inp("tiny.a64h", (B, 64, Fr, H // 2, H // 2), scale=1.5)
    a128h = inp("tiny.a128h", (B, 128, Fr, H // 2, H // 2))
    a64 = inp("tiny.a64", (B, 64, Fr, H, H))
    # inputs are re-created in the tests from the same synthetic_input keys (not stored)
    save("tiny_ops",
         t=t, shape=np.array([B, 3, Fr, H, H]),
         relpos_bias=bias, time_mlp=temb,
         init_conv=m.init_conv(x_in),
         init_temporal_attn=m.init_temporal_attn(a32, pos_bias=bias),
         downs_0_0=m.downs[0][0](a32, temb),
         downs_1_0=m.downs[1][0](a32h, temb),
         downs_1_1=m.downs[1][1](a64h, temb),
         downs_0_2=m.downs[0][2](a32),
         downs_1_2=m.downs[1][2](a64h),
         downs_0_3=m.downs[0][3](a32, pos_bias=bias, focus_present_mask=torch.zeros(B, dtype=torch.bool)),
         downs_0_4=m.downs[0][4](a32),
         ups_0_4=m.ups[0][4](a32h),
         ups_0_0=m.ups[0][0](a128h, temb),
         mid_block1=m.mid_block1(a64h, temb),
         mid_spatial_attn=m.mid_spatial_attn(a64h),
         mid_temporal_attn=m.mid_temporal_attn(a64h, pos_bias=bias),
         final_conv_0=m.final_conv[0](a64),
         final_conv_1=m.final_conv[1](a32),
         block1_noss=m.downs[0][0].block1(a32),
         eps=m(x_in, t))

    mc = build_unet(vdp, salt=1, **TINY_COND)
    cond = inp("tiny.cond", (B, 16))
    save("tiny_cond",
         t=t, cond=cond,
         eps_cond=mc(x_in, t, cond=cond, null_cond_prob=0.),
         eps_null=mc(x_in, t, cond=cond, null_cond_prob=1.),
         eps_cfg2=mc.forward_with_cond_scale(x_in, t, cond=cond, cond_scale=2.0))

    # ---- tiny diffusion: full T=8 loop with injected randomness ---------------------------
    gd = vdp.GaussianDiffusion(m, torch.nn.Identity(), image_size=H, num_frames=Fr, channels=3, timesteps=8)
    shape = (B, 3, Fr, H, H)
    x_T = inp("tiny.xT", shape)
    noises = [inp(f"tiny.z{k}", shape) for k in range(8)]
    queue = [x_T] + list(noises)
    real_randn, real_randn_like = torch.randn, torch.randn_like
    traj = []

    def fake_randn(*a, **k):
        return queue.pop(0).clone()

    def fake_randn_like(x, **k):
        return queue.pop(0).clone()

    real_p_sample = gd.p_sample

    def rec_p_sample(*a, **k):
        out = real_p_sample(*a, **k)
        traj.append(out.clone())
        return out

    torch.randn, torch.randn_like = fake_randn, fake_randn_like
    gd.p_sample = rec_p_sample
    try:
        final = gd.p_sample_loop(shape)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    assert not queue
    save("tiny_loop_T8", traj=torch.stack(traj), final=final)

    # one p_sample step with known eps (separates the posterior arithmetic from the UNet)
    gd250 = vdp.GaussianDiffusion(torch.nn.Identity(), torch.nn.Identity(), image_size=H, num_frames=Fr,
                                  channels=3, timesteps=250)
    xs = inp("step.x", (4, 3, Fr, H, H), scale=1.2)
    es = inp("step.eps", (4, 3, Fr, H, H))
    zs = inp("step.z", (4, 3, Fr, H, H))
    ts = torch.tensor([0, 1, 124, 249])

    class _Fixed(torch.nn.Module):
        def forward(self, x, t, **k):
            return es

    gd250.denoise_fn = _Fixed()
    torch.randn_like = lambda x, **k: zs.clone()
    try:
        step_out = gd250.p_sample(xs, ts)
    finally:
        torch.randn_like = real_randn_like
    save("p_sample_step_T250", t=ts, out=step_out)

    # ---- full-size eps ---------------------------------------------------------------
    m0 = build_unet(vdp, **CFG0)
    x0 = inp("cfg0.x", (1, 3, 5, 32, 32))
    arrs = dict()
    for tt in (0, 125, 249):
        arrs[f"eps_t{tt}"] = m0(x0, torch.tensor([tt]))
    save("cfg0_eps", **arrs)

    # config[0] trajectory: first 6 steps of T=250 with injected noise
    gd0 = vdp.GaussianDiffusion(m0, torch.nn.Identity(), image_size=32, num_frames=5, channels=3, timesteps=250)
    shape0 = (1, 3, 5, 32, 32)
    xT0 = inp("cfg0.xT", shape0)
    nz0 = [inp(f"cfg0.z{k}", shape0) for k in range(6)]
    img = xT0
    torch.randn_like = lambda x, **k: q0.pop(0).clone()
    q0 = list(nz0)
    try:
        for k, i in enumerate(reversed(range(250 - 6, 250))):
            img = gd0.p_sample(img, torch.full((1,), i, dtype=torch.long))
    finally:
        torch.randn_like = real_randn_like
    save("cfg0_traj6", out=img)
    del m0, gd0

    m1 = build_unet(vdp, **CFG1)
    x1 = inp("cfg1.x", (2, 1, 10, 64, 64))
    t1 = torch.tensor([500, 17])
    save("cfg1_eps", t=t1, eps=m1(x1, t1))
    del m1

    m3 = build_unet(vdp, **CFG3)
    x3 = inp("cfg3.x", (1, 3, 16, 64, 64))
    c3 = inp("cfg3.cond", (1, 768))
    t3 = torch.tensor([700])
    save("cfg3_eps", t=t3,
         eps_cfg2=m3.forward_with_cond_scale(x3, t3, cond=c3, cond_scale=2.0))


if __name__ == "__main__":
    main()
