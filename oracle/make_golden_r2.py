"""Full-loop and full-size fixtures from the REAL reference (build container only; TEST INFRASTRUCTURE).

    python -m oracle.make_golden_r2            # from the repo root

* cfg0_full_T250 : config[0] (Unet3D dim 64, 5 x 32 x 32 RGB, batch 1), the WHOLE free-running
  p_sample_loop of T = 250 steps (video_diffusion_pytorch.py:1004-1014 of the reference) with x_T and every
  per-step noise tensor injected from oracle.weightgen (closed forms: not stored); holds the final sample, the state
  after 1, 5, 25, 50, 100, 150, 200 steps, and the same from a second run of the reference in which each epsilon is
  rounded to bf16 (the reference's own sensitivity to the smallest bf16 perturbation).
* cfg3_branches  : config[3] model: the conditional and the null-conditioned epsilon separately
  (the two evaluations inside forward_with_cond_scale, video_diffusion_pytorch.py:731-748).
* cfg4_eps       : the training-config shape, Unet3D(dim 128, channels 3) on (1, 3, 16, 128, 128), one epsilon.
* p_sample_step_dyn_T250 : one p_sample step with use_dynamic_thres=True (known epsilon, injected noise) and the
  per-sample thresholds.
* cfg1_embed     : the B = 2 inputs / outputs of cfg1_eps need no new data; nothing is written for it.
Fixtures hold inputs' keys and the reference's outputs only; weights come from oracle.weightgen.
"""
from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, weightgen  # noqa: E402
from oracle.make_golden import CFG0, CFG3, build_unet, save  # noqa: E402

CFG4 = dict(dim=128, dim_mults=(1, 2, 4, 8), channels=3)


@torch.no_grad()
def main(which):
    torch.set_num_threads(8)
    vdp = ref_import.load_reference()
    inp = weightgen.synthetic_input

    if "cfg0" in which:
        m0 = build_unet(vdp, **CFG0)
        gd0 = vdp.GaussianDiffusion(m0, torch.nn.Identity(), image_size=32, num_frames=5, channels=3, timesteps=250)
        shape = (1, 3, 5, 32, 32)
        real_randn, real_randn_like = torch.randn, torch.randn_like
        marks = (1, 5, 25, 50, 100, 150, 200)
        step, states = [0], {}

        def fake_randn(*a, **k):                 # the loop's x_T draw
            return inp("cfg0.full.xT", shape)

        def fake_randn_like(x, **k):             # one draw per step, in execution order; x is the state after step[0] steps
            if step[0] in marks:
                states[step[0]] = x.clone()
            z = inp(f"cfg0.full.z{step[0]}", shape)
            step[0] += 1
            return z

        def run():
            step[0] = 0
            states.clear()
            torch.randn, torch.randn_like = fake_randn, fake_randn_like
            try:
                out = gd0.p_sample_loop(shape)
            finally:
                torch.randn, torch.randn_like = real_randn, real_randn_like
            assert step[0] == 250
            return out, dict(states)
        t0 = time.time()
        final, st = run()
        print(f"cfg0 full loop: {time.time() - t0:.1f} s")
        # the same loop with nothing changed but each epsilon rounded to bf16 on its way out of the network: how far
        # the smallest perturbation any bf16 implementation makes moves the reference's own trajectory
        plain_forward = m0.forward
        m0.forward = lambda *a, **k: plain_forward(*a, **k).to(torch.bfloat16).float()
        final_r, st_r = run()
        m0.forward = plain_forward
        rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        print("reference, eps rounded to bf16, vs reference:",
              {k: round(rel(st_r[k], st[k]), 5) for k in marks}, "final", round(rel(final_r, final), 5))
        # and the reference itself computing in bf16 the way a PyTorch user would (autocast: matmuls and convolutions in
        # bf16 with fp32 accumulation, norms and softmax in fp32): the yardstick for a bf16 implementation's whole loop
        t0 = time.time()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            final_a, st_a = run()
        print(f"cfg0 full loop under bf16 autocast: {time.time() - t0:.1f} s")
        print("reference under bf16 autocast vs reference:",
              {k: round(rel(st_a[k], st[k]), 5) for k in marks}, "final", round(rel(final_a.float(), final), 5))
        save("cfg0_full_T250_autocast", final=final_a.float(),
             states=torch.stack([st_a[k].float() for k in marks]).to(torch.float16))
        save("cfg0_full_T250", final=final, marks=torch.tensor(marks),
             states=torch.stack([st[k] for k in marks]).to(torch.float16),
             states_bf16eps=torch.stack([st_r[k] for k in marks]).to(torch.float16), final_bf16eps=final_r)
        del m0, gd0

    if "cfg3" in which:
        m3 = build_unet(vdp, **CFG3)
        x3 = inp("cfg3.x", (1, 3, 16, 64, 64))
        c3 = inp("cfg3.cond", (1, 768))
        t3 = torch.tensor([700])
        save("cfg3_branches", t=t3, eps_cond=m3(x3, t3, cond=c3, null_cond_prob=0.),
             eps_null=m3(x3, t3, cond=c3, null_cond_prob=1.))
        del m3

    if "cfg4" in which:
        m4 = build_unet(vdp, **CFG4)
        x4 = inp("cfg4.x", (1, 3, 16, 128, 128))
        t4 = torch.tensor([321])
        t0 = time.time()
        eps = m4(x4, t4)
        print(f"cfg4 forward: {time.time() - t0:.1f} s")
        save("cfg4_eps", t=t4, eps=eps.to(torch.float16))       # half precision halves the fixture; the bar is 2e-2
        del m4

    if "dyn" in which:
        # one p_sample step of the reference with dynamic thresholding (video_diffusion_pytorch.py:973-1002 with
        # use_dynamic_thres=True), known epsilon and injected noise; the timesteps make the 0.9-quantile of |x0| fall
        # below 1 (clamped to 1) in some samples and far above it in others
        Fr, H = 4, 12
        shape = (4, 3, Fr, H, H)
        gd = vdp.GaussianDiffusion(torch.nn.Identity(), torch.nn.Identity(), image_size=H, num_frames=Fr, channels=3,
                                   timesteps=250, use_dynamic_thres=True, dynamic_thres_percentile=0.9)
        xs, es, zs = inp("dyn.x", shape, scale=1.2), inp("dyn.eps", shape), inp("dyn.z", shape)
        xs[0] *= 0.25                                        # sample 0: quantile below 1, the threshold clamps to 1
        ts = torch.tensor([0, 3, 60, 249])

        class _Fixed(torch.nn.Module):
            def forward(self, x, t, **k):
                return es

        gd.denoise_fn = _Fixed()
        real_randn_like = torch.randn_like
        torch.randn_like = lambda x, **k: zs.clone()
        try:
            out = gd.p_sample(xs, ts)
        finally:
            torch.randn_like = real_randn_like
        x0 = gd.predict_start_from_noise(xs, t=ts, noise=es)
        s = torch.quantile(x0.reshape(4, -1).abs(), 0.9, dim=-1).clamp(min=1.)
        print("dyn thresholds", s.tolist())
        save("p_sample_step_dyn_T250", t=ts, out=out, s=s)


if __name__ == "__main__":
    main(sys.argv[1:] or ["cfg0", "cfg3", "cfg4", "dyn"])
