"""Where does a reduced-precision denoiser's final-sample error come from?  (TEST INFRASTRUCTURE, CPU only.)

    python -m oracle.rounding_study [--out profiles/r3_rounding_study.json] [--threads 4] [--only bf16:both,...]

Runs the oracle's free-running T = 250 loop of BASELINE configs[0] (Unet3D dim 64, 5 x 32 x 32 RGB, batch 1; the
reference's loop is video_diffusion_pytorch/video_diffusion_pytorch.py:1004-1014) with the x_T and per-step noise
of the committed fixture tests/golden/cfg0_full_T250.npz, once per rounding emulation:

    {weights only, activations only, both} x {bf16, fp16}

"weights": every convolution / projection weight of the denoiser is rounded to the format (what a packed-weight
implementation does once).  "activations": every stored layer output and every matrix-product operand is rounded
(what bf16 / fp16 rows in memory and 16-bit matrix-core operands do); accumulation, normalisation statistics, softmax
and the sampler state stay fp32, as in the HIP path.  The first and the last convolution stay fp32 (they do in the
HIP path: split-bf16 init conv, fp32 final 1x1 conv).

Each run is compared with the fp32 reference's states after 1 .. 200 steps and its final sample (the fixture), as
relative L2 error.  Nothing here touches the product path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import vdm_oracle as O  # noqa: E402
from oracle import weightgen  # noqa: E402

CFG0 = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=3)
SHAPE = (1, 3, 5, 32, 32)
T = 250


class Emu:
    def __init__(self, dtype, weights: bool, acts: bool):
        self.dtype, self.weights, self.acts = dtype, weights, acts
        self.overflow = 0

    def _r(self, t):
        r = t.to(self.dtype).float()
        if self.dtype == torch.float16:
            self.overflow += int((~torch.isfinite(r)).sum())
        return r

    def w(self, t):
        return self._r(t) if self.weights else t

    def a(self, t):
        return self._r(t) if self.acts else t


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


@torch.no_grad()
def run(sd, cfg, bufs, x_T, noises, marks):
    x, states = x_T, {}
    for k, i in enumerate(reversed(range(T))):
        if k in marks:
            states[k] = x.clone()
        t = torch.full((1,), i, dtype=torch.long)
        eps = O.unet3d_forward(sd, cfg, x, t)
        x = O.p_sample_step(bufs, x, t, eps, noises[k])
    return x, states


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r3_rounding_study.json"))
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)

    from video_diffusion_amd import Unet3D
    sd = weightgen.fill_state_dict(Unet3D(**CFG0).state_dict())
    cfg = O.UnetCfg(**CFG0)
    bufs = O.diffusion_buffers(T)
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg0_full_T250.npz"))
    marks = [int(k) for k in g["marks"]]
    ref_states = {k: torch.from_numpy(g["states"][j]).float() for j, k in enumerate(marks)}
    ref_final = torch.from_numpy(g["final"]).float()
    x_T = weightgen.synthetic_input("cfg0.full.xT", SHAPE)
    noises = [weightgen.synthetic_input(f"cfg0.full.z{k}", SHAPE) for k in range(T)]

    variants = [("fp32", None, False, False)]
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        variants += [(f"{name}:weights", dt, True, False), (f"{name}:activations", dt, False, True),
                     (f"{name}:both", dt, True, True)]
    only = [v for v in args.only.split(",") if v]
    results = {}
    if os.path.exists(args.out):
        try:
            results = json.load(open(args.out)).get("runs", {})
        except Exception:
            results = {}
    for name, dt, w, a in variants:
        if only and name not in only:
            continue
        O.EMU = None if dt is None else Emu(dt, w, a)
        t0 = time.time()
        final, states = run(sd, cfg, bufs, x_T, noises, marks)
        rec = {"states": {str(k): rel(states[k], ref_states[k]) for k in marks}, "final": rel(final, ref_final),
               "seconds": round(time.time() - t0, 1)}
        if O.EMU is not None:
            rec["fp16_overflows"] = O.EMU.overflow
        O.EMU = None
        results[name] = rec
        print(name, json.dumps(rec), flush=True)
        json.dump({"note": "oracle T=250 loop of configs[0] under emulated rounding; relative L2 error against the fp32 "
                           "reference's states / final sample (tests/golden/cfg0_full_T250.npz); oracle/rounding_study.py",
                   "runs": results}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
