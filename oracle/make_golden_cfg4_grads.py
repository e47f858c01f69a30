"""configs[4] gradient fixture from the REAL reference (build container only; TEST INFRASTRUCTURE).

    python -m oracle.make_golden_cfg4_grads            # from the repo root; ~25 GB of host memory, a few minutes

One loss forward + backward of the reference Unet3D(dim=128, dim_mults=(1,2,4,8), channels=3) on one
(1, 3, 16, 128, 128) clip with an L1 loss against a noise target - the arithmetic of p_losses' denoiser call and
loss (video_diffusion_pytorch/video_diffusion_pytorch.py:1146-1161, backward at :1451) in pixel space, fp32.
Weights, input, timestep and target come from oracle.weightgen closed forms (nothing but outputs is stored):

    x = synthetic_input("cfg4.x"), t = 321, target = synthetic_input("cfg4.noise")

The fixture holds the loss, the L2 norm of every parameter's gradient (in state_dict order of the parameters that
require grad) and, for a few layers of every kind, the first 256 entries of the flattened gradient.  Parameters the
step does not reach (the cross-attention layers: no frame conditioning is passed) have norm -1.
"""
from __future__ import annotations

import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, weightgen  # noqa: E402
from oracle.make_golden import build_unet, save  # noqa: E402

CFG4 = dict(dim=128, dim_mults=(1, 2, 4, 8), channels=3)
SLICES = ("init_conv.weight", "init_temporal_attn.fn.fn.fn.to_qkv.weight", "downs.0.0.block1.proj.weight",
          "downs.0.0.mlp.1.weight", "downs.0.2.fn.fn.to_qkv.weight", "downs.1.0.res_conv.weight", "downs.1.4.weight",
          "downs.3.1.block2.proj.weight", "mid_spatial_attn.fn.fn.fn.to_out.weight", "mid_block2.block1.norm.gamma",
          "ups.0.0.block1.proj.weight", "ups.2.4.weight", "ups.3.3.fn.fn.fn.to_qkv.weight",
          "time_rel_pos_bias.relative_attention_bias.weight", "time_mlp.1.weight", "final_conv.0.block2.proj.weight",
          "final_conv.1.weight", "final_conv.1.bias")


def main():
    torch.set_num_threads(8)
    vdp = ref_import.load_reference()
    m = build_unet(vdp, **CFG4)
    m.train()
    x = weightgen.synthetic_input("cfg4.x", (1, 3, 16, 128, 128))
    target = weightgen.synthetic_input("cfg4.noise", (1, 3, 16, 128, 128))
    t = torch.tensor([321])
    t0 = time.time()
    loss = F.l1_loss(m(x, t), target)
    loss.backward()
    print(f"cfg4 forward + backward: {time.time() - t0:.1f} s, loss {float(loss):.6f}")
    names, norms = [], []
    for n, p in m.named_parameters():
        if not p.requires_grad:
            continue
        names.append(n)
        norms.append(-1.0 if p.grad is None else float(p.grad.double().norm()))
    sl = {}
    for n in SLICES:
        g = dict(m.named_parameters())[n].grad
        sl["slice." + n] = g.reshape(-1)[:256].clone()
    save("cfg4_grads", t=t, loss=loss.detach().reshape(1), grad_norms=torch.tensor(norms, dtype=torch.float64), **sl)
    with open(os.path.join(ROOT, "tests", "golden", "cfg4_grads_names.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    print("parameters:", len(names), "without gradient:", sum(1 for v in norms if v < 0))


if __name__ == "__main__":
    main()
