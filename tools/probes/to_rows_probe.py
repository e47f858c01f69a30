#!/usr/bin/env python3
"""Which to_rows() calls of one "hip" training step really copy (input not channels-last bf16), by caller and shape."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    from video_diffusion_amd import Unet3D, train_hip
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    unet = Unet3D(dim=128, dim_mults=(1, 2, 4, 8), channels=3).to(dev)
    unet.train_backend = "hip"
    x = torch.randn(1, 3, 16, 128, 128, device=dev)
    noise = torch.randn_like(x)
    t = torch.randint(0, 1000, (1,), device=dev)
    seen = collections.Counter()
    orig = train_hip.to_rows

    def probe(v):
        view = v.permute(0, 2, 3, 4, 1)
        copies = v.dtype != torch.bfloat16 or not view.is_contiguous()
        if copies:
            fr = [f for f in traceback.extract_stack()[:-1] if "video_diffusion_amd" in f.filename][-2:]
            seen[(tuple(v.shape), str(v.dtype), tuple(v.stride()), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(fr)))] += 1
        return orig(v)
    for mod in list(sys.modules.values()):
        if mod is not None and getattr(mod, "__name__", "").startswith("video_diffusion_amd") and getattr(mod, "to_rows", None) is orig:
            mod.to_rows = probe
    store = train_hip.PackStore()
    for it in range(2):
        seen.clear()
        for p in unet.parameters():
            p.grad = None
        store.invalidate()
        with train_hip.fixed_weights(store):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.l1_loss(unet(x, t).float(), noise)
            loss.backward()
    torch.cuda.synchronize()
    for (shape, dt, stride, where), n in sorted(seen.items(), key=lambda kv: -kv[1] * kv[0][0][1] * kv[0][0][3] * kv[0][0][4]):
        print(n, shape, dt, stride, where)


if __name__ == "__main__":
    main()
