#!/usr/bin/env python3
"""Which torch ops (by input shape and Python call site) make up the glue kernels of one "hip" training step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    from video_diffusion_amd import Unet3D, train_hip
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    unet = Unet3D(dim=128, dim_mults=(1, 2, 4, 8), channels=3).to(dev)
    unet.train_backend = "hip"
    x = torch.randn(1, 3, 16, 128, 128, device=dev)
    noise = torch.randn_like(x)
    t = torch.randint(0, 1000, (1,), device=dev)
    store = train_hip.PackStore()

    def step():
        for p in unet.parameters():
            p.grad = None
        store.invalidate()
        with train_hip.fixed_weights(store):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.l1_loss(unet(x, t).float(), noise)
            loss.backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=4) if e.key.startswith("aten::") and e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    for e in rows[:40]:
        stack = " <- ".join(s.split("/")[-1][:60] for s in e.stack[:3])
        print(f"{e.self_device_time_total / 1e3:7.3f} ms x{e.count:4d} {e.key:28s} {str(e.input_shapes)[:70]:70s} {stack}")


if __name__ == "__main__":
    main()
