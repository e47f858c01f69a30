#!/usr/bin/env python3
"""Cross-stream interference probe: one fused linear-attention call on stream A while stream B runs N copies of
each distinct launch of a second plan; the call's workspace (its partial sums and folded weights) is compared bit
for bit with a serial run.  Only two extra streams are made (with more streams than hardware queues HIP
serialises them and nothing overlaps).      python tools/concurrency_probe.py [N]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from video_diffusion_amd import Unet3D
from video_diffusion_amd.engine import Plan
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Unet3D(dim=64, dim_mults=(1, 2, 4, 8), channels=1).to(dev).eval()
eng = m.engine(); eng.ensure_packed(dev)
p0, p1 = Plan(eng, 8, 10, 64, 64, dev), Plan(eng, 8, 10, 64, 64, dev)
for p in (p0, p1):
    p.x_in.copy_(torch.randn_like(p.x_in)); p.t_in.fill_(500)
    for c in p.calls:      # no side stream: keep the number of streams within the hardware queues
        assert c.fn(torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
name = "ups.3.2.fused"
i0 = [i for i, c in enumerate(p0.calls) if c.name == name][0]
ws0 = [k for k in p0.keep if isinstance(k, torch.Tensor) and k.dtype == torch.uint8][4]
main = torch.cuda.current_stream()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
p0.calls[i0].fn(main.cuda_stream); torch.cuda.synchronize()
ref = ws0.clone()
seen = set()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for j, c in enumerate(p1.calls):
    kern = (c.args or {}).get("kernel", c.name.split(".")[-1])
    key = (kern, c.name.split(".")[0] in ("downs", "ups") and c.name.split(".")[1])
    if key in seen:
        continue
    seen.add(key)
    bad = 0
    for it in range(20):
        s0.wait_stream(main); s1.wait_stream(main)
        for k in range(reps):
            c.fn(s1.cuda_stream)
            if k == 0:
                p0.calls[i0].fn(s0.cuda_stream)
        main.wait_stream(s0); main.wait_stream(s1)
        torch.cuda.synchronize()
        if not torch.equal(ws0, ref):
            bad += 1
            p0.calls[i0].fn(main.cuda_stream); torch.cuda.synchronize()
    print(f"{c.name:32s} {kern:22s} x{reps} beside the SLA call: {bad}/20 differ", flush=True)
