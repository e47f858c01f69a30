#!/usr/bin/env python3
"""Exercise the multi-GPU code path over RCCL with a world of ONE rank (all a one-GPU box offers): process-group
set-up on the device, barrier, all-reduce and the all-gather of parallel.gather_videos on device tensors."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from video_diffusion_amd.parallel import gather_videos, sample_sharded

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.randn(5, 1, 10, 64, 64, device=dev)
dist.barrier()
t = torch.tensor([3.5], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
g = gather_videos(x, 5)
s = sample_sharded(lambda a, b: x[a:b] * 2, 5)
torch.cuda.synchronize()
assert float(t) == 3.5 and torch.equal(g, x) and torch.equal(s, x * 2)
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank path ok")
