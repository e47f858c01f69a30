#!/usr/bin/env python3
"""Data-parallel training over RCCL with a world of ONE rank (all a one-GPU box offers): the Trainer wraps its loss
module in DistributedDataParallel on the "nccl" backend, so the bucketed gradient all-reduce runs over RCCL on device
tensors - with the ResnetBlocks, temporal attention layers and resampling convolutions forward and backward on the HIP
kernels (train_backend "hip")."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from PIL import Image

from video_diffusion_amd import GaussianDiffusion, Trainer, Unet3D, VAE

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29633")
os.environ.update(RANK="0", WORLD_SIZE="1", VDMI_DDP_SINGLE_RANK="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
with tempfile.TemporaryDirectory() as tmp:
    folder = os.path.join(tmp, "gifs")
    os.makedirs(folder)
    rng = np.random.RandomState(0)
    for i in range(4):
        imgs = [Image.fromarray(rng.randint(0, 255, (36, 40, 3), dtype=np.uint8)) for _ in range(6)]
        imgs[0].save(os.path.join(folder, f"v{i}.gif"), save_all=True, append_images=imgs[1:], duration=50, loop=0)
    torch.manual_seed(0)
    gd = GaussianDiffusion(Unet3D(dim=64, dim_mults=(1, 2), channels=4), VAE(channels=3, latent_dim=4, base_dim=64),
                           image_size=32, num_frames=4, channels=3, timesteps=50).to(dev)
    tr = Trainer(gd, folder, num_prev_frames=2, train_batch_size=2, train_num_steps=2, gradient_accumulate_every=2,
                 step_start_ema=0, update_ema_every=1, save_and_sample_every=10 ** 6, results_folder=os.path.join(tmp, "r"))
    assert tr.distributed
    tr.train_backend = "hip"
    before = gd.denoise_fn.downs[0][0].block1.proj.weight.detach().clone()
    logs = []
    tr.train(log_fn=logs.append)
    torch.cuda.synchronize()
    assert len(logs) == 2 and all(np.isfinite(l["loss"]) for l in logs), logs
    assert not torch.equal(before, gd.denoise_fn.downs[0][0].block1.proj.weight)
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank ddp training ok", [round(l["loss"], 4) for l in logs])
