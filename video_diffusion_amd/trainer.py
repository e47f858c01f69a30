"""Trainer of the reference (vdp.py:1187-1527 = video_diffusion_pytorch/video_diffusion_pytorch.py): constructor
arguments, GIF dataset, EMA, checkpoint format (`model-{milestone}.pt` with keys step/model/ema/scaler/opt),
the training loop and the periodic sample + save.

Gradients come from PyTorch autograd: by default over torch ops (`Unet3D.train_backend = "torch-autograd"`,
autograd.py), with `trainer.train_backend = "hip"` / "auto" (or VDMI_TRAIN_BACKEND) through autograd Functions whose
forward and backward are the HIP kernels (train_hip.py); the sampling the loop does at every milestone always runs on
the HIP plan.  What is new against the reference: data-parallel training when launched with one process per GPU (WORLD_SIZE > 1): the loss module
is wrapped in DistributedDataParallel (bucketed gradient all-reduce over RCCL, overlapped with backward; unused
parameters such as the frame cross-attention layers of an unconditioned batch are detected), each rank reads its
own shard of the dataset, rank 0 samples and saves; and the EMA update is one fused multi-tensor lerp instead
of a Python loop over parameters (vdp.py:316-329).
"""
import copy
import os
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch import nn
from torch.optim import Adam
from torch.utils import data

CHANNELS_TO_MODE = {1: 'L', 3: 'RGB', 4: 'RGBA'}


def exists(x):
    return x is not None


def noop(*args, **kwargs):
    pass


def cycle(dl):
    while True:
        for d in dl:
            yield d


def num_to_groups(num, divisor):
    groups, remainder = divmod(num, divisor)
    arr = [divisor] * groups
    if remainder > 0:
        arr.append(remainder)
    return arr


def video_tensor_to_gif(tensor, path, duration=120, loop=0, optimize=True):
    """tensor: (C, F, H, W) float in [0, 1]."""
    frames = []
    for fr in tensor.detach().cpu().clamp(0, 1).unbind(dim=1):
        arr = (fr * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
        frames.append(Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr))
    first, *rest = frames
    first.save(path, save_all=True, append_images=rest, duration=duration, loop=loop, optimize=optimize)
    return frames


def _to_tensor(img):
    arr = np.asarray(img, dtype=np.float32) / 255.0
    return torch.from_numpy(arr.reshape(arr.shape[0], arr.shape[1], -1)).permute(2, 0, 1)


def gif_to_tensor(path, channels=3, transform=_to_tensor):
    img = Image.open(path)
    mode = CHANNELS_TO_MODE[channels]
    out, i = [], 0
    while True:
        try:
            img.seek(i)
        except EOFError:
            break
        out.append(transform(img.convert(mode)))
        i += 1
    return torch.stack(out, dim=1)


def cast_num_frames(t, *, frames):
    f = t.shape[1]
    if f == frames:
        return t
    if f > frames:
        return t[:, :frames]
    return F.pad(t, (0, 0, 0, 0, 0, frames - f))


class Dataset(data.Dataset):
    """GIF folder -> {'current': (C, num_frames, S, S), 'prev': (C, num_prev_frames, S, S)} in [0, 1]
    (vdp.py:1237-1293).  Resize (shorter side, bilinear), optional horizontal flip, center crop."""

    def __init__(self, folder, image_size, channels=3, num_frames=16, num_prev_frames=4, horizontal_flip=False,
                 force_num_frames=True, exts=('gif',)):
        super().__init__()
        self.folder, self.image_size, self.channels = folder, image_size, channels
        self.num_frames, self.num_prev_frames = num_frames, num_prev_frames
        self.horizontal_flip, self.force_num_frames = horizontal_flip, force_num_frames
        self.paths = sorted(p for ext in exts for p in Path(f'{folder}').glob(f'**/*.{ext}'))

    def _transform(self, flip):
        s = self.image_size

        def fn(img):
            w, h = img.size
            if w <= h:
                nw, nh = s, max(s, int(s * h / w))
            else:
                nw, nh = max(s, int(s * w / h)), s
            img = img.resize((nw, nh), Image.BILINEAR)
            if flip:
                img = img.transpose(Image.FLIP_LEFT_RIGHT)
            left, top = int(round((nw - s) / 2.0)), int(round((nh - s) / 2.0))
            return _to_tensor(img.crop((left, top, left + s, top + s)))
        return fn

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        flip = self.horizontal_flip and bool(torch.rand(1).item() < 0.5)
        tensor = gif_to_tensor(self.paths[index], self.channels, transform=self._transform(flip))
        total = self.num_prev_frames + self.num_frames
        seq_len = tensor.shape[1]
        if seq_len >= total:
            start = torch.randint(0, seq_len - total + 1, (1,)).item()
            tensor = tensor[:, start:start + total]
        else:
            tensor = F.pad(tensor, (0, 0, 0, 0, 0, total - seq_len))
        if self.force_num_frames:
            tensor = cast_num_frames(tensor, frames=total)
        if tensor.shape[1] >= total:
            return {'current': tensor[:, self.num_prev_frames:total], 'prev': tensor[:, :self.num_prev_frames]}
        return {'current': tensor[:, :self.num_frames],
                'prev': torch.zeros(self.channels, self.num_prev_frames, self.image_size, self.image_size)}


class EMA:
    def __init__(self, beta):
        self.beta = beta

    def update_model_average(self, ma_model, current_model):
        # old * beta + (1 - beta) * new for every parameter, as one multi-tensor lerp per dtype/device group
        # The in-place update runs on the parameters themselves (under no_grad), so their version counters advance
        # and the HIP engine of the averaged model repacks its weights before the next sample().
        ma = list(ma_model.parameters())
        with torch.no_grad():
            cur = [p.detach().to(m.dtype) for p, m in zip(current_model.parameters(), ma)]
            if ma:
                torch._foreach_lerp_(ma, cur, 1.0 - self.beta)
        for mod in ma_model.modules():
            if hasattr(mod, "weights_changed"):
                mod.weights_changed()

    def update_average(self, old, new):
        if old is None:
            return new
        return old * self.beta + (1 - self.beta) * new


class _LossModule(nn.Module):
    """One micro-batch of the reference's loop body (vdp.py:1429-1449) as a module, so that
    DistributedDataParallel sees every parameter that receives a gradient in ONE forward."""

    def __init__(self, diffusion_model, null_frame_cond_prob):
        super().__init__()
        self.diffusion = diffusion_model
        self.null_frame_cond_prob = null_frame_cond_prob

    def forward(self, current_frames, prev_frames, prob_focus_present, focus_present_mask):
        kw = {}
        if self.diffusion.frame_cond_encoder is not None:
            kw["prev_frames"] = prev_frames
        diffusion_loss = self.diffusion(current_frames, prob_focus_present=prob_focus_present,
                                        focus_present_mask=focus_present_mask,
                                        null_frame_cond_prob=self.null_frame_cond_prob, **kw)
        current_normalized = current_frames * 2 - 1
        recon, mean, logvar = self.diffusion.vae(current_normalized)
        recon_loss = F.mse_loss(recon, current_normalized)
        kl_loss = -0.5 * torch.sum(1 + logvar - mean.pow(2) - logvar.exp()) / current_frames.numel()
        vae_loss = recon_loss + 0.0001 * kl_loss
        return diffusion_loss + 0.1 * vae_loss, diffusion_loss.detach(), vae_loss.detach()


class Trainer(object):
    def __init__(self, diffusion_model, folder, *, ema_decay=0.995, num_frames=16, num_prev_frames=4,
                 train_batch_size=32, train_lr=1e-4, vae_lr=1e-4, train_num_steps=100000,
                 gradient_accumulate_every=2, amp=False, step_start_ema=2000, update_ema_every=10,
                 save_and_sample_every=1000, results_folder='./results', num_sample_rows=2, max_grad_norm=None,
                 null_frame_cond_prob=0.1, frame_cond_scale=1.5):
        super().__init__()
        self.model = diffusion_model
        self.ema = EMA(ema_decay)
        graphs, diffusion_model._graphs = getattr(diffusion_model, "_graphs", {}), {}     # captured graphs are not copyable
        self.ema_model = copy.deepcopy(self.model)
        diffusion_model._graphs = graphs
        self.update_ema_every = update_ema_every
        self.step_start_ema = step_start_ema
        self.save_and_sample_every = save_and_sample_every
        self.batch_size = train_batch_size
        self.image_size = diffusion_model.image_size
        self.gradient_accumulate_every = gradient_accumulate_every
        self.train_num_steps = train_num_steps
        self.null_frame_cond_prob = null_frame_cond_prob
        self.frame_cond_scale = frame_cond_scale
        self.folder = folder

        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # VDMI_DDP_SINGLE_RANK=1: wrap in DistributedDataParallel even with one rank (a one-GPU box can then exercise the
        # bucketed RCCL all-reduce path)
        self.distributed = ((self.world > 1 or os.environ.get("VDMI_DDP_SINGLE_RANK") == "1")
                            and torch.distributed.is_available() and torch.distributed.is_initialized())

        self.ds = Dataset(folder, diffusion_model.image_size, channels=diffusion_model.channels,
                          num_frames=diffusion_model.num_frames, num_prev_frames=num_prev_frames)
        print(f'found {len(self.ds)} videos as gif files at {folder}')
        assert len(self.ds) > 0, 'need to have at least 1 video to start training'
        sampler = data.distributed.DistributedSampler(self.ds, num_replicas=self.world, rank=self.rank, shuffle=True) \
            if self.distributed else None
        self.dl = cycle(data.DataLoader(self.ds, batch_size=train_batch_size, shuffle=sampler is None, sampler=sampler,
                                        pin_memory=torch.cuda.is_available()))

        params = list(diffusion_model.denoise_fn.parameters()) + list(diffusion_model.vae.parameters())
        if diffusion_model.frame_cond_encoder is not None:
            params += list(diffusion_model.frame_cond_encoder.parameters())
        # the single-kernel form where the parameters live on the GPU (the same update; 1.2 against 2.8 ms per step for the
        # 147 M parameters of Unet3D(dim=128))
        self.opt = Adam(params, lr=train_lr, fused=all(p.is_cuda for p in params))
        self.step = 0
        self.amp = amp
        self.scaler = torch.amp.GradScaler("cuda", enabled=amp)
        self.max_grad_norm = max_grad_norm
        self.num_sample_rows = num_sample_rows
        self.results_folder = Path(results_folder)
        self.results_folder.mkdir(exist_ok=True, parents=True)
        self.reset_parameters()
        # not a constructor argument (the reference's Trainer signature stays as it is): set the attribute, or the
        # environment variable VDMI_TRAIN_BACKEND, before train()
        self.train_backend = os.environ.get("VDMI_TRAIN_BACKEND", "torch-autograd")

    @property
    def device(self):
        return next(self.model.denoise_fn.parameters()).device

    def reset_parameters(self):
        self.ema_model.load_state_dict(self.model.state_dict())

    def step_ema(self):
        if self.step < self.step_start_ema:
            self.reset_parameters()
            return
        self.ema.update_model_average(self.ema_model, self.model)

    def save(self, milestone):
        data_ = {'step': self.step, 'model': self.model.state_dict(), 'ema': self.ema_model.state_dict(),
                 'scaler': self.scaler.state_dict(), 'opt': self.opt.state_dict()}
        torch.save(data_, str(self.results_folder / f'model-{milestone}.pt'))

    def load(self, milestone, **kwargs):
        if milestone == -1:
            all_milestones = [int(p.stem.split('-')[-1]) for p in Path(self.results_folder).glob('*.pt')]
            assert len(all_milestones) > 0, 'need to have at least one milestone to load'
            milestone = max(all_milestones)
        data_ = torch.load(str(self.results_folder / f'model-{milestone}.pt'), weights_only=True)
        self.step = data_['step']
        self.model.load_state_dict(data_['model'], **kwargs)
        self.ema_model.load_state_dict(data_['ema'], **kwargs)
        self.scaler.load_state_dict(data_['scaler'])
        if 'opt' in data_:
            self.opt.load_state_dict(data_['opt'])

    def get_n_conditioning_clips(self, n: int):
        clips, remaining = [], n
        while remaining > 0:
            prev = next(self.dl)['prev'].to(self.device)
            take = min(remaining, prev.shape[0])
            clips.append(prev[:take])
            remaining -= take
        return torch.cat(clips, dim=0)

    def _sample_and_save(self, milestone):
        rows = self.num_sample_rows

        def grid(t):
            t = F.pad(t, (2, 2, 2, 2))
            n, c, f, h, w = t.shape
            return t.reshape(rows, n // rows, c, f, h, w).permute(2, 3, 0, 4, 1, 5).reshape(c, f, rows * h, (n // rows) * w)

        if self.model.frame_cond_encoder is not None:
            sample_prev = self.get_n_conditioning_clips(rows ** 2)
            cond = self.ema_model.sample(batch_size=rows ** 2, prev_frames=sample_prev, frame_cond_scale=self.frame_cond_scale)
            video_tensor_to_gif(grid(torch.cat([sample_prev, cond], dim=2)).clamp(0, 1),
                                str(self.results_folder / f'{milestone}_cond_with_context.gif'))
        uncond = torch.cat([self.ema_model.sample(batch_size=n) for n in num_to_groups(rows ** 2, self.batch_size)], dim=0)
        video_tensor_to_gif(grid(uncond), str(self.results_folder / f'{milestone}_unconditioned.gif'))
        self.save(milestone)

    def train(self, prob_focus_present=0., focus_present_mask=None, log_fn=noop):
        assert callable(log_fn)
        dev = self.device
        if hasattr(self.model.denoise_fn, "train_backend"):
            # "torch-autograd": every layer differentiates through torch ops; "hip-resnet": the ResnetBlocks (and the weight
            # gradients of init_conv / final_conv) run forward and backward on the HIP kernels (train_hip.py); "hip": the
            # temporal, linear and spatial attention layers and the Downsample / Upsample convolutions too
            backend = self.train_backend
            if backend == "auto":      # the HIP layers when the model's shapes allow them, else torch ops throughout
                from . import train_hip
                own = getattr(self.model.denoise_fn, "hip_training_supported", None)   # the guidance variant's own layers
                fits = own() if own is not None else train_hip.supports(self.model.denoise_fn)
                backend = "hip" if (dev.type == "cuda" and fits) else "torch-autograd"
            self.model.denoise_fn.train_backend = backend
        loss_core = _LossModule(self.model, self.null_frame_cond_prob)
        loss_mod = loss_core
        ddp_kw = dict(device_ids=[dev.index] if dev.type == "cuda" else None)
        unused_names = None             # distributed: parameters no step reaches, found in the first step
        if self.distributed:
            # Which parameters a step reaches is fixed by the Trainer's configuration (the cross-attention layers of a model
            # trained without previous frames, the null condition embedding of a model without a condition): the masks
            # drawn per step multiply tensors, they never remove a layer from the graph.  So the first step runs with
            # DDP's search for unused parameters; what it leaves without a gradient is then excluded from the reducer and
            # every later step runs without the walk over the autograd graph (checked per step: an excluded parameter
            # that does receive a gradient raises).
            loss_mod = nn.parallel.DistributedDataParallel(loss_core, find_unused_parameters=True, **ddp_kw)

        # packed weight sets of the HIP layers: kept over the micro-steps of an accumulation, repacked (one launch) after
        # the optimizer step has moved the parameters' version counters
        packs, store = _nullcontext, None
        if getattr(self.model.denoise_fn, "train_backend", None) in ("hip", "hip-resnet"):
            from . import train_hip
            store = train_hip.PackStore()
            packs = lambda: train_hip.fixed_weights(store)

        while self.step < self.train_num_steps:
            for i in range(self.gradient_accumulate_every):
                batch = next(self.dl)
                current_frames, prev_frames = batch['current'].to(dev), batch['prev'].to(dev)
                last = i == self.gradient_accumulate_every - 1
                sync = loss_mod.no_sync if (self.distributed and not last) else _nullcontext
                with sync(), packs():
                    with torch.autocast(dev.type, enabled=self.amp):
                        total_loss, diffusion_loss, vae_loss = loss_mod(current_frames, prev_frames, prob_focus_present,
                                                                        focus_present_mask)
                    self.scaler.scale(total_loss / self.gradient_accumulate_every).backward()
            log = {'loss': diffusion_loss.item(), 'vae_loss': vae_loss.item()}

            if self.distributed:
                named = [(n, p) for n, p in loss_core.named_parameters() if p.requires_grad]
                if unused_names is None:
                    unused_names = [n for n, p in named if p.grad is None]
                    nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(loss_core, unused_names)
                    loss_mod = nn.parallel.DistributedDataParallel(loss_core, find_unused_parameters=False, **ddp_kw)
                else:
                    late = [n for n, p in named if n in unused_set and p.grad is not None]
                    if late:
                        raise RuntimeError(f"parameters excluded from gradient averaging received gradients: {late[:4]}")
                unused_set = set(unused_names)

            if exists(self.max_grad_norm):
                self.scaler.unscale_(self.opt)
                nn.utils.clip_grad_norm_(self.model.denoise_fn.parameters(), self.max_grad_norm)
                nn.utils.clip_grad_norm_(self.model.vae.parameters(), self.max_grad_norm)
                if self.model.frame_cond_encoder is not None:
                    nn.utils.clip_grad_norm_(self.model.frame_cond_encoder.parameters(), self.max_grad_norm)
            self.scaler.step(self.opt)
            if store is not None:
                store.invalidate()      # a fused optimizer step leaves the parameters' version counters where they were
            if self.step % 20 == 0 and self.rank == 0:
                print(f'{self.step}: diff_loss={log["loss"]:.4f}, vae_loss={log["vae_loss"]:.4f}')
            self.scaler.update()
            self.opt.zero_grad()

            if self.step % self.update_ema_every == 0:
                self.step_ema()
            if self.step != 0 and (self.step + 1) % self.save_and_sample_every == 0 and self.rank == 0:
                self._sample_and_save(self.step // self.save_and_sample_every)
            log_fn(log)
            self.step += 1
        if self.rank == 0:
            print('training completed')


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
