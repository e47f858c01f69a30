"""Trainer surface of the reference (vdp.py:1187-1527): constructor arguments, EMA, checkpoint
format (`model-{milestone}.pt` with keys step/model/ema/scaler/opt) and GIF I/O.  The optimisation
step itself needs the denoiser's backward pass, which the HIP path does not provide yet; `train()`
says so instead of silently running something else.
"""
import copy
from pathlib import Path

import numpy as np
import torch
from PIL import Image
from torch.optim import Adam

CHANNELS_TO_MODE = {1: 'L', 3: 'RGB', 4: 'RGBA'}


def video_tensor_to_gif(tensor, path, duration=120, loop=0, optimize=True):
    """tensor: (C, F, H, W) float in [0, 1]."""
    frames = []
    for fr in tensor.detach().cpu().clamp(0, 1).unbind(dim=1):
        arr = (fr * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
        frames.append(Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr))
    first, *rest = frames
    first.save(path, save_all=True, append_images=rest, duration=duration, loop=loop, optimize=optimize)
    return frames


def gif_to_tensor(path, channels=3):
    img = Image.open(path)
    mode = CHANNELS_TO_MODE[channels]
    out, i = [], 0
    while True:
        try:
            img.seek(i)
        except EOFError:
            break
        arr = np.asarray(img.convert(mode), dtype=np.float32) / 255.0
        out.append(torch.from_numpy(arr.reshape(arr.shape[0], arr.shape[1], -1)).permute(2, 0, 1))
        i += 1
    return torch.stack(out, dim=1)


class EMA:
    def __init__(self, beta):
        self.beta = beta

    def update_model_average(self, ma_model, current_model):
        for cur, ma in zip(current_model.parameters(), ma_model.parameters()):
            ma.data = self.update_average(ma.data, cur.data)

    def update_average(self, old, new):
        if old is None:
            return new
        return old * self.beta + (1 - self.beta) * new


class Trainer(object):
    def __init__(self, diffusion_model, folder, *, ema_decay=0.995, num_frames=16, num_prev_frames=4,
                 train_batch_size=32, train_lr=1e-4, vae_lr=1e-4, train_num_steps=100000,
                 gradient_accumulate_every=2, amp=False, step_start_ema=2000, update_ema_every=10,
                 save_and_sample_every=1000, results_folder='./results', num_sample_rows=2, max_grad_norm=None,
                 null_frame_cond_prob=0.1, frame_cond_scale=1.5):
        super().__init__()
        self.model = diffusion_model
        self.ema = EMA(ema_decay)
        self.ema_model = copy.deepcopy(self.model)
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
        params = list(diffusion_model.denoise_fn.parameters()) + list(diffusion_model.vae.parameters())
        if diffusion_model.frame_cond_encoder is not None:
            params += list(diffusion_model.frame_cond_encoder.parameters())
        self.opt = Adam(params, lr=train_lr)
        self.step = 0
        self.amp = amp
        self.scaler = torch.amp.GradScaler("cuda", enabled=amp)
        self.max_grad_norm = max_grad_norm
        self.num_sample_rows = num_sample_rows
        self.results_folder = Path(results_folder)
        self.results_folder.mkdir(exist_ok=True, parents=True)
        self.reset_parameters()

    def reset_parameters(self):
        self.ema_model.load_state_dict(self.model.state_dict())

    def step_ema(self):
        if self.step < self.step_start_ema:
            self.reset_parameters()
            return
        self.ema.update_model_average(self.ema_model, self.model)

    def save(self, milestone):
        data = {'step': self.step, 'model': self.model.state_dict(), 'ema': self.ema_model.state_dict(),
                'scaler': self.scaler.state_dict(), 'opt': self.opt.state_dict()}
        torch.save(data, str(self.results_folder / f'model-{milestone}.pt'))

    def load(self, milestone, **kwargs):
        if milestone == -1:
            all_milestones = [int(p.stem.split('-')[-1]) for p in Path(self.results_folder).glob('*.pt')]
            assert len(all_milestones) > 0, 'need to have at least one milestone to load'
            milestone = max(all_milestones)
        data = torch.load(str(self.results_folder / f'model-{milestone}.pt'), weights_only=False)
        self.step = data['step']
        self.model.load_state_dict(data['model'], **kwargs)
        self.ema_model.load_state_dict(data['ema'], **kwargs)
        self.scaler.load_state_dict(data['scaler'])
        if 'opt' in data:
            self.opt.load_state_dict(data['opt'])

    def train(self, prob_focus_present=0., focus_present_mask=None, log_fn=None):
        raise NotImplementedError(
            "Trainer.train needs the Unet3D backward pass; the MI355X path is forward-only so far "
            "(sampling: GaussianDiffusion.p_sample_loop / sample)")
