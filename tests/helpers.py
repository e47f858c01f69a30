"""Shared test helpers: golden fixtures, synthetic weights, oracle configuration."""
import os

import numpy as np
import torch

from oracle import weightgen
from oracle.vdm_oracle import UnetCfg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = dict(dim=32, dim_mults=(1, 2), channels=3)
TINY_COND = dict(dim=32, dim_mults=(1, 2), channels=3, cond_dim=16)
CFG0 = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=3)
CFG1 = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=1)
CFG3 = dict(dim=128, dim_mults=(1, 2, 4, 8), channels=3, cond_dim=768)


def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def make_unet(salt=0, **kw):
    """The package's Unet3D with closed-form weights (identical to what make_golden gave the reference)."""
    from video_diffusion_amd import Unet3D
    m = Unet3D(**kw)
    m.load_state_dict(weightgen.fill_state_dict(m.state_dict(), salt))
    return m.eval()


def oracle_cfg(**kw):
    return UnetCfg(dim=kw["dim"], dim_mults=tuple(kw.get("dim_mults", (1, 2, 4, 8))), channels=kw.get("channels", 4),
                   cond_dim=kw.get("cond_dim"))


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


inp = weightgen.synthetic_input


# measured parity errors of the run, written by tests/conftest.py at session end (profiles/parity_errors.json is a copy)
PARITY_LOG = {}


def record(name, err, bar=None):
    """Remember a measured error (and the bar it was held to) under `name`; returns err."""
    PARITY_LOG[name] = {"rel_err": float(err), "bar": bar}
    return err
