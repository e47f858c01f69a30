"""Import the real reference in the BUILD CONTAINER ONLY (TEST INFRASTRUCTURE).

``/root/reference`` does not exist on the GPU box; nothing that runs there may
import this module.  It is used by ``oracle/make_golden.py`` to produce the
fixtures under ``tests/golden`` and by CPU tests that are skipped when the
reference tree is absent.

Three of the reference's imports are not installed in the image and carry no
arithmetic of the hot path except the rotary embedding:

* ``torchvision`` (``vdp.py:11``) — only GIF/dataset plumbing; stubbed with no-ops.
* ``einops_exts`` (``vdp.py:17``) — ``rearrange_many`` is ``map(rearrange)``; restated.
* ``rotary_embedding_torch`` (``vdp.py:19``) — replaced by the reference tree's own
  ``RotaryEmbedding`` (``video_diffusion_guidance.py:182-198``).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "video_diffusion_pytorch"))


def load_reference():
    """Returns the module ``video_diffusion_pytorch.video_diffusion_pytorch`` of the reference."""
    import torch
    from einops import rearrange

    sys.dont_write_bytecode = True            # the reference tree is read-only
    if "video_diffusion_pytorch.video_diffusion_pytorch" in sys.modules and \
            getattr(sys.modules["video_diffusion_pytorch.video_diffusion_pytorch"], "_vdmi_ref", False):
        return sys.modules["video_diffusion_pytorch.video_diffusion_pytorch"]

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvu = types.ModuleType("torchvision.utils")
    for name in ("ToTensor", "ToPILImage", "Resize", "RandomHorizontalFlip", "CenterCrop", "Lambda", "Compose"):
        setattr(tvt, name, lambda *a, **k: (lambda x: x))
    tv.transforms, tv.utils = tvt, tvu
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    sys.modules.setdefault("torchvision.utils", tvu)

    ee = types.ModuleType("einops_exts")
    ee.rearrange_many = lambda ts, pattern, **kw: map(lambda t: rearrange(t, pattern, **kw), ts)
    ee.check_shape = lambda *a, **k: None
    sys.modules.setdefault("einops_exts", ee)

    rot = types.ModuleType("rotary_embedding_torch")

    class _Placeholder(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    rot.RotaryEmbedding = _Placeholder
    sys.modules.setdefault("rotary_embedding_torch", rot)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import video_diffusion_pytorch.video_diffusion_pytorch as vdp      # noqa: E402
    import video_diffusion_guidance as vdg                              # noqa: E402

    vdp.RotaryEmbedding = vdg.RotaryEmbedding
    vdp._vdmi_ref = True
    return vdp
