"""CPU tests of host-side logic that does not need the GPU: checkpoint conventions, EMA bookkeeping, the ISA gate."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rotary_convention_follows_the_checkpoint_key():
    """A checkpoint written with the PyPI rotary_embedding_torch carries `...rotary_emb.freqs`; one written with the
    reference tree's own class (video_diffusion_guidance.py:182-198) carries `...rotary_emb.inv_freq`.  Loading sets
    the pairing convention of every rotary module accordingly (reference vdp.py:19, 635-636)."""
    from video_diffusion_amd import Unet3D
    from video_diffusion_amd.modules import RotaryEmbedding
    m = Unet3D(dim=32, dim_mults=(1, 2), channels=3)
    rots = [x for x in m.modules() if isinstance(x, RotaryEmbedding)]
    assert rots and all(r.convention == "half" for r in rots)
    sd = m.state_dict()
    pypi = {}
    for k, v in sd.items():
        if k.endswith("rotary_emb.inv_freq"):
            pypi[k[:-len("inv_freq")] + "freqs"] = v.clone()
        else:
            pypi[k] = v
    m.load_state_dict(pypi, strict=False)
    assert all(r.convention == "interleaved" for r in rots)
    m.load_state_dict(sd)
    assert all(r.convention == "half" for r in rots)


def test_ema_update_advances_parameter_versions():
    """The packed-weight cache of the HIP engine is keyed on autograd version counters: the EMA update must
    advance them (an update through `.data` would not)."""
    from video_diffusion_amd.trainer import EMA
    a, b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    before = [p._version for p in a.parameters()]
    old = [p.detach().clone() for p in a.parameters()]
    EMA(0.9).update_model_average(a, b)
    assert all(p._version > v for p, v in zip(a.parameters(), before))
    for p, o, q in zip(a.parameters(), old, b.parameters()):
        assert torch.allclose(p, o * 0.9 + q.detach() * 0.1, atol=1e-6)


def test_engine_generation_and_fingerprint_notice_data_writes():
    """`.data` writes leave pointer and version unchanged; the explicit generation bump and the value fingerprint
    are what catch them (host-side bookkeeping only: nothing is packed here)."""
    from video_diffusion_amd import Unet3D
    from video_diffusion_amd.engine import UnetEngine
    m = Unet3D(dim=32, dim_mults=(1, 2), channels=3)
    eng = UnetEngine(m)
    k0, v0 = eng._param_key("cpu"), eng._values_key()
    p = next(m.parameters())
    p.data.mul_(1.5)
    assert eng._param_key("cpu") == k0            # the blind spot
    assert eng._values_key() != v0                # what the sampling loops check once per loop
    v1 = eng._values_key()
    p.data.neg_()                                 # norm-preserving rewrites: a sign flip, two rows swapped
    assert eng._values_key() != v1
    p.data.neg_()
    assert eng._values_key() == v1
    first, second = p.data[0].clone(), p.data[1].clone()
    p.data[0], p.data[1] = second, first
    assert eng._values_key() != v1
    m._engine = eng
    m.weights_changed()
    assert eng._param_key("cpu") != k0            # what EMA / callers trigger explicitly


def test_no_kernel_has_a_chain_of_adjacent_dependent_mfmas():
    """Build gate: three or more MFMAs issued back to back into the same accumulator once corrupted a different kernel
    running on another stream on MI355X (the cause is a hypothesis, the rule is cheap): no kernel may contain one."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_chain_scan.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "chains of three or more: 0" in r.stdout


def test_small_modules_on_their_own_match_the_oracle():
    """RelativePositionBias(n, device) and SinusoidalPosEmb(t) called as the reference's modules are (vdp.py:306-312,
    339-351): bit-equal to the oracle's restatement; sub-blocks that only run inside the fused pipeline still raise."""
    import pytest
    import torch
    from oracle import vdm_oracle as O
    from video_diffusion_amd.modules import Block, RelativePositionBias, SinusoidalPosEmb
    rpb = RelativePositionBias(heads=8, num_buckets=32, max_distance=32)
    with torch.no_grad():
        for n in (1, 5, 10, 16):
            assert torch.equal(rpb(n, "cpu"), O.relpos_bias(rpb.relative_attention_bias.weight, n, 32, 32))
    t = torch.tensor([0., 3., 999.])
    assert torch.equal(SinusoidalPosEmb(64)(t), O.sinusoidal_emb(t, 64))
    with pytest.raises(NotImplementedError):
        Block(64, 64)(torch.zeros(1, 64, 1, 8, 8))


def _bench_dry(*argv, env_extra=None):
    import json
    env = dict(os.environ, VDMI_BENCH_DRY_RUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=240)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_starts_its_own_ranks_when_called_plainly():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (torch.distributed.run on a free
    local port); the dry-run mode makes the ranks only rendezvous over gloo, so this needs no GPU.  Exactly one JSON
    line comes back (rank 0's), with the world size and the per-GPU batch of the weak-scaling default and of the
    strong-scaling `--global-batch` form."""
    r, out = _bench_dry("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1 and out[0]["n_gpus"] == 2 and out[0]["ranks_met"]
    assert out[0]["scaling"] == "weak" and out[0]["batch_per_gpu"] == 16 and out[0]["global_batch"] == 32
    r, out = _bench_dry("--gpus", "2", "--global-batch", "256")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1 and out[0]["scaling"] == "strong" and out[0]["batch_per_gpu"] == 128
    # a batch that does not divide over the ranks is refused by every rank, and the parent reports the failure
    r, out = _bench_dry("--gpus", "2", "--global-batch", "255")
    assert r.returncode != 0 and not out


def test_bench_under_a_launcher_checks_the_world_size():
    """Under torch.distributed.run the ranks find WORLD_SIZE in the environment and do not launch again; a
    `--gpus` that disagrees with it is an error, not a silent single-rank run."""
    r, out = _bench_dry("--gpus", "4", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    r, out = _bench_dry("--gpus", "1", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0 and out and out[0]["n_gpus"] == 1
