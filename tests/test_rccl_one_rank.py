"""The multi-GPU path over RCCL on the one GPU a test box has: a world of one rank (process group on the device,
barrier, all-reduce, the all-gather of parallel.gather_videos).  Runs in its own process so the test session keeps
no process group."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_world_of_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0 and "rccl one-rank path ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_ddp_training_over_rccl_world_of_one_rank():
    """Trainer + DistributedDataParallel on the "nccl" backend (bucketed gradient all-reduce over RCCL), ResnetBlocks on the
    HIP backward kernels; one rank, in its own process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank_train.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "rccl one-rank ddp training ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
