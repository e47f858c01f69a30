"""The IEEE-half precision build (VDMI_PRECISION=f16, libvdmi_f16.so) on the GPU.

The precision is chosen once per process (it selects the shared library and the storage dtype), so the checks run in
a child process: the model-level parity tests of the other GPU test files under VDMI_PRECISION=f16.  There the bars
are the stated tolerance itself: 2e-2 on the final sample of the free-running T = 250 loop of configs[0] (reference
loop: video_diffusion_pytorch/video_diffusion_pytorch.py:1004-1014) and on the guided epsilon of configs[3] at
cond_scale 2 (video_diffusion_pytorch.py:731-748); the bf16 build reaches 1.3e-1 / 2.2e-2 on these two.
The child's measured errors go to gpurun_out/parity_errors_f16.json.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_half_precision_build_meets_the_stated_tolerances():
    out = os.path.join(ROOT, "gpurun_out", "parity_errors_f16.json")
    env = dict(os.environ, VDMI_PRECISION="f16", VDMI_PARITY_OUT=out)
    sel = ("cfg0_full_loop or cfg0_eps or cfg1_eps_vs or cfg3 or cfg4_eps or tiny_eps or streaming_conv or "
           "temporal_attention or spatial_linear_attention or resnet_split_path or split_plan_matches")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_configs_gpu.py"),
                        os.path.join(ROOT, "tests", "test_hip_parity.py"), "-q", "-x", "-m", "gpu", "-k", sel,
                        "--timeout", "300", "--timeout-method", "thread", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    errs = json.load(open(out))["errors"]
    assert errs["cfg0_full_loop_T250_final_sample"]["rel_err"] < 2e-2
    assert errs["cfg3_eps_guided_scale2"]["rel_err"] < 2e-2
    assert errs["cfg1_eps"]["rel_err"] < 5e-3          # per-step epsilon: 1.8e-3 measured (bf16 build: 1.4e-2)
    from tests.helpers import record
    for k in ("cfg0_full_loop_T250_final_sample", "cfg3_eps_guided_scale2", "cfg1_eps", "cfg0_eps_t0", "cfg4_eps"):
        if k in errs:
            record("f16_build_" + k, errs[k]["rel_err"], errs[k]["bar"])
