"""The allocation-free GroupNorm + SiLU entry point against the one that takes a workspace, through the C ABI.

Two calls in a row on one stream, the second with a smaller input: the static scratch array keeps the first call's
partial sums, and the second call must not be disturbed by them.  Same kernels, same order of additions: bit-exact.
Oversized inputs are refused with VDMI_E_SCRATCH.
"""
import pytest
import torch

from video_diffusion_amd import _lib

pytestmark = pytest.mark.gpu

SCRATCH = 1 << 20


def _run(fn_name, x, gamma, beta, nsamples, rows, c, groups, with_ws):
    lib = _lib.lib()
    out = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    if with_ws:
        ws = torch.empty(lib.vdmi_groupnorm_workspace_bytes(nsamples, rows, c), device=x.device, dtype=torch.uint8)
        st = lib.vdmi_groupnorm_silu(x.data_ptr(), out.data_ptr(), nsamples, rows, c, groups, gamma.data_ptr(),
                                     beta.data_ptr(), 1e-5, ws.data_ptr(), s)
    else:
        st = lib.vdmi_groupnorm_silu_noalloc(x.data_ptr(), out.data_ptr(), nsamples, rows, c, groups, gamma.data_ptr(),
                                             beta.data_ptr(), 1e-5, s)
    torch.cuda.synchronize()
    return st, out


def test_noalloc_groupnorm_matches_workspace_variant():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    for nsamples, rows, c, groups in [(3, 5000, 256, 32), (2, 700, 64, 8)]:       # the second call is smaller
        x = (torch.randn(nsamples * rows, c, generator=g) * 1.7 + 0.3).to(torch.bfloat16).to(dev)
        gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(dev)
        beta = (0.1 * torch.randn(c, generator=g)).to(dev)
        st_a, a = _run("ws", x, gamma, beta, nsamples, rows, c, groups, True)
        st_b, b = _run("noalloc", x, gamma, beta, nsamples, rows, c, groups, False)
        assert st_a == 0 and st_b == 0, (st_a, st_b)
        assert torch.isfinite(b.float()).all()
        assert torch.equal(a, b), (nsamples, rows, c)
        # against plain torch, fp32 (bf16 output rounding: 1e-2)
        xf = x.float().reshape(nsamples, rows, groups, c // groups)
        mean = xf.mean(dim=(1, 3), keepdim=True)
        var = xf.var(dim=(1, 3), unbiased=False, keepdim=True)
        y = ((xf - mean) / torch.sqrt(var + 1e-5)).reshape(nsamples, rows, c) * gamma + beta
        y = torch.nn.functional.silu(y).reshape(-1, c)
        assert float((b.float() - y).norm() / y.norm()) < 1e-2


def test_noalloc_groupnorm_rejects_oversized_input():
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    nsamples, rows, c = 64, 64 * 512, 2048          # 64 samples x 64 splits x 256 pieces x 8 B = 8 MiB of partial sums
    assert lib.vdmi_groupnorm_workspace_bytes(nsamples, rows, c) > SCRATCH
    x = torch.zeros(8, 2048, device=dev, dtype=torch.bfloat16)      # never touched: the size check comes first
    gb = torch.zeros(c, device=dev)
    st = lib.vdmi_groupnorm_silu_noalloc(x.data_ptr(), x.data_ptr(), nsamples, rows, c, 32, gb.data_ptr(), gb.data_ptr(),
                                         1e-5, torch.cuda.current_stream().cuda_stream)
    assert st == -6
    assert b"scratch" in lib.vdmi_strerror(st)


def test_noalloc_groupnorm_under_graph_capture():
    """What the entry point is for: after one eager call on the device (the address lookup), the call is captured
    into a graph and replayed on new input; the replay equals the workspace variant bit for bit."""
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(5)
    nsamples, rows, c, groups = 2, 1500, 128, 8
    x = (torch.randn(nsamples * rows, c, generator=g)).to(torch.bfloat16).to(dev)
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(dev)
    beta = (0.1 * torch.randn(c, generator=g)).to(dev)
    st, _ = _run("noalloc", x, gamma, beta, nsamples, rows, c, groups, False)       # eager first call
    assert st == 0
    out = torch.empty_like(x)
    status = []
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        status.append(lib.vdmi_groupnorm_silu_noalloc(x.data_ptr(), out.data_ptr(), nsamples, rows, c, groups,
                                                      gamma.data_ptr(), beta.data_ptr(), 1e-5,
                                                      torch.cuda.current_stream().cuda_stream))
    assert status == [0]
    x.copy_((torch.randn(nsamples * rows, c, generator=g) * 2.0).to(torch.bfloat16))
    graph.replay()
    torch.cuda.synchronize()
    st, want = _run("ws", x, gamma, beta, nsamples, rows, c, groups, True)
    assert st == 0 and torch.equal(out, want)
