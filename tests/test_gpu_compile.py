"""torch.compile over the public ops (the reference's benchmarks compile their Conch op: static_scaled_int8_quant_benchmark.py:130).

While a graph is captured the ops route through `torch.library` custom ops (conch_amd/ops/quantization/_compile.py): one opaque
node each, `fullgraph=True` holds, results equal the eager call bit for bit (it IS the same C-ABI call)."""

from __future__ import annotations

import pytest
import torch

from conch_amd.ops.quantization.fp8 import scaled_fp8_quant
from conch_amd.ops.quantization.gemm import mixed_precision_gemm, scaled_gemm
from conch_amd.ops.quantization.int8 import scaled_int8_quant

pytestmark = pytest.mark.gpu


def _quant_then_gemm(x, w_q, scale_x, scale_w):
    xq, _ = scaled_int8_quant(x, scale_x)
    return scaled_gemm(xq, w_q, scale_x.reshape(1, 1), scale_w, torch.bfloat16)


@pytest.mark.parametrize("backend", ["eager", "inductor"])
def test_compiled_quant_and_gemm_equal_eager(backend):
    torch.manual_seed(0)
    m, k, n = 48, 512, 384
    x = (torch.rand(m, k, device="cuda", dtype=torch.float16) * 200 - 100)
    w_q = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda").T
    sx, sw = torch.tensor([0.9], device="cuda"), 0.01 * torch.rand(n, 1, device="cuda")
    want = _quant_then_gemm(x, w_q, sx, sw)
    try:
        fn = torch.compile(_quant_then_gemm, backend=backend, fullgraph=True)
        got = fn(x, w_q, sx, sw)
    except Exception as exc:  # noqa: BLE001
        if backend == "inductor":
            pytest.skip(f"Inductor backend unavailable on this box: {str(exc)[:120]}")
        raise
    assert torch.equal(got, want)
    assert torch.equal(fn(x, w_q, sx, sw), want)  # second call: the cached graph


def test_compiled_fp8_quant_and_mixed_gemm_equal_eager():
    torch.manual_seed(1)
    x = torch.rand(32, 256, device="cuda", dtype=torch.float16)
    s = torch.tensor([0.01], device="cuda")
    for fnuz in (False, True):
        dt = torch.float8_e4m3fnuz if fnuz else torch.float8_e4m3fn
        want, _ = scaled_fp8_quant(x, s, output_dtype=dt)
        got, _ = torch.compile(lambda t, sc: scaled_fp8_quant(t, sc, output_dtype=dt), backend="eager", fullgraph=True)(x, s)  # noqa: B023
        assert torch.equal(got.view(torch.uint8), want.view(torch.uint8))
    k, n = 256, 128
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
    ws = (0.05 * torch.rand(k // 128, n, device="cuda") + 0.01).to(torch.float16)
    want = mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
    got = torch.compile(lambda t: mixed_precision_gemm(t, wq, ws, None, 4, 8, 128), backend="eager", fullgraph=True)(x)
    assert torch.equal(got, want)
