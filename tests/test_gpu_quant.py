"""GPU parity: static int8 / fp8 quantisation kernels vs the CPU oracle (bit-exact).

Mirrors the reference's matrices (tests/int8_quant_kernels_test.py:16-50,
tests/fp8_quant_kernels_test.py:16-57): dtype x hidden x tokens x scale = 120 cases each.  The
reference allows atol=1 (int8) / default assert_close on dequantised values (fp8) because Triton's
cast differs from torch's; the HIP kernels restate the oracle's arithmetic, so the bar here is
bit-exact codes.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle
from conch_amd.ops.quantization.fp8 import scaled_fp8_quant, static_scaled_fp8_quant
from conch_amd.ops.quantization.int8 import scaled_int8_quant, static_scaled_int8_quant
from conch_amd.third_party.vllm.utils import seed_everything
from tests.conftest import DT, from_bits, to_bits

pytestmark = pytest.mark.gpu

DTYPES = [torch.half, torch.bfloat16, torch.float]
HIDDEN = [16, 67, 768, 5137, 8193]
TOKENS = [1, 7, 83, 4096]
SCALES = [0.1, 2.1]
FP8 = {"fn": (oracle.FP8_E4M3FN, torch.float8_e4m3fn), "fnuz": (oracle.FP8_E4M3FNUZ, torch.float8_e4m3fnuz)}


@pytest.mark.parametrize("num_tokens", TOKENS)
@pytest.mark.parametrize("hidden_size", HIDDEN)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("scale", SCALES)
@torch.inference_mode()
def test_static_scaled_int8_quant(num_tokens, hidden_size, dtype, scale):
    seed_everything(0)
    x = torch.rand(num_tokens, hidden_size, dtype=dtype, device="cuda") * 1000
    scale_arg = torch.tensor([scale], dtype=torch.float32, device="cuda")
    out, ret_scale = scaled_int8_quant(x, scale_arg)
    assert ret_scale is scale_arg and out.dtype == torch.int8 and out.shape == x.shape
    ref = oracle.scaled_int8_quant_ref(x.cpu(), scale_arg.cpu())
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("flav", list(FP8))
@pytest.mark.parametrize("num_tokens", TOKENS)
@pytest.mark.parametrize("hidden_size", HIDDEN)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("scale", SCALES)
@torch.inference_mode()
def test_static_scaled_fp8_quant(flav, num_tokens, hidden_size, dtype, scale):
    name, tdt = FP8[flav]
    seed_everything(0)
    x = torch.rand(num_tokens, hidden_size, dtype=dtype, device="cuda")
    scale_arg = torch.tensor([scale], dtype=torch.float32, device="cuda")
    out, ret_scale = scaled_fp8_quant(x, scale_arg, output_dtype=tdt)
    assert ret_scale is scale_arg and out.dtype == tdt
    ref = oracle.scaled_fp8_quant_ref(x.cpu(), scale_arg.cpu(), name)
    np.testing.assert_array_equal(to_bits(out), to_bits(ref))


def test_default_fp8_flavour_is_ocp_on_gfx950():
    x = torch.rand(4, 64, device="cuda")
    out, _ = scaled_fp8_quant(x, torch.tensor([1.0], device="cuda"))
    assert out.dtype == torch.float8_e4m3fn


@pytest.mark.parametrize("dname", list(DT))
@pytest.mark.parametrize("sname", ["one", "s2p1", "zdim"])
def test_quant_edge_vectors_from_reference(golden, dname, sname):
    """Ties, negative fractions, overflow both ways, subnormals: golden outputs of the reference (`zdim`: a 0-dim scale)."""
    gi, gf = golden("quant_int8"), golden("quant_fp8")
    x = from_bits(gi[f"edge_x_{dname}_{sname}"], DT[dname]).cuda()
    s = torch.from_numpy(np.asarray(gi[f"edge_s_{dname}_{sname}"])).cuda()
    assert s.dim() == (0 if sname == "zdim" else 1)
    q, _ = scaled_int8_quant(x, s)
    np.testing.assert_array_equal(q.cpu().numpy(), gi[f"edge_q_{dname}_{sname}"])
    for flav, (_, tdt) in FP8.items():
        q8, _ = scaled_fp8_quant(x, s, output_dtype=tdt)
        np.testing.assert_array_equal(to_bits(q8), gf[f"edge_q{flav}_{dname}_{sname}"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("scale", [0.3, 2.1, 0.0123])
@pytest.mark.parametrize("shape", [(83, 768), (7, 67), (33, 200)])
def test_int8_quant_with_a_0dim_scale_rounds_the_product_like_torch(dtype, scale, shape):
    """A 0-dim scale does not take part in torch's type promotion: the reference oracle's `x * scale.reciprocal()`
    (conch/reference/quantization/int8.py:16) is rounded to x's dtype before the clamp.  The kernel reproduces that (and the
    fp32 product for every other one-element shape); the two differ on a few elements in a thousand, so the test also
    requires that they DO differ somewhere for the 16-bit dtypes -- the path is really taken.  fp8 always multiplies in fp32
    (fp8.py:16 casts x first)."""
    seed_everything(0)
    x = (torch.rand(*shape, dtype=torch.float32) * 400 - 150).to(dtype)
    if shape == (33, 200):
        x = torch.cat([x, x], dim=1)[:, 3:203]  # row-strided, unaligned: the per-row kernel
    s0, s1 = torch.tensor(scale, dtype=torch.float32), torch.tensor([scale], dtype=torch.float32)
    q0, ret = scaled_int8_quant(x.cuda(), s0.cuda())
    q1, _ = scaled_int8_quant(x.cuda(), s1.cuda())
    assert ret.dim() == 0
    assert torch.equal(q0.cpu(), oracle.scaled_int8_quant_ref(x, s0))
    assert torch.equal(q1.cpu(), oracle.scaled_int8_quant_ref(x, s1))
    if dtype != torch.float and shape == (83, 768):
        assert not torch.equal(q0, q1)
    for name, tdt in FP8.values():
        q8, _ = scaled_fp8_quant(x.cuda() / 100, s0.cuda(), output_dtype=tdt)
        np.testing.assert_array_equal(to_bits(q8), to_bits(oracle.scaled_fp8_quant_ref(x / 100, s0, name)))


def test_quant_golden_grid_subset(golden):
    gi, gf = golden("quant_int8"), golden("quant_fp8")
    for dname in DT:
        for tokens, hidden in ((7, 67), (83, 768)):
            key = f"{dname}_t{tokens}_h{hidden}_s2.1"
            s = torch.tensor([2.1], device="cuda")
            q, _ = scaled_int8_quant(from_bits(gi[f"x_{key}"], DT[dname]).cuda(), s)
            np.testing.assert_array_equal(q.cpu().numpy(), gi[f"q_{key}"])
            x8 = from_bits(gf[f"x_{key}"], DT[dname]).cuda()
            for flav, (_, tdt) in FP8.items():
                q8, _ = scaled_fp8_quant(x8, s, output_dtype=tdt)
                np.testing.assert_array_equal(to_bits(q8), gf[f"q{flav}_{key}"])


def test_fp8_all_codes_roundtrip_and_dense_sweep():
    """Every finite fp8 value re-encodes to itself; a dense fp32 sweep matches the oracle."""
    one = torch.tensor([1.0], device="cuda")
    for name, tdt in FP8.values():
        codes = np.arange(256, dtype=np.uint8)
        vals = oracle.decode_fp8(codes, name)
        finite = ~np.isnan(vals)
        x = torch.from_numpy(vals[finite]).reshape(1, -1).cuda()
        q, _ = scaled_fp8_quant(x, one, output_dtype=tdt)
        ref = oracle.scaled_fp8_quant_ref(x.cpu(), one.cpu(), name)
        np.testing.assert_array_equal(to_bits(q), to_bits(ref))
        gen = torch.Generator().manual_seed(1)
        sweep = (torch.randn(64, 4099, generator=gen) * torch.exp(torch.empty(64, 4099).uniform_(-14, 7, generator=gen)))
        q, _ = scaled_fp8_quant(sweep.cuda(), one, output_dtype=tdt)
        ref = oracle.scaled_fp8_quant_ref(sweep, one.cpu(), name)
        np.testing.assert_array_equal(to_bits(q), to_bits(ref))


def test_quant_empty_and_row_strided_inputs():
    s = torch.tensor([0.7], device="cuda")
    e, _ = scaled_int8_quant(torch.empty(0, 16, device="cuda"), s)
    assert e.shape == (0, 16)
    base = torch.rand(33, 200, device="cuda", dtype=torch.float16) * 300
    view = base[:, 3:131]  # row stride 200, unaligned start -> strided kernel
    out = torch.empty(33, 128, dtype=torch.int8, device="cuda")
    static_scaled_int8_quant(out, view, s)
    assert torch.equal(out.cpu(), oracle.scaled_int8_quant_ref(view.cpu(), s.cpu()))
    out8 = torch.empty(33, 128, dtype=torch.float8_e4m3fn, device="cuda")
    static_scaled_fp8_quant(out8, view, s)
    np.testing.assert_array_equal(to_bits(out8), to_bits(oracle.scaled_fp8_quant_ref(view.cpu(), s.cpu())))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize(("tokens", "hidden", "wide"), [(33, 4096, 12288), (5, 4112, 8224), (300, 16, 48), (2, 11008, 11008 + 64)])
def test_static_quant_of_an_aligned_column_slice(dt, tokens, hidden, wide):
    """A 16-byte aligned column slice of a wider activation (one part of a fused qkv projection) -- rows strided, not flat: the
    vectorised row kernel (contiguous 1 KiB per load instruction, chunks of 4096 elements with a tail), into a contiguous and
    into a row-strided result, int8 (fp32 product and the 0-dim-scale x.dtype product) and both fp8 flavours, bit for bit."""
    seed_everything(5)
    base = ((torch.rand(tokens, wide, device="cuda") - 0.5) * 600).to(dt)
    off = 16 if wide - hidden >= 16 else 0
    view = base[:, off:off + hidden]
    s = torch.tensor([0.7], device="cuda")
    out = torch.empty(tokens, hidden, dtype=torch.int8, device="cuda")
    static_scaled_int8_quant(out, view, s)
    assert torch.equal(out.cpu(), oracle.scaled_int8_quant_ref(view.cpu(), s.cpu()))
    big = torch.zeros(tokens, hidden + 24, dtype=torch.int8, device="cuda")
    static_scaled_int8_quant(big[:, 8:8 + hidden], view, s)
    assert torch.equal(big[:, 8:8 + hidden].cpu(), out.cpu()) and not bool(big[:, :8].any()) and not bool(big[:, 8 + hidden:].any())
    q0, _ = scaled_int8_quant(view, torch.tensor(0.7, device="cuda"))  # 0-dim: the product is rounded to x.dtype first
    assert torch.equal(q0.cpu(), oracle.scaled_int8_quant_ref(view.cpu(), torch.tensor(0.7)))
    for f8 in (torch.float8_e4m3fn, torch.float8_e4m3fnuz):
        out8 = torch.empty(tokens, hidden, dtype=f8, device="cuda")
        static_scaled_fp8_quant(out8, view, s)
        np.testing.assert_array_equal(to_bits(out8), to_bits(oracle.scaled_fp8_quant_ref(view.cpu(), s.cpu(), oracle.FP8_E4M3FNUZ if f8 is torch.float8_e4m3fnuz else oracle.FP8_E4M3FN)))


def test_quant_c1_config_bit_exact():
    """BASELINE config C1: 4096x4096 fp16, scale 2.1 (int8 bench :124-125)."""
    seed_everything(0)
    x = torch.rand(4096, 4096, dtype=torch.float16, device="cuda") * 1000
    s = torch.tensor([2.1], dtype=torch.float32, device="cuda")
    q, _ = scaled_int8_quant(x, s)
    assert torch.equal(q.cpu(), oracle.scaled_int8_quant_ref(x.cpu(), s.cpu()))


# ---------------------------------------------------------------------------------------------
# dynamic per-token quantisation: scale=None
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("num_tokens", TOKENS)
@pytest.mark.parametrize("hidden_size", [16, 67, 768, 1024, 2064, 5137, 8192, 8193, 16400])
@pytest.mark.parametrize("dtype", DTYPES)
@torch.inference_mode()
def test_dynamic_int8_quant(num_tokens, hidden_size, dtype):
    seed_everything(0)
    x = ((torch.rand(num_tokens, hidden_size, dtype=torch.float32, device="cuda") - 0.4) * 1000).to(dtype)
    if num_tokens > 2:
        x[1] = 0  # an all-zero row quantises to zeros with scale 1
    q, s = scaled_int8_quant(x)
    assert q.dtype == torch.int8 and q.shape == x.shape and s.shape == (num_tokens, 1) and s.dtype == torch.float32
    ref_q, ref_s = oracle.dynamic_scaled_int8_quant_ref(x.cpu())
    assert torch.equal(s.cpu().view(torch.int32), ref_s.view(torch.int32))
    assert torch.equal(q.cpu(), ref_q)


@pytest.mark.parametrize("flav", list(FP8))
@pytest.mark.parametrize("num_tokens", [1, 7, 83])
@pytest.mark.parametrize("hidden_size", [16, 67, 768, 2064, 5137, 8200])
@pytest.mark.parametrize("dtype", DTYPES)
@torch.inference_mode()
def test_dynamic_fp8_quant(flav, num_tokens, hidden_size, dtype):
    name, tdt = FP8[flav]
    seed_everything(0)
    x = ((torch.rand(num_tokens, hidden_size, dtype=torch.float32, device="cuda") - 0.4) * 3).to(dtype)
    q, s = scaled_fp8_quant(x, None, output_dtype=tdt)
    ref_q, ref_s = oracle.dynamic_scaled_fp8_quant_ref(x.cpu(), name)
    assert torch.equal(s.cpu().view(torch.int32), ref_s.view(torch.int32))
    np.testing.assert_array_equal(to_bits(q), to_bits(ref_q))


@pytest.mark.parametrize(("dname", "tokens", "hidden"), [(d, t, h) for d in DT for t, h in ((1, 16), (7, 67), (5, 768), (3, 2064), (2, 8200))])
def test_dynamic_quant_golden(golden, dname, tokens, hidden):
    """Rows of the REAL reference's static oracle at scale = absmax / QMAX (tests/golden/make_golden_r2.py)."""
    g = golden("quant_dynamic")
    key = f"{dname}_t{tokens}_h{hidden}"
    x = from_bits(g[f"x_{key}"], DT[dname]).cuda()
    q, s = scaled_int8_quant(x)
    np.testing.assert_array_equal(q.cpu().numpy(), g[f"qi8_{key}"])
    np.testing.assert_array_equal(s.cpu().reshape(-1).numpy().view(np.uint32), g[f"si8_{key}"].view(np.uint32))
    for flav, (_, tdt) in FP8.items():
        q8, s8 = scaled_fp8_quant(x, None, output_dtype=tdt)
        np.testing.assert_array_equal(to_bits(q8), g[f"q{flav}_{key}"])
        np.testing.assert_array_equal(s8.cpu().reshape(-1).numpy().view(np.uint32), g[f"s{flav}_{key}"].view(np.uint32))


def test_dynamic_quant_3d_and_strided_input():
    x = torch.randn(3, 5, 96, device="cuda", dtype=torch.float16)
    q, s = scaled_int8_quant(x)
    assert q.shape == x.shape and s.shape == (3, 5, 1)
    ref_q, ref_s = oracle.dynamic_scaled_int8_quant_ref(x.cpu())
    assert torch.equal(q.cpu(), ref_q) and torch.equal(s.cpu(), ref_s)
    wide = torch.randn(9, 256, device="cuda", dtype=torch.bfloat16)
    view = wide[:, 32:160]  # row stride 256, 16-byte aligned start: the vector kernel with a row stride
    q, s = scaled_int8_quant(view)
    ref_q, ref_s = oracle.dynamic_scaled_int8_quant_ref(view.cpu())
    assert torch.equal(q.cpu(), ref_q) and torch.equal(s.cpu(), ref_s)


@pytest.mark.parametrize("kind", ["int8", "fp8"])
def test_dynamic_quant_feeds_scaled_gemm(kind):
    """End to end: activations quantised per token on the device, their (M, 1) scales handed to scaled_gemm as scale_a."""
    from conch_amd.ops.quantization.gemm import scaled_gemm

    seed_everything(0)
    m, k, n = 96, 1024, 256
    x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.float32)
    if kind == "int8":
        xq, sx = scaled_int8_quant(x)
        ws = w.abs().amax(dim=1, keepdim=True) / 127
        wq = (w / ws).round().clamp(-128, 127).to(torch.int8)
    else:
        xq, sx = scaled_fp8_quant(x)
        ws = w.abs().amax(dim=1, keepdim=True) / 448
        wq = (w / ws).to(torch.float8_e4m3fn)
    out = scaled_gemm(xq, wq.T, sx, ws, torch.bfloat16)
    ref = oracle.scaled_gemm_ref(xq.cpu(), wq.cpu().T, sx.cpu(), ws.cpu(), torch.bfloat16)
    if kind == "int8":
        assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))
    else:
        torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=0, atol=2 * 2.0**-7 * ref.float().abs().max().item())
    # and the quantised pipeline approximates the unquantised product
    exact = x.float() @ w.T
    rel = (out.float() - exact).norm() / exact.norm()
    assert rel < (0.02 if kind == "int8" else 0.06)
