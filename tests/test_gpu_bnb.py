"""GPU parity: bitsandbytes-style blockwise NF4 / FP4 / 8-bit quantise + dequantise + 4-bit matmul vs the
pinned oracle (oracle/bnb.py) and the reference's golden outputs.  Matrices follow tests/quantize_blockwise_test.py and
tests/dequantize_blockwise_test.py of the reference (blocksize x size multiplier x dtype x quant type), extended to every
supported blocksize; the reference compares with assert_close, here codes, absmax and dequantised values are bit-exact."""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle
from conch_amd.kernels.quantization.bitsandbytes.dequantize_blockwise import dequantize_blockwise_launcher
from conch_amd.kernels.quantization.bitsandbytes.quantize_blockwise import quantize_blockwise_launcher
from conch_amd.ops.quantization.bitsandbytes.functional import (
    SUPPORTED_BLOCKSIZES,
    QuantState,
    _create_dynamic_map,
    dequantize_4bit,
    dequantize_blockwise,
    matmul_4bit,
    quantize_4bit,
    quantize_blockwise,
)
from conch_amd.third_party.vllm.utils import seed_everything
from tests.conftest import DT, from_bits, to_bits

pytestmark = pytest.mark.gpu

QTYPES = ["nf4", "fp4", "fp8"]


@pytest.mark.parametrize("blocksize", SUPPORTED_BLOCKSIZES)
@pytest.mark.parametrize("mult", [2.5, 6, 0.3, 37.77])
@pytest.mark.parametrize("dname", list(DT))
@pytest.mark.parametrize("qt", QTYPES)
def test_quantize_blockwise(blocksize, mult, dname, qt):
    seed_everything(2)
    n = max(2, int(blocksize * mult))
    x = torch.randn((n,), dtype=DT[dname])
    if qt == "fp4":
        x = x.uniform_(-1.0, 1.0)
    code = _create_dynamic_map() if qt == "fp8" else None
    want_q, want_absmax = oracle.quantize_blockwise_ref(x, blocksize, qt, code)
    absmax = torch.empty(want_absmax.shape, dtype=DT[dname], device="cuda")
    out = torch.zeros(want_q.shape, dtype=torch.uint8, device="cuda")
    quantize_blockwise_launcher(x.cuda(), absmax, out, None if code is None else code.cuda(), blocksize, n, qt)
    np.testing.assert_array_equal(to_bits(absmax), to_bits(want_absmax))
    np.testing.assert_array_equal(out.cpu().numpy(), want_q.numpy())


@pytest.mark.parametrize("blocksize", SUPPORTED_BLOCKSIZES)
@pytest.mark.parametrize("mult", [2.5, 6, 0.5, 37.75])
@pytest.mark.parametrize(("dname", "aname"), [("f32", "f32"), ("f16", "f16"), ("bf16", "bf16"), ("f16", "f32"), ("bf16", "f32")])
@pytest.mark.parametrize("qt", QTYPES)
def test_dequantize_blockwise(blocksize, mult, dname, aname, qt):
    seed_everything(0)
    n = int(blocksize * mult) & ~1
    nin = n if qt == "fp8" else n // 2
    xq = torch.randint(0, 255, (nin,), dtype=torch.uint8)
    absmax = torch.randn((-(-n // blocksize),), dtype=DT[aname])
    code = _create_dynamic_map() if qt == "fp8" else None
    want = oracle.dequantize_blockwise_ref(xq, absmax, n, blocksize, qt, DT[dname], code)
    out = torch.empty((n,), dtype=DT[dname], device="cuda")
    dequantize_blockwise_launcher(xq.cuda(), absmax.cuda(), out, None if code is None else code.cuda(), blocksize, n, qt)
    np.testing.assert_array_equal(to_bits(out), to_bits(want))


@pytest.mark.parametrize(("qt", "blocksize", "mult", "dname"), [(q, b, m, d) for q in QTYPES for b in (64, 1024) for m in (2.5, 6) for d in DT])
def test_golden_from_reference(golden, qt, blocksize, mult, dname):
    """Outputs of the REAL reference (conch/reference/quantization/bitsandbytes) on its own tests' inputs."""
    g = golden("bnb_blockwise")
    key = f"{qt}_b{blocksize}_m{mult}_{dname}"
    n = int(blocksize * mult)
    code = torch.from_numpy(g["dynamic_map"].copy()).cuda() if qt == "fp8" else None
    x = from_bits(g[f"qx_{key}"], DT[dname]).cuda()
    absmax = torch.empty((-(-n // blocksize),), dtype=DT[dname], device="cuda")
    out = torch.zeros(g[f"qout_{key}"].shape, dtype=torch.uint8, device="cuda")
    quantize_blockwise_launcher(x, absmax, out, code, blocksize, n, qt)
    np.testing.assert_array_equal(to_bits(absmax), g[f"qabsmax_{key}"])
    np.testing.assert_array_equal(out.cpu().numpy(), g[f"qout_{key}"])
    d = torch.empty((n,), dtype=DT[dname], device="cuda")
    dequantize_blockwise_launcher(torch.from_numpy(g[f"dx_{key}"].copy()).cuda(), from_bits(g[f"dabsmax_{key}"], DT[dname]).cuda(), d, code,
                                  blocksize, n, qt)
    np.testing.assert_array_equal(to_bits(d), g[f"dout_{key}"])


@pytest.mark.parametrize("qt", ["nf4", "fp4"])
def test_quantize_edges_golden(golden, qt):
    """All-zero block (NaN after the division: every comparison false), values on the thresholds, an odd tail."""
    g = golden("bnb_blockwise")
    x = torch.from_numpy(g[f"edge_x_{qt}"].copy()).cuda()
    n = x.numel()
    absmax = torch.empty((3,), dtype=torch.float32, device="cuda")
    out = torch.zeros(((n + 1) // 2,), dtype=torch.uint8, device="cuda")
    quantize_blockwise_launcher(x, absmax, out, None, 64, n, qt)
    np.testing.assert_array_equal(absmax.cpu().numpy().view(np.uint32), g[f"edge_absmax_{qt}"].view(np.uint32))
    np.testing.assert_array_equal(out.cpu().numpy(), g[f"edge_q_{qt}"])


def test_dynamic_map_matches_reference(golden):
    np.testing.assert_array_equal(_create_dynamic_map().numpy().view(np.uint32), golden("bnb_blockwise")["dynamic_map"].view(np.uint32))


@pytest.mark.parametrize("qt", ["nf4", "fp4"])
@pytest.mark.parametrize("compress", [False, True])
@pytest.mark.parametrize("dname", ["f16", "bf16", "f32"])
def test_functional_round_trip(qt, compress, dname):
    """quantize_4bit -> dequantize_4bit through the public API (QuantState, nested statistics): each step equals the oracle."""
    seed_everything(1)
    w = torch.randn(384, 1024, dtype=DT[dname])
    q, state = quantize_4bit(w.cuda(), blocksize=64, compress_statistics=compress, quant_type=qt)
    want_q, want_absmax = oracle.quantize_blockwise_ref(w, 64, qt, None, absmax_dtype=torch.float32)
    assert q.shape == (384 * 1024 // 2, 1) and state.shape == w.shape and state.dtype == w.dtype and state.nested == compress
    np.testing.assert_array_equal(q.cpu().view(-1).numpy(), want_q.numpy())
    if compress:
        code = _create_dynamic_map()
        # the offset is a device-side fp32 mean (its reduction order is torch's business): take it from the state
        assert abs(state.offset - want_absmax.mean().item()) <= 1e-5 * want_absmax.abs().max().item()
        stats_q, stats_absmax = oracle.quantize_blockwise_ref(want_absmax - torch.tensor(state.offset, dtype=torch.float32), 256, "fp8", code)
        np.testing.assert_array_equal(state.absmax.cpu().numpy(), stats_q.numpy())
        want_absmax = oracle.dequantize_blockwise_ref(stats_q, stats_absmax, want_absmax.numel(), 256, "fp8", torch.float32, code) + state.offset
    else:
        np.testing.assert_array_equal(to_bits(state.absmax), to_bits(want_absmax))
    back = dequantize_4bit(q, state, blocksize=64, quant_type=qt)
    want = oracle.dequantize_blockwise_ref(want_q, want_absmax, w.numel(), 64, qt, DT[dname]).reshape(w.shape)
    np.testing.assert_array_equal(to_bits(back), to_bits(want))
    # and the codes do approximate the weights
    assert (back.float().cpu() - w.float()).abs().mean().item() < (0.12 if qt == "nf4" else 0.16)


def test_blockwise_8bit_round_trip_via_ops():
    seed_everything(3)
    x = torch.randn(5000, dtype=torch.float32)
    code = _create_dynamic_map()
    q, state = quantize_blockwise(x.cuda(), code=code.cuda(), blocksize=256, quant_type="fp8")
    want_q, want_absmax = oracle.quantize_blockwise_ref(x, 256, "fp8", code, absmax_dtype=torch.float32)
    np.testing.assert_array_equal(q.cpu().numpy(), want_q.numpy())
    back = dequantize_blockwise(q, state, blocksize=256, quant_type="fp8")
    want = oracle.dequantize_blockwise_ref(want_q, want_absmax, 5000, 256, "fp8", torch.float32, code)
    np.testing.assert_array_equal(to_bits(back), to_bits(want))


@pytest.mark.parametrize("qt", ["nf4", "fp4"])
@pytest.mark.parametrize("dname", ["f16", "bf16"])
@pytest.mark.parametrize(("m", "k", "n", "blocksize"), [(16, 1024, 768, 64), (300, 512, 1000, 128), (1024, 4096, 1376, 64),
                                                         # few tiles and a long K: the 16-bit tile kernel's split-K form (round 4)
                                                         (300, 4096, 520, 64), (512, 11008, 256, 128)])
def test_matmul_4bit(qt, dname, m, k, n, blocksize):
    """x @ dequantize_4bit(W).T: the dequantised weights are bit-exact (previous tests); the product is compared with the same
    product of the oracle's dequantised weights in fp64, to one output rounding plus fp32 accumulation order."""
    seed_everything(4)
    dt = DT[dname]
    w = torch.randn(n, k, dtype=dt)
    x = (torch.rand(m, k) - 0.5).to(dt)
    q, state = quantize_4bit(w.cuda(), blocksize=blocksize, quant_type=qt)
    got = matmul_4bit(x.cuda(), q, state).cpu()
    wq, am = oracle.quantize_blockwise_ref(w, blocksize, qt, None, absmax_dtype=torch.float32)
    wd = oracle.dequantize_blockwise_ref(wq, am, n * k, blocksize, qt, dt).reshape(n, k)
    want = (x.double() @ wd.double().T)
    eps = 2.0**-10 if dt == torch.float16 else 2.0**-7
    bound = eps * want.abs() + (k * 2.0**-24) * (x.double().abs() @ wd.double().abs().T) + 1e-30
    assert ((got.double() - want).abs() <= bound).all()
    assert got.dtype == dt and got.shape == (m, n)


@pytest.mark.parametrize("qt", ["nf4", "fp4"])
@pytest.mark.parametrize(("dname", "absmax_dt"), [("f16", torch.float32), ("bf16", torch.float32), ("f16", torch.float16)])
@pytest.mark.parametrize(("m", "k", "n", "blocksize"), [(1, 1024, 64, 64), (16, 2048, 520, 128), (33, 4096, 1376, 64), (64, 1152, 256, 128),
                                                         (100, 1024, 260, 1024), (8, 11008, 512, 64), (128, 192, 64, 64), (200, 1024, 256, 64)])
def test_matmul_4bit_decode_kernel(qt, dname, absmax_dt, m, k, n, blocksize):
    """The decode-batch kernel (M <= 128: packed codes straight into MFMA operands, no materialised weight; forced with
    variant 4 so that a contract change cannot silently skip it): per element against the fp64 product of the oracle's
    dequantised weights, and against the library's dequantise-first path (same weights bit for bit, another summation order)."""
    from conch_amd import _C

    seed_everything(5)
    dt = DT[dname]
    w = torch.randn(n, k, dtype=dt)
    x = (torch.rand(m, k) - 0.5).to(dt)
    wq, am = oracle.quantize_blockwise_ref(w, blocksize, qt, None, absmax_dtype=absmax_dt)
    state = QuantState(absmax=am.cuda(), shape=w.shape, blocksize=blocksize, quant_type=qt, dtype=dt)
    q = wq.cuda()
    try:
        _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
        got = matmul_4bit(x.cuda(), q, state).cpu()
        _C.set_gemm_variant(_C.VARIANT_MFMA_SIMPLE)
        first = matmul_4bit(x.cuda(), q, state).cpu()
        _C.set_gemm_variant(_C.VARIANT_AUTO)
        auto = matmul_4bit(x.cuda(), q, state).cpu()
    finally:
        _C.set_gemm_variant(_C.VARIANT_AUTO)
    wd = oracle.dequantize_blockwise_ref(wq, am, n * k, blocksize, qt, dt).reshape(n, k)
    want = x.double() @ wd.double().T
    eps = 2.0**-10 if dt == torch.float16 else 2.0**-7
    bound = eps * want.abs() + (k * 2.0**-24) * (x.double().abs() @ wd.double().abs().T) + 1e-30
    assert ((got.double() - want).abs() <= bound).all()
    assert ((first.double() - want).abs() <= bound).all()
    assert torch.equal(auto, got) or torch.equal(auto, first)  # auto = the decode kernel up to 64 rows, a cost rule above


@pytest.mark.parametrize("qt", ["nf4", "fp4"])
@pytest.mark.parametrize(("dname", "absmax_dt"), [("f16", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize(("m", "k", "n", "blocksize"), [(1, 1024, 64, 64), (16, 2048, 520, 128), (20, 4096, 1376, 64), (32, 1152, 260, 128),
                                                         (8, 11008, 512, 64), (3, 192, 68, 64), (24, 2048, 256, 2048),
                                                         (64, 2048, 520, 64), (48, 1920, 132, 64), (130, 4096, 256, 128)])
def test_matmul_4bit_decode_fetch_orders_are_bit_identical(qt, dname, absmax_dt, m, k, n, blocksize):
    """The packed codes reach the MFMA operands through LDS-DMA (four lanes per 64-byte row piece, read back per lane; the
    64-row form stages seven of its eight pairs of steps that way and the last through registers) and the absmax of 64-element blocks through four wide loads and a cross-lane gather; CONCH_TUNE_SKINNY_GATHER = 1 is
    the register path in MFMA operand order.  Same bytes in the same operand registers: bit-identical, ragged K / N included."""
    from conch_amd import _C

    seed_everything(7)
    dt = DT[dname]
    w = torch.randn(n, k, dtype=dt)
    x = (torch.rand(m, k) - 0.5).to(dt)
    wq, am = oracle.quantize_blockwise_ref(w, blocksize, qt, None, absmax_dtype=absmax_dt)
    state = QuantState(absmax=am.cuda(), shape=w.shape, blocksize=blocksize, quant_type=qt, dtype=dt)
    q = wq.cuda()
    try:
        _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
        _C.set_tuning(_C.TUNE_SKINNY_GATHER, 1)
        regs = matmul_4bit(x.cuda(), q, state)
        _C.set_tuning(_C.TUNE_SKINNY_GATHER, 0)
        for _ in range(2):
            assert torch.equal(matmul_4bit(x.cuda(), q, state), regs)
    finally:
        _C.set_gemm_variant(_C.VARIANT_AUTO)
        _C.set_tuning(_C.TUNE_SKINNY_GATHER, 0)


@pytest.mark.parametrize("offset", [1, 4, 8, 16])
def test_matmul_4bit_takes_a_weight_at_any_byte_offset(offset):
    """The packed weight as a slice of a larger byte tensor (storage offsets of 1, 4, 8, 16 bytes: the LDS-DMA path wants 16-byte
    pieces, the register path dwords): the same product as from an aligned copy, bit for bit, at decode and at tile sizes."""
    from conch_amd import _C

    seed_everything(8)
    n, k = 260, 2048
    w = torch.randn(n, k, dtype=torch.float16)
    wq, am = oracle.quantize_blockwise_ref(w, 64, "nf4", None, absmax_dtype=torch.float32)
    flat = wq.reshape(-1).cuda()
    buf = torch.zeros(flat.numel() + 64, dtype=torch.uint8, device="cuda")
    buf[offset:offset + flat.numel()] = flat
    shifted = buf[offset:offset + flat.numel()].reshape(wq.shape)
    assert shifted.data_ptr() % 16 == offset % 16
    state = QuantState(absmax=am.cuda(), shape=w.shape, blocksize=64, quant_type="nf4", dtype=torch.float16)
    for m in (16, 48, 300):
        x = (torch.rand(m, k) - 0.5).to(torch.float16).cuda()
        want = matmul_4bit(x, flat.reshape(wq.shape), state)
        assert torch.equal(matmul_4bit(x, shifted, state), want), f"M={m}"


def test_matmul_4bit_decode_dequant_is_bit_exact():
    """Unit-vector activations isolate the in-register dequantisation: row i of the product is column j_i of the dequantised
    weight, bit for bit (nf4 and fp4, fp16 and bf16, fp32 and fp16 absmax)."""
    from conch_amd import _C

    k, n = 1024, 128
    cols = torch.tensor([0, 1, 7, 8, 31, 32, 63, 64, 127, 128, 500, 511, 512, 777, 1000, 1023])
    for qt in ("nf4", "fp4"):
        for dname, absmax_dt in (("f16", torch.float32), ("bf16", torch.float32), ("bf16", torch.bfloat16)):
            dt = DT[dname]
            seed_everything(6)
            w = torch.randn(n, k, dtype=dt)
            wq, am = oracle.quantize_blockwise_ref(w, 64, qt, None, absmax_dtype=absmax_dt)
            wd = oracle.dequantize_blockwise_ref(wq, am, n * k, 64, qt, dt).reshape(n, k)
            x = torch.zeros((len(cols), k), dtype=dt)
            x[torch.arange(len(cols)), cols] = 1.0
            state = QuantState(absmax=am.cuda(), shape=w.shape, blocksize=64, quant_type=qt, dtype=dt)
            try:
                _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
                got = matmul_4bit(x.cuda(), wq.cuda(), state).cpu()
            finally:
                _C.set_gemm_variant(_C.VARIANT_AUTO)
            want = wd[:, cols].T.contiguous()
            want = torch.where(want == 0, torch.zeros_like(want), want)  # fp4 has a -0 code; a sum of products returns +0
            np.testing.assert_array_equal(to_bits(got), to_bits(want), err_msg=f"{qt} {dname} {absmax_dt}")


@pytest.mark.parametrize("dname", list(DT))
@pytest.mark.parametrize("qt", QTYPES)
@pytest.mark.parametrize("shift", [1, 3])
def test_quantize_blockwise_takes_a_misaligned_slice(dname, qt, shift):
    """t[shift:] of a contiguous tensor is contiguous but not 16-byte aligned; the reference takes any pointer.
    The kernel falls back to per-element loads for such a base and must produce the same codes and absmax."""
    seed_everything(5)
    blocksize, n = 64, 64 * 37 + 10
    base = torch.randn((n + shift,), dtype=DT[dname])
    if qt == "fp4":
        base = base.uniform_(-1.0, 1.0)
    x = base[shift:]
    code = _create_dynamic_map() if qt == "fp8" else None
    want_q, want_absmax = oracle.quantize_blockwise_ref(x.clone(), blocksize, qt, code)
    xd = base.cuda()[shift:]
    assert xd.data_ptr() % 16 != 0 and xd.is_contiguous()
    got_q, state = quantize_blockwise(xd, blocksize=blocksize, quant_type=qt, code=None if code is None else code.cuda())
    np.testing.assert_array_equal(to_bits(state.absmax), to_bits(want_absmax.to(state.absmax.dtype)))
    np.testing.assert_array_equal(got_q.cpu().numpy().reshape(-1), want_q.numpy().reshape(-1))
