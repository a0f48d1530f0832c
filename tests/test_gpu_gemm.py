"""GPU parity: scaled_gemm and mixed_precision_gemm vs the CPU oracle, through the public ops.

Matrices follow the reference (tests/scaled_gemm_test.py:31-78, tests/mixed_precision_gemm_test.py:36-77)
with much tighter bars: int8 inputs are BIT-EXACT (int32 accumulation is exact and the epilogue
restates the oracle's order of operations); fp8 / mixed results differ from the oracle only by the
fp32 accumulation order, bounded below in units of the output dtype's epsilon.
"""

from __future__ import annotations

import functools
import math

import numpy as np
import pytest
import torch

import oracle
from conch_amd import _C
from conch_amd.ops.quantization.gemm import (
    mixed_precision_gemm,
    mixed_precision_gemm_gelu_tanh_and_mul,
    mixed_precision_gemm_silu_and_mul,
    scaled_gemm,
    scaled_gemm_gelu_tanh_and_mul,
    scaled_gemm_silu_and_mul,
)
from conch_amd.third_party.vllm.quant_utils import pack_rows, quantize_weights
from conch_amd.third_party.vllm.scalar_type import scalar_types
from conch_amd.third_party.vllm.utils import seed_everything
from tests.conftest import DT, from_bits, to_bits

pytestmark = pytest.mark.gpu

IN_T = {"int8": torch.int8, "fn": torch.float8_e4m3fn, "fnuz": torch.float8_e4m3fnuz}
SHAPES = [(128, 256, 128), (1024, 1024, 1024), (4096, 2048, 4096)]
VARIANTS = {"auto": _C.VARIANT_AUTO, "generic": _C.VARIANT_GENERIC, "simple": _C.VARIANT_MFMA_SIMPLE,
            "pingpong": _C.VARIANT_MFMA_PINGPONG, "skinny": _C.VARIANT_MFMA_SKINNY, "pingpong2": _C.VARIANT_MFMA_PINGPONG2,
            "mid": _C.VARIANT_MFMA_MID, "asm1w": _C.VARIANT_MFMA_ASM1W}
# fp accumulation-order tolerance, relative to max|C| of the case, in output-dtype epsilons
EPS = {torch.float16: 2.0**-10, torch.bfloat16: 2.0**-7}


@pytest.fixture(autouse=True)
def _reset_variant():
    yield
    _C.set_gemm_variant(_C.VARIANT_AUTO)


def make_scaled_inputs(m, k, n, in_dtype, out_dtype, sa_scalar, sb_scalar, use_bias):
    """Recipe of tests/scaled_gemm_test.py:54-73, generated on the CPU (the oracle's side)."""
    seed_everything(0)
    sa = torch.rand((1, 1), dtype=torch.float32) if sa_scalar else 0.25 * torch.rand((m, 1))
    sb = torch.rand((1, 1), dtype=torch.float32) if sb_scalar else 0.25 * torch.rand((n, 1))
    if in_dtype.is_floating_point:
        a = (0.25 * torch.rand((m, k), dtype=torch.float32)).to(in_dtype)
        b = (0.25 * torch.rand((n, k), dtype=torch.float32)).to(in_dtype).T
    else:
        a = torch.randint(-32, 32, (m, k), dtype=in_dtype)
        b = torch.randint(-32, 32, (n, k), dtype=in_dtype).T
    bias = torch.rand((n,), dtype=out_dtype) if use_bias else None
    return a, b, sa, sb, bias


def check_scaled(got, ref, in_dtype, out_dtype, inputs=None):
    """int8: bit-exact.  fp8: a global bound and -- when `inputs` = (a, b, sa, sb, bias) is given -- a PER-ELEMENT bound, so
    that an error confined to rows / columns with tiny scales cannot hide under max|C|:
        |got - ref| <= 2 eps_out (|ref| + |bias|)  +  |sa_m sb_n| * K 2^-24 * sum_k |a_mk| |b_kn|
    (two roundings to the output dtype at most one ulp apart each, plus the worst-case fp32 accumulation-order error)."""
    if in_dtype == torch.int8:
        np.testing.assert_array_equal(to_bits(got), to_bits(ref))
        return
    g, r = got.float().cpu(), ref.float()
    tol = 2.0 * EPS[out_dtype] * max(r.abs().max().item(), 1e-6)
    err = (g - r).abs().max().item()
    assert err <= tol, f"max |diff| {err:.4g} > {tol:.4g}"
    if inputs is not None:
        a, b, sa, sb, bias = inputs
        k = a.shape[1]
        s_abs = a.float().abs() @ b.float().abs()
        scale = (sa.reshape(-1, 1) if sa.numel() > 1 else sa.reshape(1, 1)).abs() * (sb.reshape(1, -1) if sb.numel() > 1 else sb.reshape(1, 1)).abs()
        # (+ two quanta of the output format's subnormal grid: below 2^-14 an fp16 rounding moves by 2^-24 whatever |ref| is --
        # rows and columns with tiny scales put outputs there)
        tiny = 2.0**-24 if out_dtype == torch.float16 else 2.0**-133
        bound = 2.0 * EPS[out_dtype] * (r.abs() + (bias.float().abs().reshape(1, -1) if bias is not None else 0.0)) \
            + scale * (k * 2.0**-24) * s_abs + 2.0 * tiny
        excess = ((g - r).abs() - bound).max().item()
        assert excess <= 0, f"per-element bound exceeded by {excess:.4g}"


def run_scaled(a, b, sa, sb, out_dtype, bias):
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    # b is a transposed view: keep the (1, K) strides on the device like the reference test does
    b_dev = b.T.contiguous().cuda().T if b.stride(0) == 1 else b.cuda()
    return scaled_gemm(dev(a), b_dev, dev(sa), dev(sb), out_dtype, dev(bias))


@pytest.mark.parametrize(("m", "k", "n"), SHAPES[:2])
@pytest.mark.parametrize("iname", list(IN_T))
@pytest.mark.parametrize("oname", ["f16", "bf16"])
@pytest.mark.parametrize("sa_scalar", [True, False])
@pytest.mark.parametrize("sb_scalar", [True, False])
@pytest.mark.parametrize("use_bias", [True, False])
def test_scaled_gemm_matrix(m, k, n, iname, oname, sa_scalar, sb_scalar, use_bias):
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], DT[oname], sa_scalar, sb_scalar, use_bias)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, DT[oname], bias)
    got = run_scaled(a, b, sa, sb, DT[oname], bias)
    assert got.shape == (m, n) and got.dtype == DT[oname]
    check_scaled(got, ref, IN_T[iname], DT[oname], (a, b, sa, sb, bias))


@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize("oname", ["f16", "bf16"])
@pytest.mark.parametrize("sa_scalar", [True, False])
@pytest.mark.parametrize("sb_scalar", [True, False])
@pytest.mark.parametrize("use_bias", [True, False])
def test_scaled_gemm_large_shape(iname, oname, sa_scalar, sb_scalar, use_bias):
    """The reference's third shape (4096, 2048, 4096), all 16 scale / bias / dtype combinations (tests/scaled_gemm_test.py:31-50)."""
    m, k, n = SHAPES[2]
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], DT[oname], sa_scalar, sb_scalar, use_bias)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, DT[oname], bias)
    check_scaled(run_scaled(a, b, sa, sb, DT[oname], bias), ref, IN_T[iname], DT[oname], (a, b, sa, sb, bias))


@pytest.mark.parametrize(("oname", "sa_scalar", "sb_scalar", "use_bias"),
                         [("bf16", False, False, True), ("f16", True, False, False), ("bf16", False, True, False)])
def test_scaled_gemm_large_shape_fnuz(oname, sa_scalar, sb_scalar, use_bias):
    m, k, n = SHAPES[2]
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T["fnuz"], DT[oname], sa_scalar, sb_scalar, use_bias)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, DT[oname], bias)
    check_scaled(run_scaled(a, b, sa, sb, DT[oname], bias), ref, IN_T["fnuz"], DT[oname], (a, b, sa, sb, bias))


@pytest.mark.parametrize("oname", ["bf16", "f16"])
def test_scaled_gemm_fp32_bias_is_added_like_the_reference(oname):
    """A bias that is not in the output dtype: the reference's `output.add_(bias)` (ops/quantization/gemm.py:249-250) adds in
    fp32 and rounds once; casting the bias first would round twice."""
    m, k, n = 256, 512, 384
    a, b, sa, sb, _ = make_scaled_inputs(m, k, n, torch.int8, DT[oname], False, False, False)
    bias = torch.rand((n,), dtype=torch.float32)
    nobias = oracle.scaled_gemm_ref(a, b, sa, sb, DT[oname], None)
    want = (nobias.float() + bias).to(DT[oname])
    got = run_scaled(a, b, sa, sb, DT[oname], bias).cpu()
    assert got.dtype == DT[oname]
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_scaled_gemm_small_scale_rows_and_columns():
    """Rows / columns whose scales are 1e-6 of the rest: a kernel that is wrong only there passes any max|C| bound."""
    m, k, n = 512, 1024, 768
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T["fn"], torch.bfloat16, False, False, False)
    sa[::7] *= 1e-6
    sb[::5] *= 1e-6
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, None)
    check_scaled(run_scaled(a, b, sa, sb, torch.bfloat16, None), ref, IN_T["fn"], torch.bfloat16, (a, b, sa, sb, None))


# ---------------------------------------------------------------------------------------------
# tile schedules of the 256-row kernel and launch forms of the split-K kernel: the schedule / launch form moves WORK,
# never the order in which one output element accumulates its K products -> results are bit-identical across them
# ---------------------------------------------------------------------------------------------
@pytest.fixture
def _reset_tuning():
    yield
    _C.set_tuning(_C.TUNE_TILE_SCHEDULE, 0)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, 0)
    _C.set_tuning(_C.TUNE_PERSISTENT, 0)
    _C.set_tuning(_C.TUNE_EPILOGUE, 0)
    _C.set_tuning(_C.TUNE_MID_STAGES, 0)
    _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)
    _C.set_tuning(_C.TUNE_SKINNY_GATHER, 0)
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, 0)
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
    _C.set_tuning(_C.TUNE_MID_SPLITK, 0)
    _C.set_tuning(_C.TUNE_SKINNY_CHUNKS, 0)
    _C.set_tuning(_C.TUNE_MIXED_STRIP_ROWS, 0)


@pytest.mark.parametrize("slices", [2, 3, 5, 8])
@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n", "with_bias"), [(256, 4096, 1024, True), (300, 2176, 520, False), (129, 1024, 8, True), (512, 11008, 256, False),
                                                         (1000, 1152, 132, True)])
def test_mid_kernel_split_k(_reset_tuning, slices, iname, m, k, n, with_bias):
    """The 128x128-tile kernel with K split over `slices` workgroups per tile (raw partial tiles to slabs, the split-K reduce kernel
    sums them in slice order and applies the epilogue): int8 bit-identical to the whole-K kernel (integer sums), fp8 within the
    oracle's per-element bound (fp32 partial sums meet in a different order); both ring depths; ragged tiles, slices of unequal
    length (K of 17 steps in 3, 5 or 8 slices)."""
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, with_bias)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_gemm_variant(_C.VARIANT_MFMA_MID)
    try:
        _C.set_tuning(_C.TUNE_MID_SPLITK, 1)
        whole = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
        _C.set_tuning(_C.TUNE_MID_SPLITK, slices)
        outs = []
        for stages in (2, 4):
            _C.set_tuning(_C.TUNE_MID_STAGES, stages)
            outs.append(run_scaled(a, b, sa, sb, torch.bfloat16, bias))
        assert torch.equal(outs[0], outs[1])  # the ring depth moves loads, not sums
        check_scaled(outs[0], ref, IN_T[iname], torch.bfloat16, (a, b, sa, sb, bias))
        if iname == "int8":
            assert torch.equal(outs[0], whole)
        for _ in range(2):
            assert torch.equal(run_scaled(a, b, sa, sb, torch.bfloat16, bias), outs[1])
    finally:
        _C.set_gemm_variant(0)


@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n"), [(4096, 512, 11008), (512, 256, 2000), (300, 384, 1376), (2304, 512, 4672), (256, 256, 3584),
                                            (1024, 384, 968), (8192, 256, 3584)])
def test_two_width_tile_schedule_is_bit_identical_and_correct(_reset_tuning, iname, m, k, n):
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, True)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    _C.set_tuning(_C.TUNE_TILE_SCHEDULE, 1)
    uniform = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
    for mode in (3, 4, 2):  # odd XCDs narrow-first; wide / narrow on alternate slots of every XCD; wide first everywhere
        _C.set_tuning(_C.TUNE_TILE_SCHEDULE, mode)
        balanced = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
        np.testing.assert_array_equal(to_bits(balanced), to_bits(uniform))
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
    check_scaled(balanced, ref, IN_T[iname], torch.bfloat16, (a, b, sa, sb, bias))


@pytest.mark.parametrize("oname", ["f16", "bf16"])
@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n", "use_bias"), [(4096, 512, 11008, False), (1000, 640, 1500, True), (300, 384, 520, True),
                                                        (257, 256, 8, False), (2304, 512, 4672, True), (512, 256, 1004, False)])
def test_row_major_epilogue_is_bit_identical(_reset_tuning, oname, iname, m, k, n, use_bias):
    """Epilogue through LDS (whole 128-byte rows per store) against the direct accumulator-layout stores: the same values,
    another route to memory -- full tiles, ragged M / N edges, rows that are not 16-byte aligned (N = 1004, 8)."""
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], DT[oname], False, False, use_bias)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    _C.set_tuning(_C.TUNE_EPILOGUE, 1)
    direct = run_scaled(a, b, sa, sb, DT[oname], bias)
    _C.set_tuning(_C.TUNE_EPILOGUE, 2)
    rows = run_scaled(a, b, sa, sb, DT[oname], bias)
    np.testing.assert_array_equal(to_bits(rows), to_bits(direct))
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, DT[oname], bias)
    check_scaled(rows, ref, IN_T[iname], DT[oname], (a, b, sa, sb, bias))


def test_row_major_epilogue_with_c_between_2_and_4_gib(_reset_tuning):
    """The whole-line epilogue addresses C through one buffer descriptor with unsigned 32-bit byte offsets: a 2.7 GB output (rows
    beyond the 2 GiB mark) must equal the direct stores'; C beyond 4 GiB falls back to them by itself."""
    m, k, n = 65536, 256, 20480
    torch.manual_seed(3)
    a = (0.25 * torch.rand((m, k), device="cuda")).to(torch.float8_e4m3fn)
    bt = (0.25 * torch.rand((n, k), device="cuda")).to(torch.float8_e4m3fn)
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    _C.set_tuning(_C.TUNE_EPILOGUE, 1)
    direct = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    _C.set_tuning(_C.TUNE_EPILOGUE, 2)
    rows = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    _C.set_gemm_variant(0)
    assert direct.numel() * 2 > 2**31
    assert torch.equal(rows.view(torch.int16), direct.view(torch.int16))
    # a band of rows beyond the 2 GiB mark against the oracle
    r0 = m - 300
    ref = oracle.scaled_gemm_ref(a[r0:].cpu(), bt.T.cpu(), sa[r0:].cpu(), sb.cpu(), torch.bfloat16, None)
    check_scaled(rows[r0:].cpu(), ref, IN_T["fn"], torch.bfloat16)


def test_row_major_epilogues_repeat_bit_identically():
    """The whole-line epilogues store with 16-byte buffer stores.  With a REGISTER soffset hipcc (ROCm 7.2) does not guard such a
    store's data registers against its next VALU write, which showed on gfx950 as a wrong first dword in a few lanes of SOME
    launches (round 3) -- so: many launches of each form, every one equal to the direct-store result."""
    a, b, sa, sb, _ = make_scaled_inputs(1024, 512, 1536, IN_T["fn"], torch.bfloat16, False, False, False)
    wt = WTYPES["uint4b8"]
    x, _, packed, w_s, _ = make_mixed_inputs(1024, 256, 1536, wt, False, torch.float16)
    xd, pd, wsd = x.cuda(), packed.cuda(), w_s.cuda()
    try:
        _C.set_tuning(_C.TUNE_EPILOGUE, 1)
        want_s = {v: None for v in (_C.VARIANT_MFMA_PINGPONG2, _C.VARIANT_MFMA_MID)}
        for v in want_s:
            _C.set_gemm_variant(v)
            want_s[v] = run_scaled(a, b, sa, sb, torch.bfloat16, None)
        _C.set_gemm_variant(0)
        want_m = {}
        for nt in (2, 3, 4, 5):
            _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
            want_m[nt] = mixed_precision_gemm(xd, pd, wsd, None, wt.size_bits, wt.bias, 128)
        _C.set_tuning(_C.TUNE_EPILOGUE, 2)
        for _ in range(25):
            for v, want in want_s.items():
                _C.set_gemm_variant(v)
                assert torch.equal(run_scaled(a, b, sa, sb, torch.bfloat16, None), want)
            _C.set_gemm_variant(0)
            for nt, want in want_m.items():
                _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
                assert torch.equal(mixed_precision_gemm(xd, pd, wsd, None, wt.size_bits, wt.bias, 128), want)
    finally:
        _C.set_gemm_variant(0)
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
        _C.set_tuning(_C.TUNE_EPILOGUE, 0)


@pytest.mark.parametrize("workgroups", [2, 7, 16, 256])
@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n"), [(4096, 512, 11008), (1024, 512, 2048), (1000, 640, 1500), (2304, 1024, 4672), (512, 4096, 768),
                                            (8192, 512, 3584)])
def test_persistent_tile_walk_is_bit_identical_and_correct(_reset_tuning, workgroups, iname, m, k, n):
    """Persistent workgroups (next tile's first two K steps and scales prefetched under the epilogue, counted waits across the
    tile boundary) against one workgroup per tile: the same per-element accumulation order, so the same bits -- for full tiles
    (counted wait behind exactly 16 stores) and for ragged ones (drained), with 1 to hundreds of tiles per workgroup."""
    a, b, sa, sb, _ = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, False)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    _C.set_tuning(_C.TUNE_PERSISTENT, 1)
    plain = run_scaled(a, b, sa, sb, torch.bfloat16, None)
    _C.set_tuning(_C.TUNE_PERSISTENT, workgroups)
    for _ in range(2):
        walk = run_scaled(a, b, sa, sb, torch.bfloat16, None)
        np.testing.assert_array_equal(to_bits(walk), to_bits(plain))
    # CONCH_TUNE_EPILOGUE = 1 is honoured by the persistent walk too: direct stores, every boundary drained instead of counted
    _C.set_tuning(_C.TUNE_EPILOGUE, 1)
    np.testing.assert_array_equal(to_bits(run_scaled(a, b, sa, sb, torch.bfloat16, None)), to_bits(plain))
    _C.set_tuning(_C.TUNE_EPILOGUE, 0)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, None)
    check_scaled(walk, ref, IN_T[iname], torch.bfloat16, (a, b, sa, sb, None))


@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n"), [(512, 4096, 1024), (300, 128, 520), (257, 256, 8), (128, 384, 136), (1024, 512, 4096), (200, 1152, 64)])
def test_mid_kernel_ring_depths_are_bit_identical(_reset_tuning, iname, m, k, n):
    """The 128x128-tile kernel with the 2-stage loop (two workgroups per CU) and the 4-stage ring (one per CU, three K steps in
    flight; K of one to three steps included): the ring depth moves loads, not the accumulation order."""
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, True)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_gemm_variant(_C.VARIANT_MFMA_MID)
    _C.set_tuning(_C.TUNE_MID_STAGES, 2)
    two = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_tuning(_C.TUNE_MID_STAGES, 4)
    four = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
    assert torch.equal(two, four)
    check_scaled(four, ref, IN_T[iname], torch.bfloat16, (a, b, sa, sb, bias))
    for _ in range(3):  # back to back: the ring of one launch must not depend on what the previous one left in LDS
        assert torch.equal(run_scaled(a, b, sa, sb, torch.bfloat16, bias), four)


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n"), [(128, 4096, 4096), (1, 1024, 24), (20, 2048, 4096), (33, 1024, 64), (48, 2048, 520),
                                            (200, 2048, 520), (256, 8192, 1376), (64, 4096, 11008), (100, 3072, 260)])
def test_one_launch_splitk_is_bit_identical_to_two_launches(_reset_tuning, mode, iname, m, k, n):
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, True)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, 1)
    two = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, mode)
    for _ in range(3):  # the arrival counters must come back to zero after every launch
        one = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
        if iname == "int8" or mode == 2:
            np.testing.assert_array_equal(to_bits(one), to_bits(two))  # same slices, same order
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
    check_scaled(one, ref, IN_T[iname], torch.bfloat16, (a, b, sa, sb, bias))


@pytest.mark.parametrize("chunks", [2, 3, 7])
@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n"), [(64, 7168, 520), (20, 11008, 256), (128, 8192, 132), (200, 5120, 64), (7, 14336, 24), (33, 3072, 68)])
def test_splitk_passes_per_slice(_reset_tuning, chunks, mode, iname, m, k, n):
    """CONCH_TUNE_SKINNY_CHUNKS: a split-K workgroup makes several passes over its K slice (accumulators kept in registers from
    pass to pass) and leaves ONE slab: int8 bit-identical to a slab per pass, fp8 within the oracle's bound (the fp32 partial
    sums meet in a different order); ragged last slices that end after any number of passes, all launch forms, K below one
    slice of `chunks` passes included."""
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, True)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, mode)
    _C.set_tuning(_C.TUNE_SKINNY_CHUNKS, 1)
    one = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_tuning(_C.TUNE_SKINNY_CHUNKS, chunks)
    for _ in range(3):
        got = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
        if iname == "int8":
            np.testing.assert_array_equal(to_bits(got), to_bits(one))
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
    check_scaled(got, ref, IN_T[iname], torch.bfloat16, (a, b, sa, sb, bias))


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n"), [(128, 4096, 4096), (1, 1024, 24), (20, 2048, 4096), (33, 1152, 68), (48, 2048, 520), (200, 2048, 520),
                                            (64, 4096, 11008), (100, 3072, 260), (7, 11008, 4096)])
def test_splitk_fetch_orders_are_bit_identical(_reset_tuning, mode, iname, m, k, n):
    """CONCH_TUNE_SKINNY_GATHER: a wave's B^T fragments fetched in MFMA operand order, or four lanes per 64-byte row piece and
    gathered across lanes -- the same bytes in the same operand registers, so the launch forms agree bit for bit (ragged N and
    M, a ragged last K slice and a row count past the last 16-row block included)."""
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, True)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, mode)
    _C.set_tuning(_C.TUNE_SKINNY_GATHER, 1)
    plain = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_tuning(_C.TUNE_SKINNY_GATHER, 2)
    for _ in range(2):
        np.testing.assert_array_equal(to_bits(run_scaled(a, b, sa, sb, torch.bfloat16, bias)), to_bits(plain))
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
    check_scaled(plain, ref, IN_T[iname], torch.bfloat16, (a, b, sa, sb, bias))


@pytest.mark.parametrize(("m", "k", "n", "mode", "iname"), [(128, 4096, 4096, 2, "int8"), (16, 4096, 11008, 2, "fn"), (32, 8192, 8192, 3, "int8"),
                                                            (8, 4096, 28672, 2, "fn")])
def test_one_launch_splitk_under_uneven_load(_reset_tuning, m, k, n, mode, iname):
    """The hand-off must not depend on timing, placement or on one workgroup per CU (the last three shapes run 688 - 1792
    workgroups of 32 KiB of LDS: up to four per CU): run it while another stream keeps the chip unevenly busy, many times,
    against the two-launch result."""
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, False)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, 1)
    want = to_bits(run_scaled(a, b, sa, sb, torch.bfloat16, bias))
    _C.set_tuning(_C.TUNE_SKINNY_MODE, mode)
    ad, bd, sad, sbd = a.cuda(), b.T.contiguous().cuda().T, sa.cuda(), sb.cuda()
    noise_a = torch.randn(3000, 3000, device="cuda")
    side = torch.cuda.Stream()
    for i in range(40):
        with torch.cuda.stream(side):
            for _ in range(1 + i % 3):
                noise_a @ noise_a
        got = scaled_gemm(ad, bd, sad, sbd, torch.bfloat16)
        np.testing.assert_array_equal(to_bits(got), want)
    torch.cuda.synchronize()


@pytest.mark.parametrize("variant", ["generic", "simple", "pingpong", "pingpong2", "mid", "skinny"])
@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n"), [(128, 256, 128), (1024, 1024, 1024), (300, 384, 520), (257, 128, 8),
                                            (512, 1152, 1376), (2304, 512, 4672), (4096, 256, 11008),
                                            (128, 1024, 4096), (200, 2048, 520), (1, 1024, 24), (256, 3072, 1376),
                                            (48, 1024, 520), (20, 2048, 4096), (33, 1024, 64),
                                            (16, 1152, 256), (96, 2944, 520), (200, 128, 64)])
def test_scaled_gemm_every_kernel_variant(variant, iname, m, k, n):
    """Each device kernel, including ragged M/N tails (partial tiles, N not a multiple of 256 / 16)."""
    if variant == "pingpong2" and k < 256:
        pytest.skip("two-phase ping-pong stages two K steps in its prologue (K >= 256)")
    if variant == "skinny" and (m > 256 or (k % 1024 and n % 4)):
        pytest.skip("skinny kernel: M <= 256, and N % 4 == 0 (split-K form) or K a multiple of 1024 (in-workgroup form)")
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, True)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_gemm_variant(VARIANTS[variant])
    check_scaled(run_scaled(a, b, sa, sb, torch.bfloat16, bias), ref, IN_T[iname], torch.bfloat16)


@pytest.mark.parametrize("iname", ["fn", "int8"])
@pytest.mark.parametrize("oname", ["bf16", "f16"])
@pytest.mark.parametrize(("vec_a", "vec_b"), [(True, True), (False, False), (True, False)])
@pytest.mark.parametrize(("m", "k", "n"), [(256, 512, 352), (256, 512, 176), (256, 1024, 704), (512, 768, 352), (128, 512, 352), (300, 768, 400),
                                             (1000, 2048, 1008), (257, 512, 368), (2304, 1024, 1056), (16, 512, 16)])
def test_asm1w_kernel_is_bit_identical_and_correct(_reset_tuning, m, k, n, vec_a, vec_b, oname, iname):
    """The one-wave-per-SIMD assembly kernel (round 5; csrc/asm/gen_gemm1w.py: 256 x 352 tiles, 352 accumulators per wave, two
    workgroup barriers per K step) against the 256 x 256-tile HIP kernel: the same MFMA on the same 128-byte K steps in the same
    order and the same epilogue arithmetic -- the same bits -- on single tiles, ragged M and N (the direct-store epilogue of edge
    waves beside the row-major one of whole waves), one raster group of every height 1..8 (+ a partial ninth), scalar and vector
    scales, both output dtypes; and against the oracle per element."""
    a, b, sa, sb, _ = make_scaled_inputs(m, k, n, IN_T[iname], DT[oname], not vec_a, not vec_b, False)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    want = run_scaled(a, b, sa, sb, DT[oname], None)
    _C.set_gemm_variant(_C.VARIANT_MFMA_ASM1W)
    for _ in range(2):
        got = run_scaled(a, b, sa, sb, DT[oname], None)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), f"{(got != want).sum().item()} elements differ"
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, DT[oname], None)
    check_scaled(got, ref, IN_T[iname], DT[oname], (a, b, sa, sb, None))


@pytest.mark.parametrize("iname", ["fn", "int8"])
@pytest.mark.parametrize("nt", [9, 7])
@pytest.mark.parametrize(("m", "k", "n"), [(256, 512, 224), (256, 512, 288), (300, 768, 400), (1000, 2048, 1008), (2304, 1024, 3584)])
def test_asm1w_kernel_other_tile_widths(_reset_tuning, m, k, n, nt, iname):
    """The 288- and 224-column instantiations of the assembly kernel (9 / 7 n tiles per wave; the dispatcher takes them where they
    fill the chip in fuller rounds: the C5 shard of 3584 columns = 16 x 224), forced through the diagnostic width hook: bit-identical
    to the 256 x 256-tile HIP kernel."""
    import ctypes

    hook = _C.load().conch_debug_gemm1w_width
    hook.restype, hook.argtypes = ctypes.c_int, [ctypes.c_int]
    a, b, sa, sb, _ = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, False)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    want = run_scaled(a, b, sa, sb, torch.bfloat16, None)
    _C.set_gemm_variant(_C.VARIANT_MFMA_ASM1W)
    hook(nt)
    try:
        got = run_scaled(a, b, sa, sb, torch.bfloat16, None)
    finally:
        hook(0)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), f"{(got != want).sum().item()} elements differ"


def test_mixed_asm1w_kernel_is_bit_identical(_reset_tuning):
    """The one-wave-per-SIMD int4 x fp16 assembly kernel (csrc/asm/gen_mixed1w.py; opt-in, CONCH_TUNE_MIXED_KERNEL = 3: it is NOT
    faster than the column-strip HIP kernel -- a lone wave cannot issue its dequantisation beside its own MFMAs,
    profiles/r05/mixed1w_first_ab.txt) against that kernel: the same dequantised weights, MFMA and K order -- the same bits, at both
    tile widths, ragged M / N included; and its contract is enforced."""
    wt = scalar_types.uint4b8
    for m, k, n in [(256, 256, 192), (300, 384, 400), (1000, 2048, 1008), (1024, 4096, 11008)]:
        a, w_ref, packed, w_s, _ = make_mixed_inputs(m, k, n, wt, False, torch.float16)
        args = (a.cuda(), packed.cuda(), w_s.cuda(), None, wt.size_bits, wt.bias, 128)
        _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, 2)
        want = mixed_precision_gemm(*args)
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, 3)
        for nt in (3, 4):
            _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
            got = mixed_precision_gemm(*args)
            assert torch.equal(got, want), f"{m}x{k}x{n} width {64 * nt}: {(got != want).sum().item()} elements differ"
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
        if m <= 1000:
            check_mixed(got, a, w_ref, k)
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(256, 256, 192, scalar_types.uint4, True, torch.float16)
    with pytest.raises(NotImplementedError):  # per-group zero points: outside the contract
        mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.cuda(), w_zp.cuda(), 4, 0, 128)


@pytest.mark.parametrize("iname", ["fn", "int8"])
@pytest.mark.parametrize("oname", ["bf16", "f16"])
@pytest.mark.parametrize(("m", "k", "n", "nt"), [(256, 512, 352, 0), (300, 768, 400, 0), (1000, 2048, 1008, 0), (2304, 1024, 1056, 0), (16, 512, 16, 0),
                                                  (512, 512, 288, 9), (300, 768, 400, 7), (2304, 1024, 3584, 7), (4096, 512, 11008, 0)])
def test_asm1w_kernel_adds_the_bias_in_its_epilogue(_reset_tuning, m, k, n, nt, oname, iname):
    """Round 6: `out = cast(sb * (sa * acc)) + bias` (conch/reference/quantization/scaled_gemm.py:24-25; the reference adds the bias
    for any shape, conch/ops/quantization/gemm.py:249-250) inside the assembly kernel -- the *_bias twins of all twelve kernels, both
    epilogues (whole waves through the LDS row image, edge waves with direct stores), the persistent walk's second tile (the bias
    of the NEXT tile is fetched and parked beside its scales), all three tile widths: bit-identical to the HIP kernel's epilogue
    (unpack, fp32 add, RNE cast) and, for int8, to the oracle."""
    import ctypes

    hook = _C.load().conch_debug_gemm1w_width
    hook.restype, hook.argtypes = ctypes.c_int, [ctypes.c_int]
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], DT[oname], False, False, True)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    want = run_scaled(a, b, sa, sb, DT[oname], bias)
    _C.set_gemm_variant(_C.VARIANT_MFMA_ASM1W)
    hook(nt)
    try:
        for _ in range(2):
            got = run_scaled(a, b, sa, sb, DT[oname], bias)
            assert torch.equal(got.view(torch.int16), want.view(torch.int16)), f"{(got != want).sum().item()} elements differ"
    finally:
        hook(0)
        _C.set_gemm_variant(_C.VARIANT_AUTO)
    if m * n <= 4_000_000:
        ref = oracle.scaled_gemm_ref(a, b, sa, sb, DT[oname], bias)
        check_scaled(got, ref, IN_T[iname], DT[oname], (a, b, sa, sb, bias))


def test_asm1w_kernel_contract_is_enforced(_reset_tuning):
    """Forced onto a problem outside its contract (e4m3fnuz; K not a multiple of 256) the variant refuses with the library's
    UNSUPPORTED status -- it never runs something else silently."""
    _C.set_gemm_variant(_C.VARIANT_MFMA_ASM1W)
    a8, b8, sa8, sb8, _ = make_scaled_inputs(256, 512, 352, torch.float8_e4m3fnuz, torch.bfloat16, False, False, False)
    with pytest.raises(NotImplementedError):
        run_scaled(a8, b8, sa8, sb8, torch.bfloat16, None)
    a3, b3, sa3, sb3, _ = make_scaled_inputs(256, 384, 352, torch.float8_e4m3fn, torch.bfloat16, False, False, False)
    with pytest.raises(NotImplementedError):
        run_scaled(a3, b3, sa3, sb3, torch.bfloat16, None)


@pytest.mark.parametrize("iname", list(IN_T))
@pytest.mark.parametrize("key_tail", [f"{o}_sa{sa}_sb{sb}_b{b}" for o in ("f16", "bf16") for sa in (0, 1) for sb in (0, 1) for b in (0, 1)])
def test_scaled_gemm_golden_from_reference(golden, iname, key_tail):
    """All 48 fixtures the reference's own oracle produced (16 scale / bias / dtype combinations x int8, e4m3fn, e4m3fnuz:
    conch/tests/scaled_gemm_test.py's matrix at its first shape) through the HIP path."""
    g = golden("scaled_gemm")
    key = f"{iname}_{key_tail}"
    oname = key_tail.split("_")[0]
    a = from_bits(g[f"a_{key}"], IN_T[iname])
    b = from_bits(g[f"bt_{key}"], IN_T[iname]).T
    sa, sb = torch.from_numpy(g[f"sa_{key}"]), torch.from_numpy(g[f"sb_{key}"])
    bias = from_bits(g[f"bias_{key}"], DT[oname]) if key_tail.endswith("b1") else None
    got = run_scaled(a, b, sa, sb, DT[oname], bias)
    check_scaled(got, from_bits(g[f"c_{key}"], DT[oname]).reshape(128, 128), IN_T[iname], DT[oname], (a, b, sa, sb, bias))


def test_scaled_gemm_layouts_and_edge_cases():
    """Row-major (N-contiguous) B, 0-dim scales, K not a multiple of 128, empty M: all legal inputs."""
    seed_everything(1)
    m, k, n = 96, 200, 72
    a = torch.randint(-32, 32, (m, k), dtype=torch.int8)
    b = torch.randint(-32, 32, (k, n), dtype=torch.int8)  # row-major: strides (n, 1)
    sa, sb = torch.tensor(0.37), 0.25 * torch.rand((n, 1))
    ref = oracle.scaled_gemm_ref(a, b, sa.reshape(1, 1), sb, torch.float16, None)
    got = scaled_gemm(a.cuda(), b.cuda(), sa.cuda(), sb.cuda(), torch.float16)
    np.testing.assert_array_equal(to_bits(got), to_bits(ref))
    empty = scaled_gemm(a[:0].cuda(), b.cuda(), sa.cuda(), sb.cuda(), torch.float16)
    assert empty.shape == (0, n)
    with pytest.raises(ValueError):
        scaled_gemm(a.cuda(), b.cuda().to(torch.uint8).view(torch.float8_e4m3fn), sa.cuda(), sb.cuda(), torch.float16)
    with pytest.raises(NotImplementedError):
        scaled_gemm(a.cuda(), b.cuda(), sa.cuda(), sb.cuda(), torch.float32)


@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n"), [(300, 200, 520), (96, 1000, 72), (1024, 384, 256), (130, 4096, 64)])
@pytest.mark.parametrize("layout", ["b_row_major", "b_row_major_slice", "a_transposed", "misaligned_views", "odd_k_only"])
def test_scaled_gemm_non_native_layouts_are_repacked(iname, m, k, n, layout):
    """Legal inputs outside the MFMA layout contract (row-major [K][N] weights, a transposed activation
    view, slices that start at odd byte offsets, K not a multiple of 128) are repacked on the device and
    still run on the MFMA kernels; results must match the oracle exactly as for native layouts."""
    seed_everything(2)
    dt = IN_T[iname]
    if dt == torch.int8:
        a = torch.randint(-32, 32, (m, k), dtype=dt)
        b = torch.randint(-32, 32, (k, n), dtype=dt)
    else:
        a = (0.25 * torch.rand((m, k))).to(dt)
        b = (0.25 * torch.rand((k, n))).to(dt)
    sa, sb = 0.25 * torch.rand((m, 1)), 0.25 * torch.rand((n, 1))
    bias = torch.rand((n,), dtype=torch.bfloat16)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
    a_d, b_d = a.cuda(), b.cuda()  # both row-major: b has strides (n, 1)
    if layout == "b_row_major_slice":  # the first n columns of a wider row-major tensor: 16-byte aligned rows, ragged 16-column groups
        wide_b = torch.zeros((k, (n + 15) // 16 * 16 + 32), dtype=torch.uint8).view(dt).cuda()
        wide_b[:, :n] = b.cuda()
        b_d = wide_b[:, :n]
    elif layout == "a_transposed":
        a_d = a.T.contiguous().cuda().T  # strides (1, m)
        b_d = b.T.contiguous().cuda().T  # native for b
    elif layout == "misaligned_views":
        big_a = torch.zeros((m + 3, k + 37), dtype=torch.uint8).view(dt).cuda()
        big_a[1:m + 1, 5:k + 5] = a.cuda()
        a_d = big_a[1:m + 1, 5:k + 5]
        big_b = torch.zeros((n + 2, k + 11), dtype=torch.uint8).view(dt).cuda()
        big_b[1:n + 1, 3:k + 3] = b.T.cuda()
        b_d = big_b[1:n + 1, 3:k + 3].T
    elif layout == "odd_k_only":
        b_d = b.T.contiguous().cuda().T
    got = scaled_gemm(a_d, b_d, sa.cuda(), sb.cuda(), torch.bfloat16, bias.cuda())
    check_scaled(got, ref, dt, torch.bfloat16)


def test_scaled_gemm_int8_saturated_operands_exact():
    """Extreme int8 values: |acc| up to 128*128*K exceeds 2^24, the int32 accumulator stays exact."""
    m, k, n = 256, 2048, 256
    a = torch.full((m, k), -128, dtype=torch.int8)
    b = torch.full((n, k), -128, dtype=torch.int8).T
    a[::3] = 127
    one = torch.ones((1, 1))
    got = scaled_gemm(a.cuda(), b.T.contiguous().cuda().T, one.cuda(), one.cuda(), torch.bfloat16)
    exact = (a.double() @ b.double()).float().to(torch.bfloat16)  # float(int32) then RNE to bf16
    np.testing.assert_array_equal(to_bits(got), to_bits(exact))


def test_scaled_gemm_c2_config_bit_exact():
    """BASELINE config C2: int8 -> bf16, per-channel scales, M=128 K=4096 N=4096."""
    a, b, sa, sb, _ = make_scaled_inputs(128, 4096, 4096, torch.int8, torch.bfloat16, False, False, False)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, None)
    np.testing.assert_array_equal(to_bits(run_scaled(a, b, sa, sb, torch.bfloat16, None)), to_bits(ref))


def test_scaled_gemm_c3_config_properties():
    """BASELINE config C3 (fp8 -> bf16, 4096x4096x11008) at full size.

    (1) every element is checked against the oracle, per element; (2) the MFMA kernel agrees with the generic
    device kernel everywhere up to accumulation order; (3) row-linearity: permuting the rows of A
    permutes the rows of C bit-for-bit (each output row depends only on its own row of A).
    """
    m, k, n = 4096, 4096, 11008
    a, b, sa, sb, _ = make_scaled_inputs(m, k, n, torch.float8_e4m3fn, torch.bfloat16, False, False, False)
    a_d, bt_d, sa_d, sb_d = a.cuda(), b.T.contiguous().cuda(), sa.cuda(), sb.cuda()
    got = scaled_gemm(a_d, bt_d.T, sa_d, sb_d, torch.bfloat16)
    # EVERY element of the headline output against the oracle, per element (round 4: rounds 1-3 checked 96 rows here and the
    # rest against the build's own generic kernel; the oracle's fp32 matmul of the full problem is a second on the GPU box's host)
    got_h = got.cpu()
    for r0 in range(0, m, 1024):
        rows = slice(r0, r0 + 1024)
        ref = oracle.scaled_gemm_ref(a[rows], b, sa[rows], sb, torch.bfloat16, None)
        check_scaled(got_h[rows], ref, torch.float8_e4m3fn, torch.bfloat16, (a[rows], b, sa[rows], sb, None))
    _C.set_gemm_variant(_C.VARIANT_GENERIC)
    slow = scaled_gemm(a_d, bt_d.T, sa_d, sb_d, torch.bfloat16)
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    tol = 2.0 * EPS[torch.bfloat16] * slow.float().abs().max().item()
    assert (got.float() - slow.float()).abs().max().item() <= tol
    perm = torch.randperm(m, device="cuda")
    got_p = scaled_gemm(a_d[perm].contiguous(), bt_d.T, sa_d[perm].contiguous(), sb_d, torch.bfloat16)
    assert torch.equal(got_p, got[perm])


def test_scaled_gemm_c5_config_shard_invariance():
    """BASELINE config C5 (fp8 -> bf16, 8192x8192x28672, N-sharded over 8 ranks) at full size on one GPU.

    (1) the column block a rank would compute (N/8 = 3584 columns, rank 0, 3 and 7) from ITS slice of B / scale_b
    equals the same columns of the full product bit-for-bit -- the property the multi-GPU path rests on
    (no reduction, so sharding must not change a single bit); (2) a 64-row band of every tile row of
    the full product, every column, against the CPU oracle.
    """
    m, k, n, world = 8192, 8192, 28672, 8
    gen = torch.Generator(device="cuda").manual_seed(5)
    a = (0.25 * torch.rand((m, k), device="cuda", generator=gen)).to(torch.float8_e4m3fn)
    bt = (0.25 * torch.rand((n, k), device="cuda", generator=gen)).to(torch.float8_e4m3fn)
    sa = 0.25 * torch.rand((m, 1), device="cuda", generator=gen)
    sb = 0.25 * torch.rand((n, 1), device="cuda", generator=gen)
    full = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    shard = n // world
    for rank in (0, 3, 7):
        lo, hi = rank * shard, (rank + 1) * shard
        part = scaled_gemm(a, bt[lo:hi].T, sa, sb[lo:hi], torch.bfloat16)
        assert torch.equal(part, full[:, lo:hi]), f"rank {rank}"
    # against the oracle: every column of 2 048 rows -- one 64-row band out of every 256-row tile row of the kernels, at a
    # different offset inside each (round 6; rounds 4-5 ran all 8 192 rows through the CPU: 75 s of the suite's budget for rows
    # whose tiles differ from the checked ones in their data only; the shard identities above cover every element bit for bit)
    a_h, b_h, sa_h, sb_h = a.cpu(), bt.cpu().T, sa.cpu(), sb.cpu()
    rows = torch.cat([torch.arange(64) + 256 * t + 64 * (t % 4) for t in range(m // 256)])
    got_rows = full[rows.cuda()].cpu()
    for r0 in range(0, rows.numel(), 1024):
        sel = rows[r0:r0 + 1024]
        ref = oracle.scaled_gemm_ref(a_h[sel], b_h, sa_h[sel], sb_h, torch.bfloat16, None)
        check_scaled(got_rows[r0:r0 + 1024], ref, torch.float8_e4m3fn, torch.bfloat16, (a_h[sel], b_h, sa_h[sel], sb_h, None))


# ---------------------------------------------------------------------------------------------
# scaled_gemm_silu_and_mul: the FFN pair in one launch
# ---------------------------------------------------------------------------------------------
MANT = {torch.float16: 10, torch.bfloat16: 7}


def check_silu(got, ref, out_dtype, exact_gemm):
    """Against the CPU oracle (torch silu: libm-grade expf; device: v_exp_f32 / v_rcp_f32, 1 ulp each in fp32).

    When the GEMM part is exact (int8): every element within TWO output ulps of the oracle (silu rounded to the output
    dtype may land on the neighbouring value, which moves the rounded product by up to two ulps) and at most 0.2 % of the
    elements off at all; with fp8 inputs the accumulation-order tolerance of the plain GEMM (2 eps of max|G|) feeds
    through silu'(g) * u, so the bound is relative to max|y| like check_scaled's.
    """
    g, r = got.float().cpu(), ref.float()
    # overflowed elements (fp16 output of the int8 recipe): same infinities / NaNs in the same places, bounds on the rest
    fin = torch.isfinite(r)
    assert torch.equal(torch.isnan(g), torch.isnan(r))
    assert torch.equal(g[~fin & ~torch.isnan(r)], r[~fin & ~torch.isnan(r)])
    g, r = g[fin], r[fin]
    if exact_gemm:
        ulp = torch.maximum(torch.ldexp(torch.ones_like(r), torch.frexp(r)[1] - 1 - MANT[out_dtype]),
                            torch.full_like(r, 2.0 ** -24))
        assert ((g - r).abs() <= 2 * ulp).all(), f"max |diff| / ulp = {((g - r).abs() / ulp).max().item():.2f}"
        assert (g != r).float().mean().item() <= 2e-3
    else:
        tol = 4.0 * EPS[out_dtype] * max(r.abs().max().item(), 1e-6)
        assert (g - r).abs().max().item() <= tol


def run_silu(a, b, sa, sb, out_dtype, bias):
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    b_dev = b.T.contiguous().cuda().T if b.stride(0) == 1 else b.cuda()
    return scaled_gemm_silu_and_mul(dev(a), b_dev, dev(sa), dev(sb), out_dtype, dev(bias))


@pytest.mark.parametrize("iname", list(IN_T))
@pytest.mark.parametrize("oname", ["f16", "bf16"])
@pytest.mark.parametrize("key_tail", ["sa1_sb1_b1", "sa0_sb0_b1", "sa0_sb0_b0", "sa1_sb0_b0"])
def test_scaled_gemm_silu_golden_from_reference(golden, iname, oname, key_tail):
    """Inputs of the reference-generated scaled_gemm fixtures, expected outputs = the reference's silu_and_mul of the
    reference's scaled_gemm (tests/golden/scaled_gemm_silu.npz).  e4m3fnuz takes the unfused fallback."""
    g, y = golden("scaled_gemm"), golden("scaled_gemm_silu")
    key = f"{iname}_{oname}_{key_tail}"
    a = from_bits(g[f"a_{key}"], IN_T[iname])
    b = from_bits(g[f"bt_{key}"], IN_T[iname]).T
    sa, sb = torch.from_numpy(g[f"sa_{key}"]), torch.from_numpy(g[f"sb_{key}"])
    bias = from_bits(g[f"bias_{key}"], DT[oname]) if key_tail.endswith("b1") else None
    got = run_silu(a, b, sa, sb, DT[oname], bias)
    assert got.shape == (128, 64) and got.dtype == DT[oname]
    check_silu(got, from_bits(y[f"y_{key}"], DT[oname]), DT[oname], exact_gemm=iname == "int8")


@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "d"), [(1024, 1024, 512), (300, 384, 260), (257, 256, 8), (512, 1152, 1376), (64, 2048, 96),
                                            (16, 1024, 1376), (33, 2048, 260)])
def test_scaled_gemm_silu_shapes_fused_equals_unfused(iname, m, k, d):
    """Ragged M / d (partial tiles, d not a multiple of 8 or 128): (1) against the oracle; (2) the fused kernel equals the
    library's own unfused pair (plain GEMM into scratch + elementwise pass, forced with variant 2) bit for bit -- the
    fused tile only re-pairs columns, every output sees the same MFMA sequence and the same epilogue arithmetic."""
    a, b, sa, sb, bias = make_scaled_inputs(m, k, 2 * d, IN_T[iname], torch.bfloat16, False, False, True)
    ref = oracle.scaled_gemm_silu_and_mul_ref(a, b, sa, sb, torch.bfloat16, bias)
    got = run_silu(a, b, sa, sb, torch.bfloat16, bias)
    check_silu(got, ref, torch.bfloat16, exact_gemm=iname == "int8")
    _C.set_gemm_variant(_C.VARIANT_MFMA_SIMPLE)
    unfused = run_silu(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    if m <= 256 and iname != "int8":
        # skinny M: auto may run the split-K kernel with the silu fused into its reduce kernel (the cost model decides) --
        # another fp32 summation order than the tiled kernel of the unfused pair
        tol = 4.0 * EPS[torch.bfloat16] * unfused.float().abs().max().item()
        assert (got.float() - unfused.float()).abs().max().item() <= tol
    else:
        assert torch.equal(got, unfused)


def test_scaled_gemm_silu_llama_ffn_shape():
    """The C3 weights as a gate/up pair (Llama-7B FFN: K = 4096, d = 11008, 4096 tokens) at full size: fused equals
    unfused bit for bit everywhere; bands of rows against the oracle."""
    m, k, d = 4096, 4096, 11008
    gen = torch.Generator(device="cuda").manual_seed(11)
    a = (0.25 * torch.rand((m, k), device="cuda", generator=gen)).to(torch.float8_e4m3fn)
    bt = (0.25 * torch.rand((2 * d, k), device="cuda", generator=gen) - 0.125).to(torch.float8_e4m3fn)
    sa = 0.25 * torch.rand((m, 1), device="cuda", generator=gen)
    sb = 0.25 * torch.rand((2 * d, 1), device="cuda", generator=gen)
    got = scaled_gemm_silu_and_mul(a, bt.T, sa, sb, torch.bfloat16)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SIMPLE)
    unfused = scaled_gemm_silu_and_mul(a, bt.T, sa, sb, torch.bfloat16)
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    assert torch.equal(got, unfused)
    rows = torch.cat([torch.arange(0, 24), torch.arange(4072, 4096)])
    ref = oracle.scaled_gemm_silu_and_mul_ref(a[rows.cuda()].cpu(), bt.cpu().T, sa[rows.cuda()].cpu(), sb.cpu(), torch.bfloat16, None)
    check_silu(got[rows.cuda()], ref, torch.bfloat16, exact_gemm=False)


# ---------------------------------------------------------------------------------------------
# mixed precision
# ---------------------------------------------------------------------------------------------
WTYPES = {"uint4b8": scalar_types.uint4b8, "uint8b128": scalar_types.uint8b128, "uint4": scalar_types.uint4,
          "uint8": scalar_types.uint8}


@functools.lru_cache(maxsize=4)
def _quantized_weights(k, n, wname, use_zp, dtype, group):
    """Host-side quantisation is the slow part of these tests; cache it across M and activations."""
    gen = torch.Generator().manual_seed(1234)
    b = (10 * (torch.rand((k, n), dtype=torch.float32, generator=gen) - 0.3)).to(dtype)
    wtype = WTYPES[wname]
    w_ref, w_q, w_s, w_zp = quantize_weights(b, wtype, group, zero_points=use_zp)
    packed = pack_rows(w_q, wtype.size_bits, *w_q.shape)
    return w_ref, packed, w_s, w_zp


def make_mixed_inputs(m, k, n, wtype, use_zp, dtype, group=128):
    """Distributions of tests/mixed_precision_gemm_test.py:57-70 (weights drawn from their own stream)."""
    seed_everything(0)
    a = (10 * (torch.rand((m, k), dtype=torch.float32) - 0.3)).to(dtype)
    wname = next(name for name, t in WTYPES.items() if t is wtype)
    w_ref, packed, w_s, w_zp = _quantized_weights(k, n, wname, use_zp, dtype, group)
    return a, w_ref, packed, w_s, w_zp


ORACLE_HALF_MACS = 1 << 27  # budget of ONE half-precision CPU matmul of the oracle (the GPU box's host runs fp16 matmuls at ~0.2 GMAC/s)


def check_mixed(got, a, w_ref, k):
    """The oracle is torch.matmul(a, w_ref) in the activation dtype; ours accumulates in fp32.

    Bound: |diff| <= 2 eps_out * max|C| (accumulation order + the oracle's own half-precision
    reduction), far inside the reference's rtol=1e-1 / atol=min(5e-2*sqrt(K), 1).

    EVERY element is held to the per-element bound against the fp64 product of the same operands.  The oracle's own
    half-precision matmul is run on every element too while m*k*n stays inside ORACLE_HALF_MACS; above that (fp16 only: the
    host CPU of the GPU box has no fast fp16 matmul -- 64 x 11008 x 4096 takes 16 s there, 0.1 s in bf16) on an evenly spread
    subset of the columns, every row (round 6; a column of C depends on its own column of w_ref only)."""
    m, n = got.shape
    g = got.float().cpu()
    exact = (a.double() @ w_ref.double()).float()
    scale = exact.abs().max().item()
    eps = EPS[got.dtype]
    assert (g - exact).abs().max().item() <= 1.0 * eps * scale
    # per element (an error confined to small-magnitude outputs cannot hide under max|C|): one output rounding of the exact
    # value plus the worst-case fp32 accumulation-order error of ITS products
    s_abs = (a.double().abs() @ w_ref.double().abs()).float()
    bound = eps * exact.abs() + (k * 2.0**-24) * s_abs + 1e-30
    excess = ((g - exact).abs() - bound).max().item()
    assert excess <= 0, f"per-element bound exceeded by {excess:.4g}"
    cols = slice(None)
    if got.dtype == torch.float16 and m * k * n > ORACLE_HALF_MACS:
        keep = max(16, ORACLE_HALF_MACS // (m * k))
        cols = torch.linspace(0, n - 1, keep).long().unique()
    ref = oracle.mixed_precision_gemm_ref(a, w_ref[:, cols].contiguous()).float()
    assert (g[:, cols] - ref).abs().max().item() <= 3.0 * eps * scale
    atol = min(5e-2 * math.sqrt(k), 1)
    torch.testing.assert_close(g[:, cols], ref, rtol=1e-1, atol=atol)  # the reference's own bar


def check_mixed_whole(got, a, w_ref, k, block=1024):
    """EVERY element of `got` against the fp64 product of the reference's own operands, a.double() @ w_ref.double() (w_ref =
    quantize_weights' dequantised weights, conch/third_party/vllm/quant_utils.py:74: the tensor tests/mixed_precision_gemm_test.py:70
    multiplies), with check_mixed's per-element bound: one rounding of the exact value to the output dtype plus the worst-case fp32
    accumulation-order error of that element's own products.  The half-precision CPU matmul of the oracle is minutes at these
    sizes (it stays on bands of rows); the fp64 product is seconds -- in blocks of rows to bound the host memory."""
    eps = EPS[got.dtype]
    g = got.float().cpu()
    wd = w_ref.double()
    wabs = wd.abs()
    worst = -1e30
    for r0 in range(0, a.shape[0], block):
        ad = a[r0:r0 + block].double()
        exact = ad @ wd
        bound = eps * exact.abs() + (k * 2.0**-24) * (ad.abs() @ wabs) + 1e-30
        worst = max(worst, ((g[r0:r0 + block].double() - exact).abs() - bound).max().item())
    assert worst <= 0, f"per-element bound exceeded by {worst:.4g}"


@pytest.mark.parametrize(("m", "k", "n"), SHAPES)
@pytest.mark.parametrize("wname", list(WTYPES))
@pytest.mark.parametrize("use_zp", [True, False])
@pytest.mark.parametrize("dname", ["f16", "bf16"])
def test_mixed_precision_gemm_matrix(m, k, n, wname, use_zp, dname):
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    got = mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(),
                               wt.size_bits, wt.bias, 128)
    assert got.shape == (m, n) and got.dtype == DT[dname]
    if m > 1024:
        # half-precision matmul on the CPU is very slow (minutes for 4096 rows): check three bands of rows
        rows = torch.cat([torch.arange(0, 64), torch.arange(m // 2 - 32, m // 2 + 32), torch.arange(m - 64, m)])
        check_mixed(got[rows.cuda()], a[rows], w_ref, k)
        check_mixed_whole(got, a, w_ref, k)  # ... and every element against the fp64 product
    else:
        check_mixed(got, a, w_ref, k)


@pytest.mark.parametrize(("m", "k", "n"), [(512, 512, 1024), (300, 384, 520), (1024, 1024, 1376), (257, 256, 200), (768, 128, 192)])
@pytest.mark.parametrize("wname", list(WTYPES))
@pytest.mark.parametrize("use_zp", [True, False])
@pytest.mark.parametrize("dname", ["f16", "bf16"])
def test_mixed_strip_kernel_is_bit_identical_and_correct(_reset_tuning, m, k, n, wname, use_zp, dname):
    """The column-strip kernel (round 4, gemm_mixed_strip.hip: every wave dequantises its own 16 / 32 columns straight into
    MFMA operand registers; X, the packed words, the scales and the zero points all arrive by LDS-DMA behind one counted wait
    per K step) against the LDS-tiled kernel: the same dequantised weights (ChunkDequant), the same MFMA, the same order of K
    per output -- the same bits, at every tile width, with ragged M / N, group boundaries inside and between steps (K = 128:
    one group; 384: three), and one to sixteen K steps (prologue / peeled last steps)."""
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    args = (a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), wt.size_bits, wt.bias, 128)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)  # any MFMA value: the tile kernels, never the decode-batch kernel
    _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, 1)
    tiled = mixed_precision_gemm(*args)
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, 2)
    for nt in (4, 3, 2):
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
        for _ in range(2):
            strip = mixed_precision_gemm(*args)
            assert torch.equal(strip, tiled), f"tile width {64 * nt}: {(strip != tiled).sum().item()} elements differ"
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
    check_mixed(mixed_precision_gemm(*args), a, w_ref, k)


@pytest.mark.parametrize(("m", "k", "n"), [(64, 1024, 512), (33, 512, 200), (128, 2048, 704), (100, 1152, 1000), (256, 1024, 384), (200, 768, 260),
                                           (48, 128, 192), (1, 256, 64)])
@pytest.mark.parametrize("wname", list(WTYPES))
@pytest.mark.parametrize("use_zp", [True, False])
@pytest.mark.parametrize("dname", ["f16", "bf16"])
def test_mixed_strip_kernel_short_tiles_and_k_split(_reset_tuning, m, k, n, wname, use_zp, dname):
    """Round 5, batched-decode sizes: the column-strip kernel on 64- / 128-row tiles (4 / 8 m tiles per wave: several
    dequantisation slices and requests per MFMA slot, deeper rings, more X fragments in flight) and with K split over
    workgroups (fp32 slabs [slice][M][N], added in slice order by the fp32-slab reduce).  Unsplit, every tile height and width
    gives the LDS-tiled kernel's bits (same dequantised weights, same MFMA, same order of K per output); split, the sums are
    grouped by slice -- checked against the oracle with the per-element bound of check_mixed -- and do not depend on the tile
    shape (the slices are the same K ranges).  Ragged M / N / last slice, one to 32 K steps, groups inside and across slices."""
    if dname == "bf16" and WTYPES[wname].size_bits == 8:
        pytest.skip("bf16 x 8-bit weights: the LDS-tiled kernel keeps that combination (mixed_gemm_strip_supported)")
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    args = (a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), wt.size_bits, wt.bias, 128)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, 1)
    tiled = mixed_precision_gemm(*args)
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, 2)
    for rows in (64, 128, 256):
        _C.set_tuning(_C.TUNE_MIXED_STRIP_ROWS, rows)
        for nt in (4, 3, 2):
            _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
            strip = mixed_precision_gemm(*args)
            assert torch.equal(strip, tiled), f"{rows}-row x {64 * nt}-column tiles: {(strip != tiled).sum().item()} elements differ"
    if n % 4 == 0:
        for split in (2, 3, 8):
            if k // 128 < split:
                continue
            _C.set_tuning(_C.TUNE_MIXED_SPLITK, split)
            first = None
            for rows, nt in ((64, 2), (128, 3), (256, 4), (0, 0)):
                _C.set_tuning(_C.TUNE_MIXED_STRIP_ROWS, rows)
                _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
                got = mixed_precision_gemm(*args)
                if first is None:
                    first = got
                    check_mixed(got, a, w_ref, k)
                else:
                    assert torch.equal(got, first), f"{split} slices: {rows}-row x {64 * nt}-column tiles differ from 64 x 128"
                assert torch.equal(mixed_precision_gemm(*args), got), f"{split} slices, {rows} rows: a second launch differs"


@pytest.mark.parametrize(("m", "k", "n"), [(64, 4096, 11008), (128, 4096, 11008), (96, 8192, 8192), (48, 4096, 14336), (200, 5120, 13824)])
def test_mixed_precision_gemm_batched_decode_sizes_take_the_strip_forms(_reset_tuning, m, k, n):
    """The dispatcher's own pick at 33..256 rows on wide / deep problems (dispatch_fit.hpp MixedStripSplit) against the oracle,
    every element (fp64 product of the reference's operands, per-element bound), on the benchmark's weight type."""
    wt = WTYPES["uint4b8"]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, False, torch.float16)
    got = mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.cuda(), None, wt.size_bits, wt.bias, 128)
    check_mixed_whole(got, a, w_ref, k)


@pytest.mark.parametrize(("m", "k", "n"), [(128, 256, 128), (33, 384, 200)])
@pytest.mark.parametrize("wname", list(WTYPES))
@pytest.mark.parametrize("use_zp", [True, False])
def test_mixed_precision_gemm_fp32_activations(m, k, n, wname, use_zp):
    """fp32 activations (the reference's benchmark offers them: benchmarks/mixed_precision_gemm_benchmark.py:99-105): scales and
    result fp32, weights dequantised in fp32 (= w_ref), fp32 accumulation -- the generic kernel (gfx950 has no fp32 matrix rate to
    speak of).  Against the oracle's fp32 matmul: the worst-case accumulation-order error of each output's own products."""
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, torch.float32)
    assert w_s.dtype == torch.float32 and w_ref.dtype == torch.float32
    got = mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), wt.size_bits, wt.bias, 128)
    assert got.dtype == torch.float32 and got.shape == (m, n)
    ref = oracle.mixed_precision_gemm_ref(a, w_ref)
    exact = a.double() @ w_ref.double()
    bound = (k * 2.0**-23) * (a.double().abs() @ w_ref.double().abs()) + 1e-30
    assert ((got.cpu().double() - exact).abs() <= bound).all()
    assert ((ref.double() - exact).abs() <= bound).all()  # the oracle sits inside the same bound
    with pytest.raises(NotImplementedError):  # fp32 activations need fp32 scales (the meta dtype is the activation dtype)
        mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.half().cuda(), None, wt.size_bits, wt.bias, 128)


def test_mixed_precision_c4_config_properties():
    """BASELINE config C4 (uint4b8, group 128, fp16, 1024x4096x11008) at full size.

    (1) bands of rows against the oracle; (2) the MFMA kernel against the generic device kernel everywhere (same
    dequantised weights bit-for-bit, accumulation order differs); (3) permuting the rows of X permutes the rows of C
    bit-for-bit; (4) a column block computed from its own slice of the packed weights / scales equals the same
    columns of the full product bit-for-bit (N-sharding invariance, 11008 / 8 = 1376 columns: ragged 192-wide tiles).
    """
    m, k, n = 1024, 4096, 11008
    wt = scalar_types.uint4b8
    a, w_ref, packed, w_s, _ = make_mixed_inputs(m, k, n, wt, False, torch.float16)
    a_d, p_d, s_d = a.cuda(), packed.cuda(), w_s.cuda()
    got = mixed_precision_gemm(a_d, p_d, s_d, None, wt.size_bits, wt.bias, 128)
    rows = torch.cat([torch.arange(0, 64), torch.arange(480, 544), torch.arange(960, 1024)])
    check_mixed(got[rows.cuda()], a[rows], w_ref, k)
    check_mixed_whole(got, a, w_ref, k)  # (1b) every element against the fp64 product of the reference's operands
    _C.set_gemm_variant(_C.VARIANT_GENERIC)
    slow = mixed_precision_gemm(a_d, p_d, s_d, None, wt.size_bits, wt.bias, 128)
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    tol = 2.0 * EPS[torch.float16] * slow.float().abs().max().item()
    assert (got.float() - slow.float()).abs().max().item() <= tol
    perm = torch.randperm(m, device="cuda")
    assert torch.equal(mixed_precision_gemm(a_d[perm].contiguous(), p_d, s_d, None, wt.size_bits, wt.bias, 128), got[perm])
    shard = n // 8
    for rank in (0, 5, 7):
        lo, hi = rank * shard, (rank + 1) * shard
        shard_args = (a_d, p_d[:, lo:hi].contiguous(), s_d[:, lo:hi].contiguous(), None, wt.size_bits, wt.bias, 128)
        # a shard has 32 tiles where the full problem has 232: the dispatcher gives it K slices (another fp32 summation order).
        # The invariance is a property of ONE kernel form -- asserted with the split off -- and the automatic choice stays
        # within the accumulation-order tolerance.
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)
        try:
            part = mixed_precision_gemm(*shard_args)
        finally:
            _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)
        assert torch.equal(part, got[:, lo:hi]), f"rank {rank}"
        auto = mixed_precision_gemm(*shard_args)
        assert (auto.float() - part.float()).abs().max().item() <= 2.0 * EPS[torch.float16] * part.float().abs().max().item()


def test_mixed_precision_readme_shape():
    """The shape behind the reference's ONLY published mixed-GEMM number (README.md:70, 1.656 ms on H100 for Conch's Triton
    kernel): the benchmark's defaults M = 4096, K = 8192, N = 4096, fp16 x uint4b8, group 128, no zero points
    (benchmarks/mixed_precision_gemm_benchmark.py:82-110, 187), inputs by its recipe (:205-214).  Bands of rows against the
    oracle (torch.matmul(a, w_ref) in fp16), every element against the fp64 product, and the two kernels that can serve the shape
    (LDS-tiled: the automatic choice at N = 4096; column-strip) against each other bit for bit."""
    m, k, n = 4096, 8192, 4096
    wt = scalar_types.uint4b8
    a, w_ref, packed, w_s, _ = make_mixed_inputs(m, k, n, wt, False, torch.float16)
    args = (a.cuda(), packed.cuda(), w_s.cuda(), None, wt.size_bits, wt.bias, 128)
    got = mixed_precision_gemm(*args)
    assert got.shape == (m, n) and got.dtype == torch.float16
    rows = torch.cat([torch.arange(0, 32), torch.arange(2032, 2064), torch.arange(4064, 4096)])
    check_mixed(got[rows.cuda()], a[rows], w_ref, k)
    check_mixed_whole(got, a, w_ref, k)
    try:
        _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, 1)
        tiled = mixed_precision_gemm(*args)
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, 2)
        strip = mixed_precision_gemm(*args)
    finally:
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, 0)
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)
        _C.set_gemm_variant(_C.VARIANT_AUTO)
    assert torch.equal(tiled, strip)
    assert torch.equal(got, tiled) or torch.equal(got, strip)


@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4", True, "bf16"), ("uint8b128", False, "f16"), ("uint8", True, "bf16"),
                                                         ("uint4b8", False, "bf16"), ("uint4", True, "f16")])
def test_mixed_precision_c4_size_other_weight_types(wname, use_zp, dname):
    """The C4 shape (1024 x 4096 x 11008) with zero points, 8-bit weights and bf16 -- the instantiations BASELINE's own
    configuration does not touch: bands of rows against the oracle, the whole product against the generic device kernel."""
    m, k, n = 1024, 4096, 11008
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    args = (a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), wt.size_bits, wt.bias, 128)
    got = mixed_precision_gemm(*args)
    rows = torch.cat([torch.arange(0, 32), torch.arange(500, 532), torch.arange(992, 1024)])
    check_mixed(got[rows.cuda()], a[rows], w_ref, k)
    check_mixed_whole(got, a, w_ref, k)
    _C.set_gemm_variant(_C.VARIANT_GENERIC)
    slow = mixed_precision_gemm(*args)
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    assert (got.float() - slow.float()).abs().max().item() <= 2.0 * EPS[DT[dname]] * slow.float().abs().max().item()


def test_scaled_gemm_c3_size_e4m3fnuz():
    """The dtype the REFERENCE hands an AMD user (conch/ops/quantization/fp8.py:27) at the headline shape: exact expansion to
    bf16 + the bf16 MFMA tile kernel.  Bands of rows against the oracle, per element."""
    m, k, n = 4096, 4096, 11008
    a, b, sa, sb, _ = make_scaled_inputs(m, k, n, torch.float8_e4m3fnuz, torch.bfloat16, False, False, False)
    got = scaled_gemm(a.cuda(), b.T.contiguous().cuda().T, sa.cuda(), sb.cuda(), torch.bfloat16)
    rows = torch.cat([torch.arange(0, 48), torch.arange(2040, 2072), torch.arange(4080, 4096)])
    ref = oracle.scaled_gemm_ref(a[rows], b, sa[rows], sb, torch.bfloat16, None)
    check_scaled(got[rows.cuda()], ref, torch.float8_e4m3fnuz, torch.bfloat16, (a[rows], b, sa[rows], sb, None))


@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "n", "n_dest", "use_bias"), [(512, 512, 512, 4, False), (1024, 256, 768, 8, True), (300, 384, 200, 3, True),
                                                                  (256, 1024, 256, 2, False), (2048, 512, 1792, 8, False)])
def test_scaled_gemm_multi_destinations_are_bit_identical(iname, m, k, n, n_dest, use_bias):
    """conch_scaled_gemm_multi (round 4, the direct all-gather of the N-sharded form): ONE launch stores the product into up to
    eight buffers -- here column blocks of eight local [M, 3N] "results" standing in for the peer-mapped ones.  Every
    destination must hold exactly what scaled_gemm returns, whole and ragged tiles alike, and nothing outside its block may be
    touched."""
    from conch_amd.kernels.quantization.gemm import scaled_gemm_multi_launcher
    from conch_amd.ops.quantization.gemm import create_scaled_metadata

    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, use_bias)
    ad, bd, sad, sbd = a.cuda(), b.T.contiguous().cuda().T, sa.cuda(), sb.cuda()
    biasd = None if bias is None else bias.cuda()
    want = scaled_gemm(ad, bd, sad, sbd, torch.bfloat16, biasd)
    lo = n  # the middle third of each result
    results = [torch.full((m, 3 * n), -7.0, dtype=torch.bfloat16, device="cuda") for _ in range(n_dest)]
    meta = create_scaled_metadata(ad, bd, sad, sbd, torch.bfloat16)
    for _ in range(2):
        scaled_gemm_multi_launcher([r[:, lo:lo + n] for r in results], ad, bd, sad, sbd, meta, bias=biasd)
        for r in results:
            assert torch.equal(r[:, lo:lo + n], want)
            assert bool((r[:, :lo] == -7.0).all()) and bool((r[:, lo + n:] == -7.0).all())
    with pytest.raises(ValueError):  # destinations of different strides
        scaled_gemm_multi_launcher([results[0][:, lo:lo + n], torch.empty((m, n), dtype=torch.bfloat16, device="cuda")], ad, bd, sad, sbd, meta)


@pytest.mark.parametrize(("m", "k", "n"), [(1024, 1024, 1024), (4096, 512, 2048), (600, 384, 1376),
                                            # decode batches (split-K skinny kernels, one- and two-launch forms) and few tiles with a
                                            # long K (the 128 x 128 tiles' split-K form): these run on a scaled copy of scale_a
                                            (16, 4096, 4096), (128, 4096, 1024), (64, 2048, 520), (8, 1024, 24), (256, 11008, 512), (200, 8192, 260)])
@pytest.mark.parametrize("where", ["none", "a", "b", "both", "nan"])
def test_scaled_gemm_e4m3fnuz_special_codes_take_the_exact_path(m, k, n, where):
    """e4m3fnuz at the fp8 rate (round 4): the tile kernels run the raw bytes on the OCP fp8 MFMA (every code is twice its
    fnuz value there, the exact 1/4 rides on scale_a) -- except the three codes that do not map: 0x7F / 0xFF are +-240 in fnuz
    and NaN in OCP, 0x80 is NaN in fnuz and -0 in OCP.  A scan of both operands sets a device-side flag and the launch falls
    to the exact bf16 expansion; the host never reads the flag.  Saturated operands (what a clamping quantiser emits) must
    come out right, per element, wherever the codes sit; a fnuz NaN must poison exactly its row / column."""
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, torch.float8_e4m3fnuz, torch.bfloat16, False, False, True)
    ab, bb = a.view(torch.uint8).clone(), b.T.contiguous().view(torch.uint8).clone()  # [M][K], [N][K]
    gen = torch.Generator().manual_seed(3)
    if where in ("a", "both"):
        idx = torch.randint(0, m * k, (37,), generator=gen)
        ab.view(-1)[idx] = torch.where(torch.rand(37, generator=gen) < 0.5, 0x7F, 0xFF).to(torch.uint8)
    if where in ("b", "both"):
        bb[n - 1, k - 1] = 0xFF  # the very last byte of the operand
        bb[5, 17] = 0x7F
    if where == "nan":
        ab[3, 100] = 0x80
        bb[7, 1] = 0x80
    a2, b2 = ab.view(torch.float8_e4m3fnuz), bb.view(torch.float8_e4m3fnuz).T
    ref = oracle.scaled_gemm_ref(a2, b2, sa, sb, torch.bfloat16, bias)
    for _ in range(2):  # twice: the flag is re-made by every call
        got = run_scaled(a2, b2, sa, sb, torch.bfloat16, bias)
        if where == "nan":
            bad = torch.zeros((m, n), dtype=torch.bool)
            bad[3, :] = True
            bad[:, 7] = True
            assert torch.equal(torch.isnan(got.float().cpu()), bad) and torch.equal(torch.isnan(ref.float()), bad)
            ok = ~bad
            assert (got.float().cpu()[ok] - ref.float()[ok]).abs().max().item() <= 2.0 * EPS[torch.bfloat16] * ref.float()[ok].abs().max().item()
        else:
            check_scaled(got, ref, torch.float8_e4m3fnuz, torch.bfloat16, (a2, b2, sa, sb, bias))
    # and a clean call right behind a flagged one takes the fp8 kernel again (same bound)
    clean = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
    check_scaled(clean, oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias), torch.float8_e4m3fnuz, torch.bfloat16, (a, b, sa, sb, bias))


@pytest.mark.parametrize(("m", "k", "n"), [(1, 4096, 4096), (16, 4096, 11008), (33, 1024, 520), (64, 2048, 2048), (128, 4096, 1024)])
def test_e4m3fnuz_decode_sizes_split_the_special_codes_in_registers(m, k, n):
    """Round 6: at decode sizes the split-K kernel takes e4m3fnuz bytes as they are -- no scan of the operands, no flag, no bf16
    expansion -- and splits +-240 (0x7F / 0xFF: OCP's NaN) and the fnuz NaN (0x80: OCP's -0) off in registers (gemm_skinny.hip,
    fnuz_split): x = clean + rem, up to four MFMAs per fragment pair, the exact 1/4 behind the K loop.  Data with +-240 in EVERY
    row of A and EVERY column of B (what an absmax quantiser emits) against the oracle per element, and against round 5's path
    (scan + gated kernels; conch_debug_fnuz_in_registers(0)) within the accumulation-order bound; a NaN poisons its row / column."""
    import ctypes

    hook = _C.load().conch_debug_fnuz_in_registers
    hook.restype, hook.argtypes = ctypes.c_int, [ctypes.c_int]
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, torch.float8_e4m3fnuz, torch.bfloat16, False, False, True)
    ab, bb = a.view(torch.uint8).clone(), b.T.contiguous().view(torch.uint8).clone()  # [M][K], [N][K]
    ra, rb = torch.arange(m), torch.arange(n)
    ab[ra, (ra * 37) % k] = torch.where(ra % 2 == 0, 0x7F, 0xFF).to(torch.uint8)
    bb[rb, (rb * 53) % k] = torch.where(rb % 3 == 0, 0xFF, 0x7F).to(torch.uint8)
    bb[n - 1, k - 1] = 0x7F  # the very last byte of the operand
    ab[0, 5], bb[0, 5] = 0x7F, 0xFF  # both operands saturated at the same k: the rem x rem product
    a2, b2 = ab.view(torch.float8_e4m3fnuz), bb.view(torch.float8_e4m3fnuz).T
    ref = oracle.scaled_gemm_ref(a2, b2, sa, sb, torch.bfloat16, bias)
    got = run_scaled(a2, b2, sa, sb, torch.bfloat16, bias)
    check_scaled(got, ref, torch.float8_e4m3fnuz, torch.bfloat16, (a2, b2, sa, sb, bias))
    hook(0)
    try:
        old = run_scaled(a2, b2, sa, sb, torch.bfloat16, bias)
    finally:
        hook(1)
    check_scaled(old, ref, torch.float8_e4m3fnuz, torch.bfloat16, (a2, b2, sa, sb, bias))
    ab[m // 2, 9] = 0x80
    bb[3, k - 2] = 0x80
    a3, b3 = ab.view(torch.float8_e4m3fnuz), bb.view(torch.float8_e4m3fnuz).T
    got = run_scaled(a3, b3, sa, sb, torch.bfloat16, bias).float().cpu()
    bad = torch.zeros((m, n), dtype=torch.bool)
    bad[m // 2, :] = True
    bad[:, 3] = True
    assert torch.equal(torch.isnan(got), bad)
    ref3 = oracle.scaled_gemm_ref(a3, b3, sa, sb, torch.bfloat16, bias).float()
    assert torch.equal(torch.isnan(ref3), bad)
    if (~bad).any():  # (one row: everything is poisoned)
        assert (got[~bad] - ref3[~bad]).abs().max().item() <= 2.0 * EPS[torch.bfloat16] * ref3[~bad].abs().max().item()


@pytest.mark.parametrize("wname", list(WTYPES))
@pytest.mark.parametrize("zp", [1, 0])
@pytest.mark.parametrize("dname", ["f16", "bf16"])
def test_mixed_precision_gemm_golden_from_reference(golden, wname, zp, dname):
    g = golden("mixed_gemm")
    key = f"{wname}_zp{zp}_{dname}"
    bits, bias, group = (int(v) for v in g[f"meta_{key}"])
    dtype = DT[dname]
    a = from_bits(g[f"a_{key}"], dtype)
    packed = torch.from_numpy(g[f"packed_{key}"].copy())
    w_s = from_bits(g[f"ws_{key}"], dtype)
    w_zp = torch.from_numpy(g[f"wzp_{key}"].copy()) if zp else None
    got = mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), bits, bias,
                               group)
    check_mixed(got, a, from_bits(g[f"wref_{key}"], dtype), a.shape[1])


@pytest.mark.parametrize("nt", [2, 3, 4, 5])
@pytest.mark.parametrize(("m", "k", "n"), [(1024, 256, 1536), (1100, 128, 1000), (512, 384, 768), (2048, 128, 520)])
@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"), ("uint8b128", False, "bf16")])
def test_mixed_row_major_epilogue_is_bit_identical(_reset_tuning, nt, m, k, n, wname, use_zp, dname):
    """Whole tiles of the LDS-tiled mixed kernel leave through an LDS image and whole-line write-through stores (round 3); ragged
    tiles and the direct route (CONCH_TUNE_EPILOGUE = 1) store from the accumulator layout.  Same casts, same bits -- at every
    tile shape (a wave's share of a row is 64, 96 or 128 bytes: the image is built by the workgroup), with whole and ragged
    tiles in one launch."""
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
    try:
        _C.set_tuning(_C.TUNE_EPILOGUE, 1)
        direct = mixed_precision_gemm(dev(a), dev(packed), dev(w_s), dev(w_zp), wt.size_bits, wt.bias, 128)
        _C.set_tuning(_C.TUNE_EPILOGUE, 2)
        rows = mixed_precision_gemm(dev(a), dev(packed), dev(w_s), dev(w_zp), wt.size_bits, wt.bias, 128)
    finally:
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
        _C.set_tuning(_C.TUNE_EPILOGUE, 0)
    np.testing.assert_array_equal(to_bits(rows), to_bits(direct))
    check_mixed(rows, a, w_ref, k)


@pytest.mark.parametrize("variant", ["auto"])
@pytest.mark.parametrize("nt", [2, 3, 4, 5])
@pytest.mark.parametrize(("m", "k", "n"), [(300, 256, 520), (1024, 512, 1376), (64, 128, 200), (512, 64, 256), (1100, 384, 300)])
@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"),
                                                         ("uint8b128", True, "f16"), ("uint8", False, "bf16")])
def test_mixed_precision_every_tile_width(variant, nt, m, k, n, wname, use_zp, dname):
    """The MFMA kernel at each tile shape (256 rows x 64*nt columns; nt = 5: the 512 x 128 tile),
    ragged M / N and the shortest K (one to three K steps) included."""
    wt = WTYPES[wname]
    group = 64 if k < 128 else 128
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname], group)
    _C.set_gemm_variant(VARIANTS[variant])
    _C.check(_C.load().conch_set_tuning(1, nt), "set_tuning")
    try:
        got = mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(),
                                   wt.size_bits, wt.bias, group)
    finally:
        _C.load().conch_set_tuning(1, 0)
    check_mixed(got, a, w_ref, k)


@pytest.mark.parametrize("variant", ["auto", "generic"])
def test_mixed_precision_dequant_is_bit_exact(variant):
    """X = identity isolates the dequantisation: C must equal w_ref bit-for-bit, MFMA and generic kernels."""
    k = n = 256
    _C.set_gemm_variant(VARIANTS[variant])
    for dname in ("f16", "bf16"):
        for wname, wt in WTYPES.items():
            for use_zp in (False, True):
                _, w_ref, packed, w_s, w_zp = make_mixed_inputs(8, k, n, wt, use_zp, DT[dname])
                eye = torch.eye(k, dtype=DT[dname])
                got = mixed_precision_gemm(eye.cuda(), packed.cuda(), w_s.cuda(),
                                           None if w_zp is None else w_zp.cuda(), wt.size_bits, wt.bias, 128)
                np.testing.assert_array_equal(to_bits(got), to_bits(w_ref), err_msg=f"{dname} {wname} zp={use_zp}")


def test_mixed_precision_scalar_zero_point_and_output_dtype():
    wt = scalar_types.uint4
    a, _, packed, w_s, _ = make_mixed_inputs(64, 256, 128, wt, False, torch.float16)
    zp = torch.tensor([7], dtype=torch.int32)
    w = oracle.dequantize_packed(packed, w_s, zp, 4, 0, 128)
    got = mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.cuda(), zp.cuda(), 4, 0, 128, output_dtype=torch.bfloat16)
    assert got.dtype == torch.bfloat16
    exact = (a.double() @ w.double()).float()
    assert (got.float().cpu() - exact).abs().max().item() <= EPS[torch.bfloat16] * exact.abs().max().item()


# ---------------------------------------------------------------------------------------------
# mixed_precision_gemm_silu_and_mul (SURVEY.md 8(f) N3 for int4 / int8 weights)
# ---------------------------------------------------------------------------------------------
def check_mixed_silu(got, a, w_ref, min_representable=0.5):
    """y = silu(g) * u amplifies the GEMM's accumulation-order tolerance (2 eps of max|G|, check_mixed) by up to
    max(|u|, |g|): bound 4 eps * max|G|^2-scale, i.e. relative to max|y| of the exact pair; plus the pair computed from
    OUR plain GEMM output must match to 2 output ulps (hardware exp / rcp against libm, as for the scaled form)."""
    exact_g = a.double() @ w_ref.double()
    d = exact_g.shape[1] // 2
    exact = (torch.nn.functional.silu(exact_g[:, :d]) * exact_g[:, d:]).float()
    g = got.float().cpu()
    eps = EPS[got.dtype]
    # the test recipe's magnitudes (|G| ~ 1e3) overflow fp16 in the product: those elements must be inf / huge on both
    # sides; the bound is asserted on the representable ones
    ok = exact.abs() < 3.0e4 if got.dtype == torch.float16 else torch.ones_like(exact, dtype=torch.bool)
    assert ok.float().mean().item() >= min_representable
    assert torch.isfinite(g[ok]).all() and (g[~ok].abs() > 2.0e4).all()
    if ok.any():
        assert (g[ok] - exact[ok]).abs().max().item() <= 6.0 * eps * exact[ok].abs().max().item()


@pytest.mark.parametrize("wname", list(WTYPES))
@pytest.mark.parametrize("zp", [1, 0])
@pytest.mark.parametrize("dname", ["f16", "bf16"])
def test_mixed_precision_silu_golden_from_reference(golden, wname, zp, dname):
    """Fixture inputs generated by the reference; expected = the reference's silu_and_mul of its matmul(a, w_ref)."""
    g, y = golden("mixed_gemm"), golden("mixed_gemm_silu")
    key = f"{wname}_zp{zp}_{dname}"
    bits, bias, group = (int(v) for v in g[f"meta_{key}"])
    dtype = DT[dname]
    a = from_bits(g[f"a_{key}"], dtype)
    packed = torch.from_numpy(g[f"packed_{key}"].copy())
    w_s = from_bits(g[f"ws_{key}"], dtype)
    w_zp = torch.from_numpy(g[f"wzp_{key}"].copy()) if zp else None
    got = mixed_precision_gemm_silu_and_mul(a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), bits, bias, group)
    assert got.shape == (32, 64) and got.dtype == dtype
    want = from_bits(y[f"y_{key}"], dtype).float()
    # fp16: most products of the reference recipe (|G| ~ 1e3) overflow; compare the representable ones.  The oracle's G is a
    # half-precision CPU matmul (up to 4 bf16 ulps from the exact product), doubled by the product.
    ok = torch.isfinite(want) & (want.abs() < 3.0e4) if dtype == torch.float16 else torch.isfinite(want)
    if ok.any():
        tol = 12.0 * EPS[dtype] * want[ok].abs().max().item()
        assert (got.float().cpu()[ok] - want[ok]).abs().max().item() <= tol
    check_mixed_silu(got, a, from_bits(g[f"wref_{key}"], dtype), min_representable=0.0)


@pytest.mark.parametrize(("m", "k", "d"), [(300, 256, 260), (1024, 512, 688), (64, 128, 100), (512, 64, 128), (16, 1024, 260),
                                            (48, 2048, 1376)])
@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"),
                                                         ("uint8b128", True, "f16"), ("uint8", False, "bf16")])
def test_mixed_precision_silu_fused_equals_unfused(m, k, d, wname, use_zp, dname):
    """Ragged M / d and the shortest K: against the exact pair, and bit-for-bit against the library's own unfused pair
    (plain MFMA kernel at the 256-column tile into scratch + elementwise pass, forced with variant 2)."""
    wt = WTYPES[wname]
    group = 64 if k < 128 else 128
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, 2 * d, wt, use_zp, DT[dname], group)
    a = (a.float() * (0.02 if k >= 256 else 0.05)).to(DT[dname])  # |G| ~ 10..50: gate values where silu is not linear, products inside fp16
    args = (a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), wt.size_bits, wt.bias, group)
    got = mixed_precision_gemm_silu_and_mul(*args)
    check_mixed_silu(got, a, w_ref)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SIMPLE)
    _C.check(_C.load().conch_set_tuning(1, 4), "set_tuning")
    try:
        unfused = mixed_precision_gemm_silu_and_mul(*args)
    finally:
        _C.load().conch_set_tuning(1, 0)
        _C.set_gemm_variant(_C.VARIANT_AUTO)
    if m <= 256:
        # decode batch: the decode-batch kernel with the silu fused into its reduce kernel (other summation order)
        tol = 4.0 * EPS[DT[dname]] * unfused.float().abs().max().item()
        assert (got.float() - unfused.float()).abs().max().item() <= tol
    else:
        assert torch.equal(got, unfused)


# ---------------------------------------------------------------------------------------------
# decode-batch kernel of mixed_precision_gemm (M <= 64, gemm_mixed_skinny.hip)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize(("m", "k", "n"), [(1, 1024, 64), (16, 2048, 520), (20, 1024, 4096), (33, 4096, 1376), (64, 2048, 256), (48, 1024, 100),
                                            (8, 1152, 256), (40, 2944, 520), (16, 128, 64), (100, 1024, 256), (128, 2048, 520), (250, 1152, 256)])
@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"),
                                                         ("uint8b128", True, "f16"), ("uint8", False, "bf16")])
def test_mixed_precision_decode_batches(m, k, n, wname, use_zp, dname):
    """The automatic choice for M <= 64, N % 4 == 0, any K (a ragged last 1024-element slice included) (forced with variant 4 so that a contract change cannot
    silently skip it): against the oracle, and against the LDS-tiled kernel (other summation order: 2 eps of max|C|)."""
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    args = (a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), wt.size_bits, wt.bias, 128)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
    got = mixed_precision_gemm(*args)
    check_mixed(got, a, w_ref, k)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    tiled = mixed_precision_gemm(*args)
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    auto = mixed_precision_gemm(*args)
    assert torch.equal(auto, got if m <= 32 else auto)  # auto = the decode-batch kernel up to 32 rows (a cost rule above)
    if not (torch.equal(auto, got) or torch.equal(auto, tiled)):
        check_mixed(auto, a, w_ref, k)  # 33..256 rows: the K-split strip forms (their own K slices)
    tol = 2.0 * EPS[DT[dname]] * tiled.float().abs().max().item()
    assert (got.float() - tiled.float()).abs().max().item() <= tol


def test_mixed_precision_decode_dequant_is_bit_exact():
    """Unit-vector activations isolate the dequantisation in the decode-batch kernel: row i of C must be row j_i of w_ref
    bit for bit (the same claim test_mixed_precision_dequant_is_bit_exact makes for the tiled and generic kernels)."""
    k, n = 1024, 256
    rows = torch.tensor([0, 1, 7, 8, 31, 32, 63, 64, 127, 128, 500, 511, 512, 777, 1000, 1023])
    for dname in ("f16", "bf16"):
        for wname, wt in WTYPES.items():
            for use_zp in (False, True):
                _, w_ref, packed, w_s, w_zp = make_mixed_inputs(8, k, n, wt, use_zp, DT[dname])
                x = torch.zeros((len(rows), k), dtype=DT[dname])
                x[torch.arange(len(rows)), rows] = 1.0
                _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
                got = mixed_precision_gemm(x.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(),
                                           wt.size_bits, wt.bias, 128)
                np.testing.assert_array_equal(to_bits(got), to_bits(w_ref[rows]), err_msg=f"{dname} {wname} zp={use_zp}")


# ---------------------------------------------------------------------------------------------
# kernel modes conch.ops never selects, through the launchers with hand-set metadata
# ---------------------------------------------------------------------------------------------
import dataclasses  # noqa: E402

from conch_amd.kernels.quantization.gemm import (  # noqa: E402
    ChannelScaleMode,
    WeightGroupMode,
    mixed_precision_gemm_launcher,
    scaled_gemm_launcher,
)
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata, create_scaled_metadata  # noqa: E402
from tests.test_oracle_golden import MODE_CASES, MODE_TAGS, mode_inputs  # noqa: E402

GMODE = {oracle.GROUP_NONE: WeightGroupMode.NONE, oracle.GROUP_SHIFT: WeightGroupMode.SHIFT,
         oracle.GROUP_SYMMETRIC_NO_SHIFT: WeightGroupMode.SYMMETRIC_NO_SHIFT,
         oracle.GROUP_SYMMETRIC_WITH_SHIFT: WeightGroupMode.SYMMETRIC_WITH_SHIFT, oracle.GROUP_ASYMMETRIC: WeightGroupMode.ASYMMETRIC}
CMODE = {oracle.CHANNEL_NONE: ChannelScaleMode.NONE, oracle.CHANNEL_WEIGHT_ONLY: ChannelScaleMode.WEIGHT_ONLY,
         oracle.CHANNEL_ACTIVATION_ONLY: ChannelScaleMode.ACTIVATION_ONLY,
         oracle.CHANNEL_WEIGHT_AND_ACTIVATION: ChannelScaleMode.WEIGHT_AND_ACTIVATION}


def run_mixed_mode(x, packed, scales, zeros, nbits, bias, group, gmode, cmode, out_dtype=None):
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    xd, pd, sd, zd = dev(x), dev(packed), dev(scales), dev(zeros)
    md = create_mixed_precision_metadata(xd, pd, sd, zd, nbits, bias, group, output_dtype=out_dtype)
    md = dataclasses.replace(md, weight_group_mode=GMODE[gmode], channel_scale_mode=CMODE[cmode],
                             zero_is_scalar=zeros is not None and zeros.numel() == 1)
    out = torch.empty((x.shape[0], packed.shape[1]), dtype=md.output_dtype, device="cuda")
    mixed_precision_gemm_launcher(out, xd, pd, sd, zd, md)
    return out.cpu()


@pytest.mark.parametrize("case", list(MODE_CASES))
@pytest.mark.parametrize(("nbits", "bias"), MODE_TAGS)
def test_kernel_modes_match_reference_triton_kernel_golden(golden, nbits, bias, case):
    """Golden = the reference's own Triton kernel (CPU interpreter).  Identity activations: bit for bit."""
    g = golden("gemm_modes")
    tag, gmode, cmode, packed, ws, cs, zeros = mode_inputs(g, nbits, bias, case)
    scales = cs if cmode == oracle.CHANNEL_WEIGHT_ONLY else ws
    got = run_mixed_mode(torch.eye(128, dtype=torch.float16), packed, scales, zeros, nbits, bias, 64, gmode, cmode)
    np.testing.assert_array_equal(to_bits(got), g[f"c_{tag}_{case}_eye"])
    x = from_bits(g["x_rand"], torch.float16)
    got = run_mixed_mode(x, packed, scales, zeros, nbits, bias, 64, gmode, cmode).float()
    want = from_bits(g[f"c_{tag}_{case}_rand"], torch.float16).float()
    assert (got - want).abs().max().item() <= 2.0**-10 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize("dname", ["f16", "bf16"])
@pytest.mark.parametrize("case", list(MODE_CASES))
@pytest.mark.parametrize(("nbits", "bias"), [(2, 2), (4, 8), (8, 128), (8, 0), (1, 0), (16, 0)])
def test_kernel_modes_tiled_path_dequantises_bit_exactly(dname, nbits, bias, case):
    """The dequantise-first + 16-bit MFMA path (K % 64 == 0, large enough) against the oracle's restatement, both dtypes:
    identity activations expose the dequantised weights (x channel scale) bit for bit; random activations to accumulation
    order.  1- and 16-bit weights exercise the widths the reference's docstring promises (gemm.py:4)."""
    dt = DT[dname]
    gmode, cmode, _, zname = MODE_CASES[case]
    k, n, group = 512, 320, 128
    seed_everything(nbits * 100 + bias)
    hi = 2**nbits
    # 16-bit weights: small values, so that neither the weights (65535 -> inf in fp16) nor the K = 512 sums overflow fp16
    wq = torch.randint(0, min(hi, 64), (k, n), dtype=torch.int64).to(torch.int32) if nbits == 16 else \
        torch.randint(0, hi, (k, n), dtype=torch.int64).to(torch.int32)
    packed = torch.from_numpy(oracle.pack_rows_ref(wq.numpy(), nbits))
    ws = (torch.rand(k // group, n) * 0.37 + 0.01).to(dt)
    cs = (torch.rand(1, n) * 1.5 + 0.25).to(dt)
    zeros = {None: None, "zi": torch.randint(0, min(hi, 256), (k // group, n), dtype=torch.int32),
             "zf": (torch.rand(k // group, n) * 5 - 2.5).to(dt), "scalar": torch.tensor([3], dtype=torch.int32)}[zname]
    w = oracle.dequantize_modes_ref(packed, ws, zeros, nbits, bias, group, gmode, dt)
    scales = cs if cmode == oracle.CHANNEL_WEIGHT_ONLY else ws
    eye = torch.eye(k, dtype=dt)
    got = run_mixed_mode(eye, packed, scales, zeros, nbits, bias, group, gmode, cmode)
    want = oracle.mixed_precision_gemm_modes_ref(eye, w, cs, cmode, dt)
    np.testing.assert_array_equal(to_bits(got), to_bits(want))
    x = (2 * torch.rand(300, k) - 1).to(dt)
    got = run_mixed_mode(x, packed, scales, zeros, nbits, bias, group, gmode, cmode).float()
    want = oracle.mixed_precision_gemm_modes_ref(x, w, cs, cmode, dt).float()
    tol = 2.0 * EPS[dt] * max(want.abs().max().item(), 1e-6)
    assert (got - want).abs().max().item() <= tol


@pytest.mark.parametrize("oname", ["f16", "bf16"])
@pytest.mark.parametrize("cname", ["none", "wonly", "aonly", "wa"])
@pytest.mark.parametrize(("m", "k", "n"), [(128, 128, 64), (512, 1024, 768), (100, 1024, 260)])
def test_scaled_launcher_channel_modes(oname, cname, m, k, n):
    """ChannelScaleMode set by hand on the scaled launcher's metadata (kernels/quantization/gemm.py:408-440): int8, bit-exact."""
    cmode = {"none": oracle.CHANNEL_NONE, "wonly": oracle.CHANNEL_WEIGHT_ONLY, "aonly": oracle.CHANNEL_ACTIVATION_ONLY,
             "wa": oracle.CHANNEL_WEIGHT_AND_ACTIVATION}[cname]
    a, b, sa, sb, _ = make_scaled_inputs(m, k, n, torch.int8, DT[oname], False, False, False)
    ad, bd, sad, sbd = a.cuda(), b.T.contiguous().cuda().T, sa.cuda(), sb.cuda()
    md = dataclasses.replace(create_scaled_metadata(ad, bd, sad, sbd, DT[oname]), channel_scale_mode=CMODE[cmode])
    out = torch.empty((m, n), dtype=DT[oname], device="cuda")
    scaled_gemm_launcher(out, ad, bd, sad, sbd, md)
    want = oracle.scaled_gemm_modes_ref(a, b, sa, sb, DT[oname], cmode)
    np.testing.assert_array_equal(to_bits(out), to_bits(want))


def test_scaled_launcher_channel_modes_golden(golden):
    """The same against the reference's Triton kernel (fp16 outputs; NONE / WEIGHT_ONLY / ACTIVATION_ONLY bit for bit)."""
    g = golden("gemm_modes")
    a = torch.from_numpy(g["sc_a"].copy()).cuda()
    b = torch.from_numpy(g["sc_bt"].copy()).cuda().T
    sa, sb = torch.from_numpy(g["sc_sa"].copy()).cuda(), torch.from_numpy(g["sc_sb"].copy()).cuda()
    for cname, cmode in (("none", ChannelScaleMode.NONE), ("wonly", ChannelScaleMode.WEIGHT_ONLY), ("aonly", ChannelScaleMode.ACTIVATION_ONLY)):
        md = dataclasses.replace(create_scaled_metadata(a, b, sa, sb, torch.float16), channel_scale_mode=cmode)
        out = torch.empty((128, 64), dtype=torch.float16, device="cuda")
        scaled_gemm_launcher(out, a, b, sa, sb, md)
        np.testing.assert_array_equal(to_bits(out), g[f"sc_c_f16_{cname}"])


def test_nsharded_gemm_world1_equals_scaled_gemm():
    """conch_amd.distributed on the GPU with one rank: the GEMM writes the row-major result directly (no gather, no copy) and
    must equal the plain op bit for bit; gathered_blocks() is a view of the same buffer."""
    from conch_amd.distributed import NShardedScaledGemm

    m, k, n = 2048, 1024, 3584
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, torch.float8_e4m3fn, torch.bfloat16, False, False, True)
    ad, bd, sad, sbd, biasd = a.cuda(), b.T.contiguous().cuda().T, sa.cuda(), sb.cuda(), bias.cuda()
    want = scaled_gemm(ad, bd, sad, sbd, torch.bfloat16, biasd)
    for panels in (1, 2, 4):
        op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cuda"), panels=panels)
        got = op(ad, bd, sad, sbd, biasd)
        np.testing.assert_array_equal(to_bits(got), to_bits(want))
        pan = op.gathered_panels(ad, bd, sad, sbd, biasd)
        assert pan.shape == (panels, 1, m // panels, n)
        np.testing.assert_array_equal(to_bits(pan[panels - 1, 0]), to_bits(want[m - m // panels:]))
        blocks = op.gathered_blocks(ad, bd, sad, sbd, biasd)
        assert blocks.shape == (1, m, n)
        np.testing.assert_array_equal(to_bits(blocks[0]), to_bits(want))


# ---------------------------------------------------------------------------------------------
# the GeGLU pairs (second activation): gelu_tanh_and_mul fused behind both GEMMs
# ---------------------------------------------------------------------------------------------
def check_gelu(got, ref, gemm_out, out_dtype, exact_gemm):
    """gelu_tanh_and_mul against the oracle (torch's F.gelu(approximate="tanh")).

    torch evaluates 0.5 x (1 + tanh z) in fp32: for gates below about -3 the sum 1 + tanh z cancels and its result is a
    multiple of 2^-24 -- the gate value carries an ABSOLUTE error of up to |x| 2^-24, far more than an fp16 / bf16 ulp of the
    tiny result.  The device evaluates the same function as x sigmoid(2z) (no cancellation), so on those elements the two
    differ by torch's own evaluation error, amplified by |up|.  Bound per element, with g, u = gate, up of the (exact) GEMM:
        |dy| <= |u| (|g| 2^-22 + 2 ulp(gelu(g))) + 2 ulp(y);   elsewhere the silu pair's bar (two output ulps)."""
    if not exact_gemm:
        check_silu(got, ref, out_dtype, exact_gemm=False)
        return
    gv, r = got.float().cpu(), ref.float()
    d = gemm_out.shape[1] // 2
    gate, up = gemm_out[:, :d].float(), gemm_out[:, d:].float()
    fin = torch.isfinite(r) & torch.isfinite(gate) & torch.isfinite(up)
    assert torch.equal(torch.isnan(gv), torch.isnan(r))

    def ulp(t):
        return torch.maximum(torch.ldexp(torch.ones_like(t), torch.frexp(t)[1] - 1 - MANT[out_dtype]), torch.full_like(t, 2.0**-24))

    s = torch.nn.functional.gelu(gate, approximate="tanh")
    bound = up.abs() * (gate.abs() * 2.0**-22 + 2 * ulp(s)) + 2 * ulp(r)
    assert ((gv - r).abs()[fin] <= bound[fin]).all(), f"worst excess {((gv - r).abs() - bound)[fin].max().item():.3g}"
    # gates between about -6 and -2 sit in torch's cancellation regime: its value there is off by up to 2^-12 relative, which
    # moves a bf16 rounding a few per cent of the time (measured 5.4 % of ALL elements on the reference's own recipe)
    assert (gv[fin] != r[fin]).float().mean().item() <= 1e-1


def run_gelu(a, b, sa, sb, out_dtype, bias):
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    b_dev = b.T.contiguous().cuda().T if b.stride(0) == 1 else b.cuda()
    return scaled_gemm_gelu_tanh_and_mul(dev(a), b_dev, dev(sa), dev(sb), out_dtype, dev(bias))


@pytest.mark.parametrize("iname", list(IN_T))
@pytest.mark.parametrize("oname", ["f16", "bf16"])
@pytest.mark.parametrize("key_tail", ["sa1_sb1_b1", "sa0_sb0_b1", "sa0_sb0_b0", "sa1_sb0_b0"])
def test_scaled_gemm_gelu_golden_from_reference(golden, iname, oname, key_tail):
    """Expected = the reference's gelu_tanh_and_mul of the reference's scaled_gemm (tests/golden/scaled_gemm_gelu.npz); same
    bars as the silu pair (hardware exp / rcp against torch's tanh: two output ulps, <= 0.2 % of the elements off)."""
    g, y = golden("scaled_gemm"), golden("scaled_gemm_gelu")
    key = f"{iname}_{oname}_{key_tail}"
    a = from_bits(g[f"a_{key}"], IN_T[iname])
    b = from_bits(g[f"bt_{key}"], IN_T[iname]).T
    sa, sb = torch.from_numpy(g[f"sa_{key}"]), torch.from_numpy(g[f"sb_{key}"])
    bias = from_bits(g[f"bias_{key}"], DT[oname]) if key_tail.endswith("b1") else None
    got = run_gelu(a, b, sa, sb, DT[oname], bias)
    assert got.shape == (128, 64) and got.dtype == DT[oname]
    check_gelu(got, from_bits(y[f"y_{key}"], DT[oname]), from_bits(g[f"c_{key}"], DT[oname]).reshape(128, 128), DT[oname],
               exact_gemm=iname == "int8")


@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize(("m", "k", "d"), [(1024, 1024, 512), (300, 384, 260), (257, 256, 8), (64, 2048, 96), (16, 1024, 1376)])
def test_scaled_gemm_gelu_shapes_fused_equals_unfused(iname, m, k, d):
    a, b, sa, sb, bias = make_scaled_inputs(m, k, 2 * d, IN_T[iname], torch.bfloat16, False, False, True)
    ref = oracle.scaled_gemm_gelu_tanh_and_mul_ref(a, b, sa, sb, torch.bfloat16, bias)
    got = run_gelu(a, b, sa, sb, torch.bfloat16, bias)
    check_gelu(got, ref, oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias), torch.bfloat16, exact_gemm=iname == "int8")
    _C.set_gemm_variant(_C.VARIANT_MFMA_SIMPLE)
    unfused = run_gelu(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    if m <= 256 and iname != "int8":
        tol = 4.0 * EPS[torch.bfloat16] * unfused.float().abs().max().item()
        assert (got.float() - unfused.float()).abs().max().item() <= tol
    else:
        assert torch.equal(got, unfused)
    # and it is not the silu pair
    assert not torch.equal(got, run_silu(a, b, sa, sb, torch.bfloat16, bias))


@pytest.mark.parametrize(("m", "k", "d"), [(300, 256, 260), (1024, 512, 688), (16, 1024, 260)])
@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"), ("uint8", False, "bf16")])
def test_mixed_precision_gelu_fused(m, k, d, wname, use_zp, dname):
    """int4 / int8-weight GeGLU pair: against the exact pair in fp64 (6 eps of max|y|: the GEMM's accumulation-order tolerance
    through the gate) and bit for bit against the library's own unfused pair."""
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, 2 * d, wt, use_zp, DT[dname], 128)
    a = (a.float() * 0.02).to(DT[dname])
    args = (a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), wt.size_bits, wt.bias, 128)
    got = mixed_precision_gemm_gelu_tanh_and_mul(*args)
    exact_g = a.double() @ w_ref.double()
    exact = (torch.nn.functional.gelu(exact_g[:, :d], approximate="tanh") * exact_g[:, d:]).float()
    ok = exact.abs() < 3.0e4
    assert (got.float().cpu()[ok] - exact[ok]).abs().max().item() <= 6.0 * EPS[DT[dname]] * exact[ok].abs().max().item()
    _C.set_gemm_variant(_C.VARIANT_MFMA_SIMPLE)
    try:
        unfused = mixed_precision_gemm_gelu_tanh_and_mul(*args)
    finally:
        _C.set_gemm_variant(_C.VARIANT_AUTO)
    if m <= 256:
        assert (got.float() - unfused.float()).abs().max().item() <= 4.0 * EPS[DT[dname]] * unfused.float().abs().max().item()
    else:
        assert torch.equal(got, unfused)


# ---------------------------------------------------------------------------------------------
# weight pre-pack
# ---------------------------------------------------------------------------------------------
from conch_amd.ops.quantization.prepack import (  # noqa: E402
    mixed_precision_gemm_prepacked,
    prepack_mixed_weights,
    unpack_mixed_weights,
)


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("tile_nt", [2, 3, 4])
@pytest.mark.parametrize(("k", "n"), [(64, 64), (256, 1376), (512, 200), (4096, 11008), (128, 1)])
def test_prepack_round_trip(bits, tile_nt, k, n):
    """unpack(prepack(w)) == w for every width / tile width, ragged N included (clamped columns of the last tile)."""
    w = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
    pre = prepack_mixed_weights(w, bits, tile_nt=tile_nt)
    assert pre.image.numel() == -(-n // (64 * tile_nt)) * (k // 64) * 512 * tile_nt * (1 if bits == 4 else 2)
    assert torch.equal(unpack_mixed_weights(pre), w)
    assert not torch.equal(pre.image[: w.numel()].view_as(w), w) or k * n <= 64  # it IS a different layout


@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"), ("uint8b128", False, "bf16"),
                                                         ("uint8", True, "f16"), ("uint4b8", False, "bf16")])
@pytest.mark.parametrize(("m", "k", "n"), [(1024, 512, 1376), (300, 256, 520), (1100, 512, 1024), (257, 128, 200)])
@pytest.mark.parametrize("tile_nt", [None, 2, 3])
def test_prepacked_gemm_is_bit_identical(wname, use_zp, dname, m, k, n, tile_nt):
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    plain = mixed_precision_gemm(dev(a), dev(packed), dev(w_s), dev(w_zp), wt.size_bits, wt.bias, 128)
    pre = prepack_mixed_weights(dev(packed), wt.size_bits, m_hint=m, per_group_zero_points=use_zp, tile_nt=tile_nt)
    got = mixed_precision_gemm_prepacked(dev(a), pre, dev(w_s), dev(w_zp), wt.bias, 128)
    assert torch.equal(got, plain)
    check_mixed(got, a, w_ref, k)


@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"), ("uint8b128", False, "bf16"),
                                                         ("uint8", True, "f16")])
@pytest.mark.parametrize(("m", "k", "n"), [(1100, 512, 1024), (1300, 256, 520), (512, 128, 128)])
def test_tall_tile_is_bit_identical(wname, use_zp, dname, m, k, n):
    """The 512 x 128 tile (half the dequantisation per MFMA) against the dispatcher's choice: the tile shape moves work, not the
    order in which an output accumulates its K products.  Plain weights and the 128-column pre-packed image."""
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 4 if not use_zp else 3)
    try:
        base = mixed_precision_gemm(dev(a), dev(packed), dev(w_s), dev(w_zp), wt.size_bits, wt.bias, 128)
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 5)
        tall = mixed_precision_gemm(dev(a), dev(packed), dev(w_s), dev(w_zp), wt.size_bits, wt.bias, 128)
        pre = prepack_mixed_weights(dev(packed), wt.size_bits, tile_nt=2)
        tall_pre = mixed_precision_gemm_prepacked(dev(a), pre, dev(w_s), dev(w_zp), wt.bias, 128)
    finally:
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
    assert torch.equal(tall, base)
    assert torch.equal(tall_pre, base)
    check_mixed(tall, a, w_ref, k)


def test_prepacked_gemm_c4_config():
    """BASELINE config C4 on a pre-packed weight: the same bits as the plain op at full size."""
    m, k, n = 1024, 4096, 11008
    wt = scalar_types.uint4b8
    a, w_ref, packed, w_s, _ = make_mixed_inputs(m, k, n, wt, False, torch.float16)
    plain = mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.cuda(), None, wt.size_bits, wt.bias, 128)
    pre = prepack_mixed_weights(packed.cuda(), wt.size_bits, m_hint=m)
    assert pre.tile_nt == 3  # 232 tiles of 192 columns: one round of the chip
    assert torch.equal(mixed_precision_gemm_prepacked(a.cuda(), pre, w_s.cuda(), None, wt.bias, 128), plain)


# ---------------------------------------------------------------------------------------------
# K that is not a multiple of the split-K slice: the Llama-7B down projection (K = 11008 = 10 slices + 768) at decode sizes
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize("m", [1, 16, 128, 200])
def test_scaled_gemm_down_projection_decode(iname, m):
    k, n = 11008, 4096
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T[iname], torch.bfloat16, False, False, True)
    ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
    got = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
    check_scaled(got, ref, IN_T[iname], torch.bfloat16, (a, b, sa, sb, bias))
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    auto = run_scaled(a, b, sa, sb, torch.bfloat16, bias)
    check_scaled(auto, ref, IN_T[iname], torch.bfloat16, (a, b, sa, sb, bias))


def test_scaled_gemm_ragged_slice_ignores_what_lies_behind_k():
    """The steps past K of the last slice must contribute nothing even if the memory behind a row holds NaN patterns: A and B are
    views of wider arrays whose tail columns are 0x7f (e4m3fn NaN)."""
    m, k, n, pad = 24, 1152, 64, 1024
    a_wide = torch.full((m, k + pad), 0x7F, dtype=torch.uint8).view(torch.float8_e4m3fn)
    b_wide = torch.full((n, k + pad), 0x7F, dtype=torch.uint8).view(torch.float8_e4m3fn)
    a0 = (0.25 * torch.rand((m, k))).to(torch.float8_e4m3fn)
    b0 = (0.25 * torch.rand((n, k))).to(torch.float8_e4m3fn)
    a_wide.view(torch.uint8)[:, :k] = a0.view(torch.uint8)
    b_wide.view(torch.uint8)[:, :k] = b0.view(torch.uint8)
    sa, sb = 0.25 * torch.rand((m, 1)), 0.25 * torch.rand((n, 1))
    ref = oracle.scaled_gemm_ref(a0, b0.T, sa, sb, torch.bfloat16, None)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
    got = scaled_gemm(a_wide.cuda()[:, :k], b_wide.cuda()[:, :k].T, sa.cuda(), sb.cuda(), torch.bfloat16)
    assert torch.isfinite(got.float()).all()
    check_scaled(got, ref, torch.float8_e4m3fn, torch.bfloat16, (a0, b0.T, sa, sb, None))


@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"), ("uint8b128", False, "f16")])
@pytest.mark.parametrize("m", [1, 16, 64])
def test_mixed_precision_down_projection_decode(m, wname, use_zp, dname):
    k, n = 11008, 4096
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    args = (a.cuda(), packed.cuda(), w_s.cuda(), None if w_zp is None else w_zp.cuda(), wt.size_bits, wt.bias, 128)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
    got = mixed_precision_gemm(*args)
    check_mixed(got, a, w_ref, k)
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    auto = mixed_precision_gemm(*args)
    if m <= 32:
        assert torch.equal(auto, got)  # auto = the decode-batch kernel
    else:
        check_mixed(auto, a, w_ref, k)  # from 33 rows the K-split strip forms may take it (other K slices: other fp32 sums)


def test_mixed_precision_ragged_slice_ignores_inf_behind_k():
    """X is a view of a wider array whose tail columns are +inf: the steps past K of the last slice must not read them."""
    m, k, n, pad = 8, 1152, 128, 1024
    wt = scalar_types.uint4b8
    a, w_ref, packed, w_s, _ = make_mixed_inputs(m, k, n, wt, False, torch.float16)
    wide = torch.full((m, k + pad), float("inf"), dtype=torch.float16)
    wide[:, :k] = a
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
    got = mixed_precision_gemm(wide.cuda()[:, :k], packed.cuda(), w_s.cuda(), None, wt.size_bits, wt.bias, 128)
    assert torch.isfinite(got.float()).all()
    check_mixed(got, a, w_ref, k)


# ---------------------------------------------------------------------------------------------
# split-K form of the LDS-tiled mixed kernel (few tiles: M of a few hundred rows, N of a few thousand columns)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("split", [2, 4, 8])
@pytest.mark.parametrize(("m", "k", "n"), [(512, 2048, 1024), (300, 2048, 520), (1024, 1152, 256), (700, 2048, 1024), (256, 2048, 1024), (100, 4096, 520),
                                            (192, 2176, 256)])
@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"), ("uint8b128", False, "bf16")])
def test_mixed_precision_split_k(_reset_tuning, split, m, k, n, wname, use_zp, dname):
    """K slices per tile, fp32 slabs, reduce in slice order: against the oracle, against the unsplit kernel (another summation
    order: 2 eps of max|C|), plain and pre-packed weights; K whose slices are not equal (1152 = 18 steps) included."""
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    args = (dev(a), dev(packed), dev(w_s), dev(w_zp), wt.size_bits, wt.bias, 128)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)  # the LDS-tiled kernel, not the decode kernel
    try:
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)
        whole = mixed_precision_gemm(*args)
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, split)
        got = mixed_precision_gemm(*args)
        pre = prepack_mixed_weights(dev(packed), wt.size_bits, m_hint=m, per_group_zero_points=use_zp)
        got_pre = mixed_precision_gemm_prepacked(dev(a), pre, dev(w_s), dev(w_zp), wt.bias, 128)
        again = mixed_precision_gemm(*args)
    finally:
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)
        _C.set_gemm_variant(_C.VARIANT_AUTO)
    check_mixed(got, a, w_ref, k)
    tol = 2.0 * EPS[DT[dname]] * whole.float().abs().max().item()
    assert (got.float() - whole.float()).abs().max().item() <= tol
    assert torch.equal(got_pre, got) and torch.equal(again, got)  # same slices, same order: deterministic


# ---------------------------------------------------------------------------------------------
# static activation quantiser fused into the GEMM (second half)
# ---------------------------------------------------------------------------------------------
from conch_amd.ops.quantization.fp8 import scaled_fp8_quant  # noqa: E402
from conch_amd.ops.quantization.gemm import static_quant_scaled_gemm  # noqa: E402
from conch_amd.ops.quantization.int8 import scaled_int8_quant  # noqa: E402


@pytest.mark.parametrize("iname", ["int8", "fn"])
@pytest.mark.parametrize("xname", ["f16", "bf16"])
@pytest.mark.parametrize(("m", "k", "n"), [(1, 1024, 64), (16, 4096, 520), (33, 1152, 1376), (64, 2048, 256), (100, 11008, 512), (128, 4096, 4096),
                                            (200, 1024, 260), (256, 384, 128), (512, 1024, 768), (24, 1000, 72), (48, 1024, 66)])
def test_static_quant_scaled_gemm_equals_the_pair(_reset_tuning, iname, xname, m, k, n):
    """One call == static quantiser + scaled_gemm with scale_a = the quantiser's scale, bit for bit: decode sizes (the split-K
    kernel quantises A on its way into LDS; ragged K slices, ragged M / N), sizes the tile kernels take (unfused pair inside),
    and layouts the fused path cannot take (K not a multiple of 128, N not a multiple of 4)."""
    seed_everything(21)
    in_dt = IN_T[iname]
    x = (4.0 * (torch.rand((m, k)) - 0.5)).to(DT[xname])
    x[0, 0] = 1000.0  # saturates both formats
    if in_dt == torch.int8:
        bt = torch.randint(-32, 32, (n, k), dtype=torch.int8)
        sx = torch.tensor([0.02], dtype=torch.float32)
        quant = scaled_int8_quant
    else:
        bt = (0.25 * torch.rand((n, k))).to(in_dt)
        sx = torch.tensor([0.01], dtype=torch.float32)
        quant = scaled_fp8_quant
    sb = 0.25 * torch.rand((n, 1))
    bias = torch.rand((n,)).to(torch.bfloat16)
    xd, bd, sxd, sbd, biasd = x.cuda(), bt.cuda().T, sx.cuda(), sb.cuda(), bias.cuda()
    got = static_quant_scaled_gemm(xd, bd, sxd, sbd, torch.bfloat16, biasd)
    xq, _ = quant(xd, sxd)
    # the pair on the kernel the fused call used: split-K two-launch form where it can run, the dispatcher's choice elsewhere
    fused_layout = m <= 256 and n % 4 == 0 and k % 128 == 0
    if fused_layout and in_dt != torch.int8:
        _C.set_tuning(_C.TUNE_SKINNY_MODE, 1)
    pair = scaled_gemm(xq, bd, sxd.reshape(1, 1), sbd, torch.bfloat16, biasd)
    if in_dt == torch.int8 or not fused_layout:
        assert torch.equal(got, pair)
    else:
        # the dispatcher may have run either side on another kernel (fp32 summation order): equal when both took the split-K form
        _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)
        forced_pair = scaled_gemm(xq, bd, sxd.reshape(1, 1), sbd, torch.bfloat16, biasd)
        forced = static_quant_scaled_gemm(xd, bd, sxd, sbd, torch.bfloat16, biasd)
        _C.set_gemm_variant(_C.VARIANT_AUTO)
        assert torch.equal(forced, forced_pair)
        tol = 4.0 * EPS[torch.bfloat16] * pair.float().abs().max().item()
        assert (got.float() - pair.float()).abs().max().item() <= tol
    ref = oracle.scaled_gemm_ref(xq.cpu(), bt.T, sx.reshape(1, 1), sb, torch.bfloat16, bias)
    check_scaled(got, ref, in_dt, torch.bfloat16)


@pytest.mark.parametrize(("m", "k", "n"), [(1, 4096, 4096), (8, 1024, 64), (16, 2048, 520), (20, 11008, 512), (32, 1152, 256), (3, 8192, 1376)])
@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16"), ("uint8b128", False, "bf16")])
def test_mixed_decode_one_launch_is_bit_identical_to_two_launches(_reset_tuning, m, k, n, wname, use_zp, dname):
    """GEMV sizes: the reduce of the K slices inside the decode kernel (last-arriving workgroup of a tile, slice order) against
    the separate reduce kernel; repeated launches (the counters reset themselves)."""
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    args = (dev(a), dev(packed), dev(w_s), dev(w_zp), wt.size_bits, wt.bias, 128)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, 1)
    two = mixed_precision_gemm(*args)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, 2)
    for _ in range(4):
        assert torch.equal(mixed_precision_gemm(*args), two)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, 0)
    assert torch.equal(mixed_precision_gemm(*args), two)
    check_mixed(two, a, w_ref, k)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize(("m", "k", "n"), [(1, 4096, 4096), (8, 1024, 64), (16, 2048, 520), (20, 11008, 512), (32, 1152, 256), (48, 2048, 1376), (3, 8192, 68)])
@pytest.mark.parametrize(("wname", "use_zp", "dname"), [("uint4b8", False, "f16"), ("uint4", True, "bf16")])
def test_mixed_decode_word_fetch_orders_are_bit_identical(_reset_tuning, mode, m, k, n, wname, use_zp, dname):
    """int4 decode kernel: the packed words in MFMA operand order into registers (CONCH_TUNE_SKINNY_GATHER = 1) against 16 bytes
    per lane through LDS-DMA (= 2; the automatic choice while the launch is at most one workgroup per CU): the same words in the
    same operand registers, in both launch forms, ragged N / K and a column block past N included."""
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname])
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    args = (dev(a), dev(packed), dev(w_s), dev(w_zp), wt.size_bits, wt.bias, 128)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, mode)
    _C.set_tuning(_C.TUNE_SKINNY_GATHER, 1)
    regs = mixed_precision_gemm(*args)
    _C.set_tuning(_C.TUNE_SKINNY_GATHER, 2)
    for _ in range(2):
        assert torch.equal(mixed_precision_gemm(*args), regs)
    check_mixed(regs, a, w_ref, k)


@pytest.mark.parametrize(("m", "k", "n"), [(16, 4096, 11008), (32, 8192, 8192), (8, 4096, 28672)])
def test_mixed_decode_one_launch_with_several_workgroups_per_cu(_reset_tuning, m, k, n):
    """688 / 1024 / 1792 workgroups of the one-launch decode kernel on 256 CUs (two to four resident per CU, several rounds):
    the write-through / ticket / sc1-load hand-off must not depend on one workgroup per CU.  40 launches, a second stream
    keeping some CUs busy with an unrelated GEMM half of the time, every result against the two-launch form bit for bit."""
    wt = scalar_types.uint4b8
    a, w_ref, packed, w_s, _ = make_mixed_inputs(m, k, n, wt, False, torch.float16)
    args = (a.cuda(), packed.cuda(), w_s.cuda(), None, wt.size_bits, wt.bias, 128)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, 1)
    two = mixed_precision_gemm(*args)
    _C.set_tuning(_C.TUNE_SKINNY_MODE, 2)
    side = torch.cuda.Stream()
    big = torch.randn((2048, 2048), device="cuda", dtype=torch.float16)
    outs = []
    for i in range(40):
        if i % 2:
            with torch.cuda.stream(side):
                torch.matmul(big, big)
        outs.append(mixed_precision_gemm(*args))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, two), f"launch {i}"
    check_mixed(two, a, w_ref, k)


@pytest.mark.parametrize("nt", [2, 3, 4, 5])
@pytest.mark.parametrize(("wname", "use_zp", "dname", "group"), [("uint4b8", False, "f16", 128), ("uint4", True, "bf16", 64)])
def test_mixed_precision_split_k_on_every_tile_shape(_reset_tuning, nt, wname, use_zp, dname, group):
    """The split-K form on each tile shape (the 512 x 128 tile included) and with 64-element weight groups (one step per group)."""
    if nt == 4 and use_zp:
        pytest.skip("256-column tiles are not built for per-group zero points")
    m, k, n = 600, 1536, 520
    wt = WTYPES[wname]
    a, w_ref, packed, w_s, w_zp = make_mixed_inputs(m, k, n, wt, use_zp, DT[dname], group)
    dev = lambda t: None if t is None else t.cuda()  # noqa: E731
    args = (dev(a), dev(packed), dev(w_s), dev(w_zp), wt.size_bits, wt.bias, group)
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
    try:
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)
        whole = mixed_precision_gemm(*args)
        outs = []
        for split in (2, 4, 8):
            _C.set_tuning(_C.TUNE_MIXED_SPLITK, split)
            outs.append(mixed_precision_gemm(*args))
    finally:
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
        _C.set_gemm_variant(_C.VARIANT_AUTO)
    tol = 2.0 * EPS[DT[dname]] * whole.float().abs().max().item()
    for got in outs:
        check_mixed(got, a, w_ref, k)
        assert (got.float() - whole.float()).abs().max().item() <= tol


def test_nsharded_gemm_exchange_path_through_a_one_rank_rccl_group():
    """The exchange path of conch_amd.distributed -- GEMM into the gather slot, all_gather_into_tensor on a side stream under the
    next panel's GEMM, unpack to row-major -- on the real backend (RCCL), as far as a one-GPU box allows: a one-rank process
    group.  The result must equal scaled_gemm bit for bit, in every layout the class hands out."""
    import os
    import socket

    import torch.distributed as dist

    from conch_amd.distributed import NShardedScaledGemm

    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        m, k, n = 2048, 1024, 3584
        a, b, sa, sb, bias = make_scaled_inputs(m, k, n, torch.float8_e4m3fn, torch.bfloat16, False, False, True)
        ad, bd, sad, sbd, biasd = a.cuda(), b.T.contiguous().cuda().T, sa.cuda(), sb.cuda(), bias.cuda()
        want = scaled_gemm(ad, bd, sad, sbd, torch.bfloat16, biasd)
        for panels in (1, 2, 4):
            op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cuda"), panels=panels, force_collective=True)
            for _ in range(3):  # buffers and the side stream are reused across calls
                got = op(ad, bd, sad, sbd, biasd)
                torch.cuda.synchronize()
                np.testing.assert_array_equal(to_bits(got), to_bits(want))
            pan = op.gathered_panels(ad, bd, sad, sbd, biasd)
            torch.cuda.synchronize()
            assert pan.shape == (panels, 1, m // panels, n)
            np.testing.assert_array_equal(to_bits(pan[panels - 1, 0]), to_bits(want[m - m // panels:]))
            blocks = op.gathered_blocks(ad, bd, sad, sbd, biasd)
            torch.cuda.synchronize()
            assert blocks.shape == (1, m, n)
            np.testing.assert_array_equal(to_bits(blocks[0]), to_bits(want))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("col_off", [8, 64, 200])
@pytest.mark.parametrize("variant_name", ["VARIANT_MFMA_PINGPONG2", "VARIANT_MFMA_MID", "VARIANT_MFMA_PINGPONG"])
def test_whole_line_epilogue_into_a_strided_column_block(_reset_tuning, col_off, variant_name):
    """An N-shard writes its column block into a wider row-major buffer: the output is a view with a column offset (16-byte but
    not 128-byte aligned rows for offsets 8 and 200) and a row stride larger than N.  The whole-line epilogue addresses it
    through a buffer descriptor over the view; it must write exactly the block and nothing around it."""
    m, k, n, wide = 512, 256, 768, 1280
    a, b, sa, sb, bias = make_scaled_inputs(m, k, n, IN_T["fn"], torch.bfloat16, False, False, True)
    ad, bd, sad, sbd, biasd = a.cuda(), b.T.contiguous().cuda().T, sa.cuda(), sb.cuda(), bias.cuda()
    md = create_scaled_metadata(ad, bd, sad, sbd, torch.bfloat16)
    _C.set_gemm_variant(getattr(_C, variant_name))
    try:
        want = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        _C.set_tuning(_C.TUNE_EPILOGUE, 1)
        scaled_gemm_launcher(want, ad, bd, sad, sbd, md, bias=biasd)
        _C.set_tuning(_C.TUNE_EPILOGUE, 2)
        buf = torch.full((m, wide), -7.0, dtype=torch.bfloat16, device="cuda")
        scaled_gemm_launcher(buf[:, col_off:col_off + n], ad, bd, sad, sbd, md, bias=biasd)
    finally:
        _C.set_gemm_variant(0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(to_bits(buf[:, col_off:col_off + n].contiguous()), to_bits(want))
    assert bool((buf[:, :col_off] == -7.0).all()) and bool((buf[:, col_off + n:] == -7.0).all())


@pytest.mark.parametrize("m", [16, 48, 300])
def test_mixed_gemm_takes_packed_weights_that_are_a_column_slice(_reset_tuning, m):
    """`w_q_packed` as columns [1, 1 + N) of a wider int32 tensor (row stride N + 9 words, base pointer 4 bytes off a 16-byte
    boundary): the decode kernel's 16-byte LDS-DMA pieces do not apply, its dword loads and the tiled kernel do -- the same result
    as from a contiguous copy, bit for bit."""
    wt = scalar_types.uint4b8
    k, n = 2048, 520
    a, w_ref, packed, w_s, _ = make_mixed_inputs(m, k, n, wt, False, torch.float16)
    wide = torch.zeros((packed.shape[0], n + 9), dtype=torch.int32, device="cuda")
    wide[:, 1:1 + n] = packed.cuda()
    view = wide[:, 1:1 + n]
    assert view.stride(0) == n + 9 and view.data_ptr() % 16 == 4
    # (K unsplit in the tile kernels: the aligned copy may otherwise take the strip kernel's K slices -- round 5 -- and the misaligned
    # view the LDS-tiled kernel's own split: the same sums in another grouping)
    _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)
    want = mixed_precision_gemm(a.cuda(), packed.cuda(), w_s.cuda(), None, wt.size_bits, wt.bias, 128)
    got = mixed_precision_gemm(a.cuda(), view, w_s.cuda(), None, wt.size_bits, wt.bias, 128)
    assert torch.equal(got, want)
    check_mixed(want, a, w_ref, k)


def test_randomised_dispatch_cross_check():
    """tools/fuzz_dispatch.py as a test: 600 random (op, dtype, M, K, N, scale shape, bias / zero point) draws through the AUTOMATIC
    kernel choice -- every dispatcher branch: split-K 32 / 64 / 128 rows in all launch forms, 128 x 128 and 256 x 256 tiles, repack,
    decode-batch and tiled mixed kernels -- against the generic device kernel (int8 bit for bit, fp8 and mixed within 2 eps of
    max|C|).  A child process: the tool owns the tuning state it flips."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_dispatch.py"), "600", "3"], capture_output=True, text=True, timeout=200, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "600 cases, 0 mismatches" in r.stdout


def test_randomised_assembly_kernel_cross_check():
    """tools/fuzz_dispatch.py --asm: 150 random draws FORCED onto the one-wave-per-SIMD assembly kernel -- int8 / fp8, bf16 / fp16,
    scalar / vector scales, with and without a bias (round 6's *_bias twins), the three tile widths, ragged M and N -- against the
    generic device kernel: int8 bit for bit, fp8 within 2 eps of max|C|."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_dispatch.py"), "150", "11", "--asm"], capture_output=True, text=True, timeout=200, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "150 cases, 0 mismatches" in r.stdout
