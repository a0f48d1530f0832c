"""The plain-C oracle (oracle/oracle_c.c, no torch involved) against the reference's golden vectors
and against the Python oracle: int8 / fp8 quantisation, int8 scaled GEMM, packed-weight dequantisation."""

from __future__ import annotations

import ctypes

import numpy as np
import pytest
import torch

import oracle
from oracle import build_c
from tests.conftest import DT, from_bits, to_bits

CDT = {"f32": 0, "f16": 1, "bf16": 2}
P = ctypes.c_void_p


@pytest.fixture(scope="module")
def clib():
    lib = ctypes.CDLL(str(build_c.build()))
    lib.oracle_int8_quant.argtypes = [P, P, ctypes.c_int, ctypes.c_float, ctypes.c_int64]
    lib.oracle_fp8_quant.argtypes = [P, P, ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_int]
    lib.oracle_scaled_gemm_int8.argtypes = [P, P, P, P, ctypes.c_int64, P, ctypes.c_int64, P] + [ctypes.c_int64] * 3 + [ctypes.c_int]
    lib.oracle_dequant_packed.argtypes = [P, P, P, P, ctypes.c_int] + [ctypes.c_int64] * 2 + [ctypes.c_int] * 4
    return lib


def ptr(a: np.ndarray):
    return a.ctypes.data_as(P)


GRID = [(d, t, h, s) for d in DT for t in (1, 7, 83) for h in (16, 67, 768) for s in (0.1, 2.1)]


@pytest.mark.parametrize(("dname", "tokens", "hidden", "scale"), GRID)
def test_c_quantisers_match_reference_golden(clib, golden, dname, tokens, hidden, scale):
    key = f"{dname}_t{tokens}_h{hidden}_s{scale}"
    gi, gf = golden("quant_int8"), golden("quant_fp8")
    x = np.ascontiguousarray(gi[f"x_{key}"])
    out = np.empty(x.size, dtype=np.int8)
    clib.oracle_int8_quant(ptr(out), ptr(x), CDT[dname], scale, x.size)
    np.testing.assert_array_equal(out.reshape(x.shape), gi[f"q_{key}"])
    x8 = np.ascontiguousarray(gf[f"x_{key}"])
    for fnuz, tag in ((0, "fn"), (1, "fnuz")):
        out8 = np.empty(x8.size, dtype=np.uint8)
        clib.oracle_fp8_quant(ptr(out8), ptr(x8), CDT[dname], scale, x8.size, fnuz)
        np.testing.assert_array_equal(out8.reshape(x8.shape), gf[f"q{tag}_{key}"])


@pytest.mark.parametrize("dname", list(DT))
def test_c_quantisers_edge_vectors(clib, golden, dname):
    gi, gf = golden("quant_int8"), golden("quant_fp8")
    for sname in ("one", "s2p1"):
        x = np.ascontiguousarray(gi[f"edge_x_{dname}_{sname}"])
        s = float(gi[f"edge_s_{dname}_{sname}"].reshape(-1)[0])
        out = np.empty(x.size, dtype=np.int8)
        clib.oracle_int8_quant(ptr(out), ptr(x), CDT[dname], s, x.size)
        np.testing.assert_array_equal(out.reshape(x.shape), gi[f"edge_q_{dname}_{sname}"])
        for fnuz, tag in ((0, "fn"), (1, "fnuz")):
            out8 = np.empty(x.size, dtype=np.uint8)
            clib.oracle_fp8_quant(ptr(out8), ptr(x), CDT[dname], s, x.size, fnuz)
            np.testing.assert_array_equal(out8.reshape(x.shape), gf[f"edge_q{tag}_{dname}_{sname}"])


@pytest.mark.parametrize("oname", ["f16", "bf16"])
@pytest.mark.parametrize(("sa_s", "sb_s", "use_bias"), [(1, 1, 1), (0, 0, 1), (0, 1, 0), (1, 0, 0)])
def test_c_scaled_gemm_int8_matches_reference_golden(clib, golden, oname, sa_s, sb_s, use_bias):
    g = golden("scaled_gemm")
    key = f"int8_{oname}_sa{sa_s}_sb{sb_s}_b{use_bias}"
    a, bt = np.ascontiguousarray(g[f"a_{key}"]), np.ascontiguousarray(g[f"bt_{key}"])
    sa = np.ascontiguousarray(g[f"sa_{key}"].reshape(-1).astype(np.float32))
    sb = np.ascontiguousarray(g[f"sb_{key}"].reshape(-1).astype(np.float32))
    bias = np.ascontiguousarray(g[f"bias_{key}"]) if use_bias else None
    m, k = a.shape
    n = bt.shape[0]
    c = np.empty((m, n), dtype=np.uint16)
    clib.oracle_scaled_gemm_int8(ptr(c), ptr(a), ptr(bt), ptr(sa), sa.size, ptr(sb), sb.size,
                                 ptr(bias) if bias is not None else None, m, n, k, CDT[oname])
    np.testing.assert_array_equal(c, g[f"c_{key}"].reshape(m, n))


MIXED = [(w, z, d) for w in ("uint4b8", "uint8b128", "uint4", "uint8") for z in (1, 0) for d in ("f16", "bf16")]


@pytest.mark.parametrize(("wname", "zp", "dname"), MIXED)
def test_c_dequant_matches_w_ref(clib, golden, wname, zp, dname):
    g = golden("mixed_gemm")
    key = f"{wname}_zp{zp}_{dname}"
    bits, bias, group = (int(v) for v in g[f"meta_{key}"])
    packed = np.ascontiguousarray(g[f"packed_{key}"])
    ws = np.ascontiguousarray(g[f"ws_{key}"])
    wzp = np.ascontiguousarray(g[f"wzp_{key}"]) if zp else None
    k, n = g[f"wref_{key}"].shape
    w = np.empty((k, n), dtype=np.uint16)
    clib.oracle_dequant_packed(ptr(w), ptr(packed), ptr(ws), ptr(wzp) if wzp is not None else None, 0, k, n, bits, bias,
                               group, CDT[dname])
    np.testing.assert_array_equal(w, g[f"wref_{key}"])


def test_c_float_conversions_agree_with_torch():
    """The hand-written fp16 / bf16 rounding in oracle_c.c against torch on a dense sweep (via the GEMM epilogue)."""
    lib = ctypes.CDLL(str(build_c.build()))
    rng = np.random.default_rng(0)
    vals = (rng.standard_normal(20000) * np.exp(rng.uniform(-20, 12, 20000))).astype(np.float32)
    vals = np.concatenate([vals, np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 6.1e-5], dtype=np.float32)])
    # C[0][j] = out16(sb[j] * (1 * 1)) with a 1x1 int8 product of 1: exercises to16() on arbitrary floats
    a = np.ones((1, 1), dtype=np.int8)
    bt = np.ones((vals.size, 1), dtype=np.int8)
    sa = np.ones(1, dtype=np.float32)
    lib.oracle_scaled_gemm_int8.argtypes = [P, P, P, P, ctypes.c_int64, P, ctypes.c_int64, P] + [ctypes.c_int64] * 3 + [ctypes.c_int]
    for dname, tdt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        c = np.empty((1, vals.size), dtype=np.uint16)
        lib.oracle_scaled_gemm_int8(ptr(c), ptr(a), ptr(bt), ptr(sa), 1, ptr(vals), vals.size, None, 1, vals.size, 1, CDT[dname])
        want = to_bits(torch.from_numpy(vals.copy()).to(tdt))
        np.testing.assert_array_equal(c.reshape(-1), want)
