"""Pin the CPU oracle (oracle/) to the golden vectors produced by the real reference.

CPU-only.  Every comparison is bit-exact: the oracle restates the reference's torch arithmetic
in the same order, so even the floating-point outputs must agree to the last bit.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle
from tests.conftest import DT, from_bits, to_bits

torch.set_num_threads(1)

GRID = [(d, t, h, s) for d in DT for t in (1, 7, 83) for h in (16, 67, 768) for s in (0.1, 2.1)]
FP8 = {"fn": (oracle.FP8_E4M3FN, torch.float8_e4m3fn), "fnuz": (oracle.FP8_E4M3FNUZ, torch.float8_e4m3fnuz)}


@pytest.mark.parametrize(("dname", "tokens", "hidden", "scale"), GRID)
def test_int8_quant_grid(golden, dname, tokens, hidden, scale):
    g = golden("quant_int8")
    key = f"{dname}_t{tokens}_h{hidden}_s{scale}"
    x = from_bits(g[f"x_{key}"], DT[dname])
    q = oracle.scaled_int8_quant_ref(x, torch.tensor([scale], dtype=torch.float32))
    np.testing.assert_array_equal(q.numpy(), g[f"q_{key}"])


@pytest.mark.parametrize("dname", list(DT))
@pytest.mark.parametrize("sname", ["one", "s2p1", "zdim"])
def test_int8_quant_edges(golden, dname, sname):
    g = golden("quant_int8")
    x = from_bits(g[f"edge_x_{dname}_{sname}"], DT[dname])
    s = torch.from_numpy(g[f"edge_s_{dname}_{sname}"])
    q = oracle.scaled_int8_quant_ref(x, s)
    np.testing.assert_array_equal(q.numpy(), g[f"edge_q_{dname}_{sname}"])


@pytest.mark.parametrize("flav", list(FP8))
@pytest.mark.parametrize(("dname", "tokens", "hidden", "scale"), GRID)
def test_fp8_quant_grid(golden, flav, dname, tokens, hidden, scale):
    g = golden("quant_fp8")
    key = f"{dname}_t{tokens}_h{hidden}_s{scale}"
    x = from_bits(g[f"x_{key}"], DT[dname])
    q = oracle.scaled_fp8_quant_ref(x, torch.tensor([scale], dtype=torch.float32), FP8[flav][0])
    np.testing.assert_array_equal(to_bits(q), g[f"q{flav}_{key}"])


@pytest.mark.parametrize("flav", list(FP8))
@pytest.mark.parametrize("dname", list(DT))
@pytest.mark.parametrize("sname", ["one", "s2p1", "zdim"])
def test_fp8_quant_edges(golden, flav, dname, sname):
    g = golden("quant_fp8")
    x = from_bits(g[f"edge_x_{dname}_{sname}"], DT[dname])
    s = torch.from_numpy(g[f"edge_s_{dname}_{sname}"])
    q = oracle.scaled_fp8_quant_ref(x, s, FP8[flav][0])
    np.testing.assert_array_equal(to_bits(q), g[f"edge_q{flav}_{dname}_{sname}"])


@pytest.mark.parametrize("flav", list(FP8))
def test_fp8_codec_matches_torch_cast(flav):
    """The numpy bit-level codec agrees with torch's cast on every code and on random floats."""
    name, tdt = FP8[flav]
    codes = np.arange(256, dtype=np.uint8)
    want = torch.from_numpy(codes.copy()).view(tdt).float().numpy()
    got = oracle.decode_fp8(codes, name)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(np.nan_to_num(got), np.nan_to_num(want))
    rng = np.random.default_rng(0)
    vals = np.concatenate(
        [
            (rng.standard_normal(20000) * np.exp(rng.uniform(-12, 7, 20000))).astype(np.float32),
            want[~np.isnan(want)],
            # exact midpoints between neighbouring codes (ties-to-even cases)
            ((want[:-1] + want[1:]) / 2)[~np.isnan(want[:-1] + want[1:])],
            np.array([0.0, -0.0, 448.0, 464.0, 465.0, 240.0, 248.0, 249.0, 1e9, -1e9], dtype=np.float32),
        ]
    ).astype(np.float32)
    want_codes = torch.from_numpy(vals.copy()).to(tdt).view(torch.uint8).numpy()
    np.testing.assert_array_equal(oracle.encode_fp8(vals, name), want_codes)


IN_T = {"int8": torch.int8, "fn": torch.float8_e4m3fn, "fnuz": torch.float8_e4m3fnuz}
GEMM_KEYS = [
    (i, o, sa, sb, b) for i in IN_T for o in ("f16", "bf16") for sa in (1, 0) for sb in (1, 0) for b in (1, 0)
]


@pytest.mark.parametrize(("iname", "oname", "sa_s", "sb_s", "use_bias"), GEMM_KEYS)
def test_scaled_gemm(golden, iname, oname, sa_s, sb_s, use_bias):
    g = golden("scaled_gemm")
    key = f"{iname}_{oname}_sa{sa_s}_sb{sb_s}_b{use_bias}"
    a = from_bits(g[f"a_{key}"], IN_T[iname])
    b = from_bits(g[f"bt_{key}"], IN_T[iname]).T
    sa = torch.from_numpy(g[f"sa_{key}"])
    sb = torch.from_numpy(g[f"sb_{key}"])
    bias = from_bits(g[f"bias_{key}"], DT[oname]) if use_bias else None
    c = oracle.scaled_gemm_ref(a, b, sa, sb, DT[oname], bias)
    np.testing.assert_array_equal(to_bits(c), g[f"c_{key}"])


@pytest.mark.parametrize(("iname", "oname", "sa_s", "sb_s", "use_bias"), GEMM_KEYS)
def test_scaled_gemm_silu_and_mul(golden, iname, oname, sa_s, sb_s, use_bias):
    """The FFN pair (scaled_gemm -> silu_and_mul), expected outputs from the reference's two functions."""
    g, y = golden("scaled_gemm"), golden("scaled_gemm_silu")
    key = f"{iname}_{oname}_sa{sa_s}_sb{sb_s}_b{use_bias}"
    a = from_bits(g[f"a_{key}"], IN_T[iname])
    b = from_bits(g[f"bt_{key}"], IN_T[iname]).T
    sa = torch.from_numpy(g[f"sa_{key}"])
    sb = torch.from_numpy(g[f"sb_{key}"])
    bias = from_bits(g[f"bias_{key}"], DT[oname]) if use_bias else None
    got = oracle.scaled_gemm_silu_and_mul_ref(a, b, sa, sb, DT[oname], bias)
    assert got.shape == (a.shape[0], b.shape[1] // 2)
    np.testing.assert_array_equal(to_bits(got), y[f"y_{key}"])
    # the GeGLU pair (scaled_gemm -> the reference's gelu_tanh_and_mul)
    got = oracle.scaled_gemm_gelu_tanh_and_mul_ref(a, b, sa, sb, DT[oname], bias)
    np.testing.assert_array_equal(to_bits(got), golden("scaled_gemm_gelu")[f"y_{key}"])


MIXED_KEYS = [(w, z, d) for w in ("uint4b8", "uint8b128", "uint4", "uint8") for z in (1, 0) for d in ("f16", "bf16")]


@pytest.mark.parametrize(("wname", "zp", "dname"), MIXED_KEYS)
def test_quantize_pack_dequant_mixed(golden, wname, zp, dname):
    g = golden("mixed_gemm")
    key = f"{wname}_zp{zp}_{dname}"
    bits, bias, group = (int(v) for v in g[f"meta_{key}"])
    dtype = DT[dname]
    b = from_bits(g[f"b_{key}"], dtype)
    w_ref, w_q, w_s, w_zp = oracle.quantize_weights_ref(b, bits, bias, group, bool(zp))
    np.testing.assert_array_equal(w_q.numpy(), g[f"wq_{key}"])
    np.testing.assert_array_equal(to_bits(w_s), g[f"ws_{key}"])
    np.testing.assert_array_equal(to_bits(w_ref), g[f"wref_{key}"])
    if zp:
        np.testing.assert_array_equal(w_zp.numpy(), g[f"wzp_{key}"])
    else:
        assert w_zp is None
    packed = oracle.pack_rows_ref(w_q.numpy(), bits)
    np.testing.assert_array_equal(packed, g[f"packed_{key}"])
    np.testing.assert_array_equal(oracle.unpack_rows_ref(packed, bits), g[f"wq_{key}"])
    # the kernel-side dequantisation restated from the packed tensor is bit-identical to w_ref
    deq = oracle.dequantize_packed(torch.from_numpy(packed), w_s, w_zp, bits, bias, group)
    np.testing.assert_array_equal(to_bits(deq), g[f"wref_{key}"])
    a = from_bits(g[f"a_{key}"], dtype)
    c = oracle.mixed_precision_gemm_ref(a, w_ref)
    np.testing.assert_array_equal(to_bits(c), g[f"c_{key}"])
    # the FFN pair (mixed GEMM -> the reference's silu_and_mul)
    y = oracle.mixed_precision_gemm_silu_and_mul_ref(a, w_ref)
    np.testing.assert_array_equal(to_bits(y), golden("mixed_gemm_silu")[f"y_{key}"])
    y = oracle.mixed_precision_gemm_gelu_tanh_and_mul_ref(a, w_ref)
    np.testing.assert_array_equal(to_bits(y), golden("mixed_gemm_gelu")[f"y_{key}"])


DYN_CASES = [(d, t, h) for d in DT for t, h in ((1, 16), (7, 67), (5, 768), (3, 2064), (2, 8200))]


@pytest.mark.parametrize(("dname", "tokens", "hidden"), DYN_CASES)
def test_dynamic_quant_rows_equal_reference_static_oracle(golden, dname, tokens, hidden):
    """SURVEY.md 8(f) N1.  Golden: the REAL reference's static oracle applied row by row with scale = absmax / QMAX
    (tests/golden/make_golden_r2.py); the oracle's vectorised restatement must reproduce codes AND scales bit for bit."""
    g = golden("quant_dynamic")
    key = f"{dname}_t{tokens}_h{hidden}"
    x = from_bits(g[f"x_{key}"], DT[dname])
    q, s = oracle.dynamic_scaled_int8_quant_ref(x)
    np.testing.assert_array_equal(q.numpy(), g[f"qi8_{key}"])
    np.testing.assert_array_equal(s.reshape(-1).numpy().view(np.uint32), g[f"si8_{key}"].view(np.uint32))
    for flav, (name, _) in FP8.items():
        q8, s8 = oracle.dynamic_scaled_fp8_quant_ref(x, name)
        np.testing.assert_array_equal(to_bits(q8), g[f"q{flav}_{key}"])
        np.testing.assert_array_equal(s8.reshape(-1).numpy().view(np.uint32), g[f"s{flav}_{key}"].view(np.uint32))


# ---------------------------------------------------------------------------------------------
# kernel modes conch.ops never selects: golden = the reference's OWN Triton kernel under the CPU
# interpreter (tests/golden/make_golden_r2.py gen_modes), fp16
# ---------------------------------------------------------------------------------------------
MODE_TAGS = [(2, 2), (4, 8), (4, 0), (8, 128), (8, 0)]
MODE_CASES = {
    "none": (oracle.GROUP_NONE, oracle.CHANNEL_NONE, "ws", None),
    "shift": (oracle.GROUP_SHIFT, oracle.CHANNEL_NONE, "ws", "zi"),
    "shift_scalar": (oracle.GROUP_SHIFT, oracle.CHANNEL_NONE, "ws", "scalar"),
    "sym": (oracle.GROUP_SYMMETRIC_NO_SHIFT, oracle.CHANNEL_NONE, "ws", None),
    "symshift": (oracle.GROUP_SYMMETRIC_WITH_SHIFT, oracle.CHANNEL_NONE, "ws", "zi"),
    "symshift_scalar": (oracle.GROUP_SYMMETRIC_WITH_SHIFT, oracle.CHANNEL_NONE, "ws", "scalar"),
    "asym": (oracle.GROUP_ASYMMETRIC, oracle.CHANNEL_NONE, "ws", "zf"),
    "none_wonly": (oracle.GROUP_NONE, oracle.CHANNEL_WEIGHT_ONLY, "cs", None),
    "shift_wonly": (oracle.GROUP_SHIFT, oracle.CHANNEL_WEIGHT_ONLY, "cs", "zi"),
}


def mode_inputs(g, nbits, bias, case):
    tag = f"b{nbits}_bias{bias}"
    gmode, cmode, sname, zname = MODE_CASES[case]
    packed = torch.from_numpy(g[f"packed_{tag}"].copy())
    ws = from_bits(g[f"ws_{tag}"], torch.float16)
    cs = from_bits(g[f"cs_{tag}"], torch.float16)
    zeros = None
    if zname == "zi":
        zeros = torch.from_numpy(g[f"zi_{tag}"].copy())
    elif zname == "zf":
        zeros = from_bits(g[f"zf_{tag}"], torch.float16)
    elif zname == "scalar":
        zeros = torch.tensor([3], dtype=torch.int32)
    return tag, gmode, cmode, packed, ws, cs, zeros


@pytest.mark.parametrize("case", list(MODE_CASES))
@pytest.mark.parametrize(("nbits", "bias"), MODE_TAGS)
def test_kernel_modes_match_reference_triton_kernel(golden, nbits, bias, case):
    g = golden("gemm_modes")
    tag, gmode, cmode, packed, ws, cs, zeros = mode_inputs(g, nbits, bias, case)
    w = oracle.dequantize_modes_ref(packed, ws, zeros, nbits, bias, 64, gmode, torch.float16)
    # identity activations: the kernel's output IS its dequantised weight matrix (times the channel scale) -> bit for bit
    eye = torch.eye(128, dtype=torch.float16)
    got = oracle.mixed_precision_gemm_modes_ref(eye, w, cs, cmode, torch.float16)
    np.testing.assert_array_equal(to_bits(got), g[f"c_{tag}_{case}_eye"])
    # random activations: fp32 accumulation order of the interpreter's dot vs torch's matmul -> one output ulp
    x = from_bits(g["x_rand"], torch.float16)
    got = oracle.mixed_precision_gemm_modes_ref(x, w, cs, cmode, torch.float16).float()
    want = from_bits(g[f"c_{tag}_{case}_rand"], torch.float16).float()
    assert (got - want).abs().max().item() <= 2.0**-10 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize("oname", ["f16"])
@pytest.mark.parametrize("cname", ["none", "wonly", "aonly", "wa"])
def test_scaled_channel_modes_match_reference_triton_kernel(golden, oname, cname):
    g = golden("gemm_modes")
    a = torch.from_numpy(g["sc_a"].copy())
    b = torch.from_numpy(g["sc_bt"].copy()).T
    sa, sb = torch.from_numpy(g["sc_sa"].copy()), torch.from_numpy(g["sc_sb"].copy())
    cmode = {"none": oracle.CHANNEL_NONE, "wonly": oracle.CHANNEL_WEIGHT_ONLY, "aonly": oracle.CHANNEL_ACTIVATION_ONLY,
             "wa": oracle.CHANNEL_WEIGHT_AND_ACTIVATION}[cname]
    got = oracle.scaled_gemm_modes_ref(a, b, sa, sb, DT[oname], cmode)
    want = from_bits(g[f"sc_c_{oname}_{cname}"], DT[oname])
    if cname == "wa":
        # the Triton kernel multiplies (sa * sb) first (gemm.py:440), the PyTorch oracle sb * (sa * acc)
        # (reference/quantization/scaled_gemm.py:21-22): one output ulp apart on a few elements
        diff = (got.float() - want.float()).abs()
        eps = 2.0**-10 if oname == "f16" else 2.0**-7
        assert (diff <= eps * want.float().abs()).all() and (diff > 0).float().mean().item() < 0.01
    else:
        np.testing.assert_array_equal(to_bits(got), to_bits(want))
