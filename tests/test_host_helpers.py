"""CPU-only tests of the host side: C-ABI library surface, format helpers, metadata / error paths."""

from __future__ import annotations

import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from conch_amd import _C
from conch_amd.kernels.quantization.gemm import ChannelScaleMode, WeightGroupMode
from conch_amd.ops.quantization.fp8 import scaled_fp8_quant
from conch_amd.ops.quantization.gemm import (
    create_mixed_precision_metadata,
    create_scaled_metadata,
    mixed_precision_gemm,
    mixed_precision_gemm_silu_and_mul,
    scaled_gemm,
    scaled_gemm_silu_and_mul,
)
from conch_amd.ops.quantization.int8 import scaled_int8_quant
from conch_amd.third_party.vllm.quant_utils import pack_rows, quantize_weights
from conch_amd.third_party.vllm.scalar_type import scalar_types
from tests.conftest import DT, ROOT, from_bits, to_bits

WTYPES = {
    "uint4b8": scalar_types.uint4b8,
    "uint8b128": scalar_types.uint8b128,
    "uint4": scalar_types.uint4,
    "uint8": scalar_types.uint8,
}


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "conch_amd.h").read_text()
    declared = set(re.findall(r"\b(conch_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed from include/conch_amd.h"
    assert declared == set(_C.EXPORTED_SYMBOLS)
    lib = _C.load()
    for name in declared:
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr)  # noqa: SLF001
    assert lib.conch_abi_version() == _C.ABI_VERSION
    assert int(re.search(r"#define CONCH_AMD_ABI_VERSION (\d+)", header).group(1)) == _C.ABI_VERSION


def test_c_abi_validation_without_gpu():
    """Argument validation happens before any HIP call, so it can be exercised on a CPU box."""
    lib = _C.load()
    rc = lib.conch_scaled_gemm(None, None, None, None, None, None, 4, 4, 4, 4, 1, 1, 4, 4, 1, 1, 1, 77, _C.DT_BF16, None)
    assert rc == 2 and b"input dtype" in lib.conch_last_error()
    rc = lib.conch_scaled_gemm(None, None, None, None, None, None, 4, 4, 4, 4, 1, 1, 4, 4, 1, 1, 1, _C.DT_INT8, _C.DT_BF16, None)
    assert rc == 1 and b"NULL" in lib.conch_last_error()
    rc = lib.conch_scaled_gemm(None, None, None, None, None, None, -1, 4, 4, 4, 1, 1, 4, 4, 1, 1, 1, _C.DT_INT8, _C.DT_BF16, None)
    assert rc == 1
    # empty problems are a no-op
    assert lib.conch_scaled_gemm(None, None, None, None, None, None, 0, 4, 4, 4, 1, 1, 4, 4, 1, 1, 1, _C.DT_INT8, _C.DT_BF16, None) == 0
    assert lib.conch_static_scaled_int8_quant(None, None, None, 0, 16, 16, 16, _C.DT_FP16, None) == 0
    assert lib.conch_static_scaled_int8_quant(None, None, None, 2, 16, 16, 16, _C.DT_FP16, None) == 1
    rc = lib.conch_mixed_precision_gemm(None, None, None, None, None, 4, 4, 8, 8, 4, 4, 4, 4, 3, 0, 8, 0, _C.DT_FP16, _C.DT_FP16, None)
    assert rc == 1 and b"weight_bits" in lib.conch_last_error()
    # the modes entry point: ACTIVATION_ONLY needs scales the mixed launcher never passes
    rc = lib.conch_mixed_precision_gemm_modes(None, None, None, None, None, None, 4, 4, 8, 8, 4, 4, 4, 4, 4, 0, 8, 0, 0, 2,
                                              _C.DT_FP16, _C.DT_FP16, None)
    assert rc == 2 and b"activation scales" in lib.conch_last_error()
    with pytest.raises(ValueError):
        _C.check(1, "x")
    with pytest.raises(NotImplementedError):
        _C.check(2, "x")
    with pytest.raises(_C.ConchError):
        _C.check(3, "x")


def test_fused_ffn_ops_validation_without_gpu():
    """scaled_gemm_silu_and_mul / mixed_precision_gemm_silu_and_mul: C-ABI checks and host-side argument errors."""
    lib = _C.load()
    # the contained GEMM is validated on its 2 * n_out columns: bad dtype -> unsupported, NULL pointers -> invalid
    rc = lib.conch_scaled_gemm_silu_and_mul(None, None, None, None, None, None, 4, 4, 4, 4, 1, 1, 4, 4, 1, 1, 1, 77, _C.DT_BF16, None)
    assert rc == 2 and b"input dtype" in lib.conch_last_error()
    rc = lib.conch_scaled_gemm_silu_and_mul(None, None, None, None, None, None, 4, 4, 4, 4, 1, 1, 4, 4, 1, 1, 1, _C.DT_INT8, _C.DT_BF16, None)
    assert rc == 1 and b"NULL" in lib.conch_last_error()
    assert lib.conch_scaled_gemm_silu_and_mul(None, None, None, None, None, None, 0, 4, 4, 4, 1, 1, 4, 4, 1, 1, 1, _C.DT_INT8, _C.DT_BF16, None) == 0
    rc = lib.conch_mixed_precision_gemm_silu_and_mul(None, None, None, None, None, 4, 4, 8, 8, 4, 4, 4, 4, 3, 0, 8, 0, _C.DT_FP16, _C.DT_FP16, None)
    assert rc == 1 and b"weight_bits" in lib.conch_last_error()
    # the modes entry point: ACTIVATION_ONLY needs scales the mixed launcher never passes
    rc = lib.conch_mixed_precision_gemm_modes(None, None, None, None, None, None, 4, 4, 8, 8, 4, 4, 4, 4, 4, 0, 8, 0, 0, 2,
                                              _C.DT_FP16, _C.DT_FP16, None)
    assert rc == 2 and b"activation scales" in lib.conch_last_error()
    # host side: B / the packed weights need an even number of columns [gate | up]; host tensors are refused
    s = torch.tensor([1.0])
    a = torch.zeros(4, 128, dtype=torch.int8)
    b_odd = torch.zeros(7, 128, dtype=torch.int8).T
    with pytest.raises(ValueError, match="even number of columns"):
        scaled_gemm_silu_and_mul(a, b_odd, s, s, torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scaled_gemm_silu_and_mul(a, torch.zeros(8, 128, dtype=torch.int8).T, s, s, torch.bfloat16)
    x = torch.zeros(4, 128, dtype=torch.float16)
    with pytest.raises(ValueError, match="even number of columns"):
        mixed_precision_gemm_silu_and_mul(x, torch.zeros(16, 7, dtype=torch.int32), torch.ones(1, 7, dtype=torch.float16), None, 4, 8, 128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mixed_precision_gemm_silu_and_mul(x, torch.zeros(16, 8, dtype=torch.int32), torch.ones(1, 8, dtype=torch.float16), None, 4, 8, 128)


def test_tuning_knob_roundtrip():
    _C.set_gemm_variant(_C.VARIANT_GENERIC)
    assert _C.get_gemm_variant() == _C.VARIANT_GENERIC
    _C.set_gemm_variant(_C.VARIANT_AUTO)
    assert _C.get_gemm_variant() == 0


def test_every_tuning_key_of_the_header_is_settable_and_mirrored():
    """include/conch_amd.h's conch_tuning_key_t, conch_amd/_C.py's TUNE_* constants and the library's key range agree."""
    import re
    from pathlib import Path

    header = (Path(__file__).resolve().parent.parent / "include" / "conch_amd.h").read_text()
    keys = {name: int(val) for name, val in re.findall(r"CONCH_(TUNE_[A-Z0-9_]+) = (\d+)", header)}
    assert keys.pop("TUNE__COUNT") == len(keys) == _C.TUNE_COUNT  # the array bound, not a key
    assert sorted(keys.values()) == list(range(len(keys)))
    for name, val in keys.items():
        assert getattr(_C, name) == val, name
        _C.set_tuning(val, 1)
        assert _C.load().conch_get_tuning(val) == 1
        _C.set_tuning(val, 0)
    with pytest.raises(Exception):
        _C.set_tuning(len(keys), 1)


def test_ops_refuse_cpu_tensors():
    """No silent CPU fallback: host tensors are an error."""
    x = torch.rand(4, 16)
    s = torch.tensor([1.0])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scaled_int8_quant(x, s)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scaled_fp8_quant(x, s)
    a = torch.zeros(4, 128, dtype=torch.int8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scaled_gemm(a, a.T.contiguous().T, s, s, torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mixed_precision_gemm(torch.zeros(4, 128, dtype=torch.float16), torch.zeros(16, 8, dtype=torch.int32),
                             torch.ones(1, 8, dtype=torch.float16), None, 4, 8, 128)


def test_dynamic_quant_needs_a_device():
    # scale=None is dynamic per-token quantisation here (the reference raises NotImplementedError:
    # ops/quantization/int8.py:42-44, fp8.py:46-48); like every op it refuses host tensors
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scaled_int8_quant(torch.rand(2, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scaled_fp8_quant(torch.rand(2, 2))


def test_workspace_queries_and_reserve_validation():
    lib = _C.load()
    # split-K slabs dominate a decode-shaped call, operand copies a prefill-shaped one
    assert lib.conch_scaled_gemm_workspace_bytes(128, 4096, 4096) >= 4 * 128 * 4096 * 4
    assert lib.conch_scaled_gemm_workspace_bytes(4096, 11008, 4096) >= 2 * (4096 + 11008) * 4096
    assert lib.conch_mixed_precision_gemm_workspace_bytes(16, 11008, 4096) >= 4 * 16 * 11008 * 4
    assert lib.conch_scaled_gemm_workspace_bytes(0, 4, 4) == 0
    assert lib.conch_reserve_scratch(None, -1) == 1


def test_scalar_types():
    assert (scalar_types.uint4b8.size_bits, scalar_types.uint4b8.bias) == (4, 8)
    assert (scalar_types.uint4b8.min(), scalar_types.uint4b8.max()) == (-8, 7)
    assert (scalar_types.uint8b128.min(), scalar_types.uint8b128.max()) == (-128, 127)
    assert (scalar_types.uint4.min(), scalar_types.uint4.max()) == (0, 15)
    assert (scalar_types.uint8.min(), scalar_types.uint8.max()) == (0, 255)
    assert str(scalar_types.uint4b8) == "uint4b8" and str(scalar_types.uint8) == "uint8"


@pytest.mark.parametrize("wname", list(WTYPES))
@pytest.mark.parametrize("zp", [1, 0])
@pytest.mark.parametrize("dname", ["f16", "bf16"])
def test_quantize_weights_and_pack_rows_match_reference(golden, wname, zp, dname):
    g = golden("mixed_gemm")
    key = f"{wname}_zp{zp}_{dname}"
    wt = WTYPES[wname]
    b = from_bits(g[f"b_{key}"], DT[dname])
    w_ref, w_q, w_s, w_zp = quantize_weights(b, wt, 128, zero_points=bool(zp))
    np.testing.assert_array_equal(to_bits(w_ref), g[f"wref_{key}"])
    np.testing.assert_array_equal(w_q.numpy(), g[f"wq_{key}"])
    np.testing.assert_array_equal(to_bits(w_s), g[f"ws_{key}"])
    if zp:
        np.testing.assert_array_equal(w_zp.numpy(), g[f"wzp_{key}"])
    else:
        assert w_zp is None
    packed = pack_rows(w_q, wt.size_bits, *w_q.shape)
    assert packed.dtype == torch.int32
    np.testing.assert_array_equal(packed.numpy(), g[f"packed_{key}"])


def test_pack_rows_rejects_bad_shapes():
    with pytest.raises(ValueError):
        pack_rows(torch.zeros(7, 4, dtype=torch.int32), 4, 7, 4)
    with pytest.raises(ValueError):
        pack_rows(torch.zeros(8, 4, dtype=torch.int32), 4, 16, 4)


def test_scaled_metadata_and_strict_errors():
    a = torch.zeros(8, 128, dtype=torch.int8)
    b = torch.zeros(16, 128, dtype=torch.int8).T
    sa, sb = torch.ones(8, 1), torch.ones(16, 1)
    md = create_scaled_metadata(a, b, sa, sb, torch.bfloat16, strict=True)
    assert (md.m_dim, md.k_dim, md.n_dim) == (8, 128, 16)
    assert md.acc_dtype == torch.int32 and md.meta_dtype == torch.float32
    assert md.channel_scale_mode == ChannelScaleMode.WEIGHT_AND_ACTIVATION
    assert md.weight_group_mode == WeightGroupMode.NONE
    assert not md.data_contiguous  # b is a transposed view
    f8 = create_scaled_metadata(a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn), sa, sb, torch.float16)
    assert f8.acc_dtype == torch.float32
    with pytest.raises(ValueError, match="dimensions of input tensor a"):
        create_scaled_metadata(a[0], b, sa, sb, torch.bfloat16, strict=True)
    with pytest.raises(ValueError, match="same datatype"):
        create_scaled_metadata(a, b.to(torch.uint8), sa, sb, torch.bfloat16, strict=True)
    with pytest.raises(ValueError, match="scale_a shape"):
        create_scaled_metadata(a, b, torch.ones(7, 1), sb, torch.bfloat16, strict=True)
    with pytest.raises(ValueError, match="scale_b"):
        create_scaled_metadata(a, b, sa, torch.ones(16), torch.bfloat16, strict=True)
    # non-strict validates nothing
    create_scaled_metadata(a, b, torch.ones(7, 1), sb, torch.bfloat16)


def test_mixed_metadata_and_strict_errors():
    x = torch.zeros(4, 256, dtype=torch.float16)
    wq = torch.zeros(32, 8, dtype=torch.int32)
    ws = torch.ones(2, 8, dtype=torch.float16)
    zp = torch.zeros(2, 8, dtype=torch.int32)
    md = create_mixed_precision_metadata(x, wq, ws, None, 4, 8, 128, strict=True)
    assert (md.m_dim, md.k_dim, md.n_dim) == (4, 256, 8)
    assert md.elements_per_sample == 8 and md.unpack_mask == 15 and not md.zero_is_scalar
    assert md.weight_group_mode == WeightGroupMode.SYMMETRIC_NO_SHIFT
    assert md.output_dtype == torch.float16 and md.acc_dtype == torch.float32 and md.meta_dtype == torch.float16
    assert md.channel_scale_mode == ChannelScaleMode.NONE and md.data_contiguous
    md = create_mixed_precision_metadata(x, wq, ws, zp, 4, 8, 128, output_dtype=torch.bfloat16)
    assert md.weight_group_mode == WeightGroupMode.SYMMETRIC_WITH_SHIFT and md.output_dtype == torch.bfloat16
    md = create_mixed_precision_metadata(x, wq, ws, torch.zeros(1, dtype=torch.int32), 8, 128, 128)
    assert md.zero_is_scalar and md.elements_per_sample == 4 and md.unpack_mask == 255
    with pytest.raises(ValueError, match="w_q_packed"):
        create_mixed_precision_metadata(x, wq[0], ws, None, 4, 8, 128, strict=True)
    with pytest.raises(ValueError, match="packed weights"):
        create_mixed_precision_metadata(x, wq.to(torch.int64), ws, None, 4, 8, 128, strict=True)
    with pytest.raises(ValueError, match="w_s shape"):
        create_mixed_precision_metadata(x, wq, ws[:1], None, 4, 8, 128, strict=True)
    with pytest.raises(ValueError, match="w_zp shape"):
        create_mixed_precision_metadata(x, wq, ws, zp[:1], 4, 8, 128, strict=True)
    with pytest.raises(NotImplementedError):
        create_mixed_precision_metadata(x, wq, ws, None, 4, 8, 128, scaled_activations=True, strict=True)


def test_bnb_dynamic_map_and_shapes(golden):
    """Host side of the bitsandbytes API: the dynamic 8-bit map equals the reference's bit for bit; shape
    helpers follow functional.py:107-123; unsupported arguments raise like the reference's."""
    from conch_amd.ops.quantization.bitsandbytes import functional as F

    np.testing.assert_array_equal(F._create_dynamic_map().numpy().view(np.uint32), golden("bnb_blockwise")["dynamic_map"].view(np.uint32))
    assert F.get_absmax_shape(160, 64) == (3,)
    assert F.get_quantized_output_shape(161, "nf4") == (81, 1) and F.get_quantized_output_shape(161, "fp8") == (161,)
    assert F.get_quantized_output_shape(256, "fp4", torch.bfloat16) == (64, 1)
    with pytest.raises(NotImplementedError):
        F.quantize_blockwise(torch.zeros(64), quant_type="int4")
    with pytest.raises(NotImplementedError):
        F.quantize_blockwise(torch.zeros(64), blocksize=96)
    with pytest.raises(ValueError):
        F.dequantize_blockwise(torch.zeros(32, dtype=torch.uint8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.quantize_4bit(torch.zeros(128))
    lib = _C.load()
    assert lib.conch_bnb_quantize_blockwise(None, None, None, None, 128, 96, 0, _C.DT_FP16, _C.DT_FP32, None) == 1
    assert lib.conch_bnb_dequantize_blockwise(None, None, None, None, 128, 64, 2, _C.DT_FP16, _C.DT_FP32, None) == 1  # no code book
    assert lib.conch_bnb_dequantize_blockwise(None, None, None, None, 0, 64, 0, _C.DT_FP16, _C.DT_FP32, None) == 0


def test_matmul_4bit_rejects_states_that_do_not_describe_the_weight():
    """the C entry takes no buffer sizes, so a packed tensor / absmax vector / blocksize that does not match
    quant_state.shape must be refused in Python, before any device pointer is formed (runs without a GPU)."""
    from conch_amd.ops.quantization.bitsandbytes.functional import QuantState, matmul_4bit

    n, k, bs = 64, 128, 64
    x = torch.zeros((4, k), dtype=torch.float16)
    wq = torch.zeros((n * k // 2,), dtype=torch.uint8)
    absmax = torch.ones((n * k // bs,), dtype=torch.float32)

    def state(**kw):
        base = {"absmax": absmax, "shape": torch.Size((n, k)), "dtype": torch.float16, "blocksize": bs, "quant_type": "nf4", "code": None}
        base.update(kw)
        return QuantState(**base)

    with pytest.raises(ValueError, match="w_packed holds"):
        matmul_4bit(x, wq[:-1], state())
    with pytest.raises(ValueError, match="w_packed holds"):
        matmul_4bit(x, torch.zeros((n * k,), dtype=torch.uint8), state())
    with pytest.raises(ValueError, match="absmax has"):
        matmul_4bit(x, wq, state(absmax=absmax[:-1]))
    with pytest.raises(ValueError, match="absmax has"):
        matmul_4bit(x, wq, state(blocksize=128))  # absmax sized for blocks of 64
    with pytest.raises(NotImplementedError, match="blocksize"):
        matmul_4bit(x, wq, state(blocksize=96))
    with pytest.raises(ValueError, match="does not multiply"):
        matmul_4bit(x[:, :-1], wq, state())
    # a consistent state gets past the size checks and is refused only for living on the CPU
    with pytest.raises((RuntimeError, ValueError), match="(?i)device|cuda|rocm|gpu"):
        matmul_4bit(x, wq, state())


def test_headline_kernels_have_no_waterfalled_buffer_instructions():
    """Round 3: hipcc had kept the B operand's buffer descriptor in VGPRs and wrapped half of the scaled GEMM's LDS-DMA
    instructions in waterfall loops (4 % of C3).  The descriptor inputs now go through v_readfirstlane (common.hpp,
    make_uniform_rsrc); this compiles the two scaled tile-kernel sources to assembly (no GPU needed, ~25 s) and checks that no
    buffer instruction sits in such a loop.  tools/isa_waterfalls.py without arguments checks every source (minutes)."""
    import shutil
    import subprocess
    import sys

    from conch_amd import _build

    if shutil.which(_build.HIPCC) is None:
        pytest.skip(f"{_build.HIPCC} not found: the ISA check needs the compiler")
    csrc = ROOT / "conch_amd" / "csrc"
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "isa_waterfalls.py"), str(csrc / "gemm_mfma.hip"), str(csrc / "gemm_mid.hip")],
                         capture_output=True, text=True, check=False)
    assert res.returncode == 0, "tools/isa_waterfalls.py failed (a waterfall loop, or the compile itself):\n" + res.stdout + res.stderr


def test_host_shim_builds_loads_and_declines_what_it_does_not_take():
    """The C++ host path of the hot ops (conch_amd/csrc_host/host_shim.cpp): built in-tree by conch_amd._build.build_host_shim, bound
    to the SAME libconch_amd.so, and strictly a fast path -- anything but the plain device case gets None and the Python
    launchers decide (here: host tensors, which they refuse loudly)."""
    from conch_amd import _build
    from conch_amd.kernels.quantization import _fast

    assert _build.build_host_shim().exists()
    host = _fast._load_host_shim()
    assert host is not None, "conch_amd/_conch_host.so does not load against this torch build"
    a = torch.zeros(4, 128, dtype=torch.int8)
    s = torch.ones(1)
    assert host.scaled_gemm(a, a.T, s, s, torch.bfloat16, None) is None
    assert host.static_quant(torch.rand(4, 16), s, 0) is None
    x = torch.rand(4, 128, dtype=torch.float16)
    assert host.mixed_precision_gemm(x, torch.zeros(16, 8, dtype=torch.int32), torch.ones(1, 8, dtype=torch.float16), None, 4, 8, 128) is None


def test_install_as_conch_aliases_the_reference_import_paths():
    """`conch_amd.install_as_conch()`: the import lines of the reference's tests (tests/scaled_gemm_test.py:11,
    mixed_precision_gemm_test.py:12, int8_quant_kernels_test.py:11, fp8_quant_kernels_test.py:11) resolve to this package.
    Run in a child interpreter so that the aliases do not leak into this session."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import conch_amd\n"
        "names = conch_amd.install_as_conch()\n"
        "from conch.ops.quantization.gemm import scaled_gemm, mixed_precision_gemm\n"
        "from conch.ops.quantization.int8 import scaled_int8_quant\n"
        "from conch.ops.quantization.fp8 import scaled_fp8_quant\n"
        "from conch.third_party.vllm.quant_utils import quantize_weights, pack_rows\n"
        "from conch.third_party.vllm.scalar_type import scalar_types\n"
        "from conch.platforms import current_platform\n"
        "from conch.utils.benchmark import BenchmarkMetadata, benchmark_it\n"
        "import conch_amd.ops.quantization.gemm as g\n"
        "assert scaled_gemm is g.scaled_gemm and mixed_precision_gemm is g.mixed_precision_gemm\n"
        "assert 'conch' in names and 'conch.ops.quantization.gemm' in names\n"
        "print('ok')\n" % str(root)
    )
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, check=False)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), res.stdout + res.stderr


def test_mixed_dispatcher_picks_are_pinned_without_a_gpu():
    """conch_debug_mixed_plan: the kernel (and the strip kernel's tile rows x columns x K slices) run_mixed would launch, from the
    cost models alone (csrc/dispatch_fit.hpp; 256 CUs assumed without a device).  Pins the picks the round-5 sweeps measured as the
    fastest forms (profiles/r05/mixed_mid_sweep.txt, mixed_rows_tall_after.txt), so that a refit that moves one shows up here."""
    import ctypes

    lib = _C.load()
    fn = lib.conch_debug_mixed_plan
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)]
    out = (ctypes.c_int * 4)()

    def plan(m, k, n, bits=4, zp=0):
        assert fn(m, n, k, bits, 1, zp, out) == 0, lib.conch_last_error()
        return tuple(out)

    decode, tiles, strip = 1, 2, 3
    for m in (1, 16, 32):
        assert plan(m, 4096, 11008)[0] == decode                 # GEMV / small decode batches: the one-launch decode kernel
    assert plan(64, 4096, 4096)[0] == decode                      # narrow and shallow: nothing to gain from K slices
    assert plan(64, 4096, 11008) == (strip, 64, 192, 4)           # batched decode on a wide problem: 64-row tiles, four K slices
    assert plan(128, 4096, 11008) == (strip, 128, 192, 4)
    assert plan(64, 8192, 28672) == (strip, 64, 256, 2)
    assert plan(384, 4096, 11008) == (strip, 128, 192, 1)         # three rows of 128-row tiles instead of two of 256
    assert plan(1024, 4096, 4096) == (strip, 128, 128, 1)
    assert plan(1024, 4096, 11008) == (strip, 256, 192, 1)        # C4: the full-chip 256-row tile, unsplit (unchanged since round 4)
    assert plan(4096, 8192, 4096)[0] == tiles                     # the reference's README shape: 256 x 256 LDS tiles, one round
    try:
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)                    # "never split K": back to the round-4 choices
        assert plan(64, 4096, 11008)[0] == decode
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, 3)                    # the assembly kernel forced outside its contract (8-bit weights)
        assert fn(1024, 11008, 4096, 8, 1, 0, out) == 0 and out[0] == -1
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, 2)                    # the strip kernel forced at a decode size
        _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
        _C.set_tuning(_C.TUNE_MIXED_STRIP_ROWS, 128)
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 2)
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 3)
        assert plan(100, 4096, 4096) == (strip, 128, 128, 3)
    finally:
        for key in (_C.TUNE_MIXED_SPLITK, _C.TUNE_MIXED_KERNEL, _C.TUNE_MIXED_STRIP_ROWS, _C.TUNE_MIXED_TILE_NT):
            _C.set_tuning(key, 0)
        _C.set_gemm_variant(_C.VARIANT_AUTO)


def test_scaled_dispatcher_picks_are_pinned_without_a_gpu():
    """conch_debug_scaled_plan: the kernel the MFMA path of scaled_gemm would launch (cost models of csrc/dispatch_fit.hpp and
    kAsm1wFit; 256 CUs assumed without a device).  The BASELINE configurations and the shapes profiles/r05/asm1w_widths.txt
    measured: a refit that takes C3 off the assembly kernel fails here, not in a benchmark three rounds later."""
    import ctypes

    lib = _C.load()
    fn = lib.conch_debug_scaled_plan
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    out = (ctypes.c_int * 2)()
    fp8, int8 = 3, 4  # CONCH_DT_FP8_E4M3FN, CONCH_DT_INT8 (include/conch_amd.h)

    def plan(m, k, n, dt):
        assert fn(m, n, k, dt, out) == 0, lib.conch_last_error()
        return tuple(out)

    tiles256, tiles128, skinny, asm = 0, 1, 2, 3
    for dt in (fp8, int8):
        assert plan(4096, 4096, 11008, dt) == (asm, 352)      # C3: exactly two 256 x 352 tiles per CU
        assert plan(8192, 8192, 28672, dt) == (asm, 288)      # C5 on one GPU
        assert plan(8192, 8192, 3584, dt) == (asm, 224)       # C5's shard at 8 GPUs
        assert plan(2048, 4096, 11008, dt) == (asm, 352)
        assert plan(128, 4096, 4096, dt)[0] == skinny          # C2
        assert plan(512, 4096, 4096, dt)[0] == tiles128
        assert plan(4096, 8192, 4096, dt)[0] == tiles256       # 256 tiles of 256 x 256: one full round
    try:
        _C.set_gemm_variant(_C.VARIANT_MFMA_ASM1W)             # forced outside its contract (K = 384): an error, not another kernel
        assert fn(4096, 11008, 384, fp8, out) == 0 and out[0] == -1
    finally:
        _C.set_gemm_variant(_C.VARIANT_AUTO)


def test_mixed_plan_invariants_on_random_shapes():
    """conch_debug_mixed_plan on 400 random aligned shapes: whatever the cost models pick is a form the kernels have -- a known
    kernel; strip tiles of 64 / 128 / 256 rows x 128 / 192 / 256 columns in 1..8 K slices, never more slices than 128-element groups,
    K slices only where N % 4 == 0, the short tiles and the slices only up to the row count the models were fitted on; the decode
    kernel only up to its 256 rows."""
    import ctypes
    import random

    lib = _C.load()
    fn = lib.conch_debug_mixed_plan
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)]
    out = (ctypes.c_int * 4)()
    rng = random.Random(5)
    for _ in range(400):
        m = rng.choice([1, 7, 16, 32, 33, 48, 64, 65, 100, 128, 129, 200, 256, 257, 300, 384, 512, 700, 1024, 1500, 1536, 1537, 2048, 4096])
        k = 128 * rng.randint(1, 224)
        n = rng.choice([64, 128, 260, 520, 1376, 2048, 4096, 4100, 5120, 8192, 11008, 13824, 28672])
        bits, zp, x_dt = rng.choice([4, 8]), rng.choice([0, 1]), rng.choice([1, 2])  # CONCH_DT_FP16 / _BF16
        assert fn(m, n, k, bits, x_dt, zp, out) == 0, lib.conch_last_error()
        pick, rows, cols, slices = tuple(out)
        assert pick in (0, 1, 2, 3), (m, k, n, pick)
        if pick == 1:
            assert m <= 256
        if pick == 3:
            assert rows in (64, 128, 256) and cols in (128, 192, 256) and 1 <= slices <= 8, (m, k, n, tuple(out))
            assert slices <= k // 128 and (slices == 1 or n % 4 == 0), (m, k, n, tuple(out))
            assert not (x_dt == 2 and bits == 8)  # bf16 x 8-bit stays on the LDS-tiled kernel
            if m > 1536:
                assert rows == 256 and slices == 1, (m, k, n, tuple(out))
            if m <= 64:
                assert rows == 64
        else:
            assert (rows, cols, slices) == (0, 0, 0)
