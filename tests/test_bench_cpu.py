"""bench.py's oracle check of the timed op (Leg.parity) without a GPU: the leg's inputs are built on the CPU, the oracle's own
output stands in for the device result (must pass) and a copy with ONE corrupted element must fail -- so a bench line whose
`parity.ok` is true means something."""

from __future__ import annotations

import sys
from pathlib import Path

import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
import oracle  # noqa: E402

CPU = torch.device("cpu")


@pytest.fixture()
def small_workloads(monkeypatch):
    monkeypatch.setitem(bench.WORKLOADS, "t_fp8", ("scaled_fp8", 384, 512, 256))
    monkeypatch.setitem(bench.WORKLOADS, "t_i8", ("scaled_int8", 128, 512, 256))
    monkeypatch.setitem(bench.WORKLOADS, "t_mixed", ("mixed_int4", 320, 256, 128))
    monkeypatch.setitem(bench.WORKLOADS, "t_quant", ("quant_int8", 64, 0, 96))


def _flip_one(t: torch.Tensor, row: int, col: int) -> torch.Tensor:
    bad = t.clone()
    if bad.dtype == torch.int8:
        bad[row, col] = bad[row, col] + 1 if bad[row, col] < 127 else 126
    else:
        bad[row, col] = bad[row, col] * 1.5 + 1.0
    return bad


@pytest.mark.parametrize("name", ["t_fp8", "t_i8"])
def test_parity_of_a_scaled_leg(small_workloads, name):
    leg = bench.Leg(name, CPU, seed=0)
    ref = oracle.scaled_gemm_ref(leg.a, leg.b, leg.sa, leg.sb, torch.bfloat16)
    res = leg.parity(ref)
    assert res["ok"] and res["max_excess"] <= 0 and res["checked_rows"] >= 100 and res["oracle"] == "oracle.scaled_gemm_ref"
    assert res["bit_exact"] is (name == "t_i8")
    # rows 0 and M-1 are always among the checked ones
    assert not leg.parity(_flip_one(ref, 0, 7))["ok"]
    assert not leg.parity(_flip_one(ref, leg.m - 1, leg.n - 1))["ok"]


def test_parity_of_the_mixed_leg(small_workloads):
    leg = bench.Leg("t_mixed", CPU, seed=0)
    # a device result: fp32 accumulation, one rounding to fp16
    got = (leg.x.float() @ leg.w_ref.float()).to(torch.float16)
    res = leg.parity(got)
    assert res["ok"] and res["reference_tolerance_met"] and res["checked_rows"] >= 100
    assert not leg.parity(_flip_one(got, 0, 3))["ok"]


def test_parity_of_the_quantiser_leg(small_workloads):
    leg = bench.Leg("t_quant", CPU, seed=0)
    want = oracle.scaled_int8_quant_ref(leg.x, leg.scale)
    res = leg.parity((want, leg.scale))
    assert res["ok"] and res["bit_exact"] and res["checked_rows"] == 64
    assert leg.parity(_flip_one(want, 5, 5))["max_excess"] == 1


def test_picked_kernel_names_the_assembly_kernel_for_c3():
    """The headline leg's `parity.kernel` is what the automatic dispatcher launches for C3 (no GPU needed: the plan is a pure
    function of the problem); the inputs are not built (4096 x 11008 on the CPU is not this test's business)."""
    leg = object.__new__(bench.Leg)
    leg.kind, leg.m, leg.k, leg.n = "scaled_fp8", 4096, 4096, 11008
    assert leg.picked_kernel().startswith("conch_gemm1w_fp8_bf16 256x352")
    leg.kind, leg.m, leg.k, leg.n = "scaled_int8", 128, 4096, 4096
    assert leg.picked_kernel().startswith("skinny_splitk_kernel")
    leg.kind, leg.m, leg.k, leg.n = "mixed_int4", 1024, 4096, 11008
    assert leg.picked_kernel().startswith("mixed_strip_kernel 256x192 tiles, 1 K slice")
