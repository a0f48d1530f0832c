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


def test_traffic_entries_are_written_from_counter_passes_and_refused_when_stale(tmp_path, monkeypatch):
    """profiles/traffic.json is written by tools/write_traffic.py from rocprofv3 --pmc CSVs (per-launch averages of the workload's
    dominant kernel; FETCH_SIZE doubled, KiB) and bench.load_traffic uses an entry only while the kernel's sources hash to what was
    measured (no hand-maintained numbers carried from round to round)."""
    import csv
    import json
    import shutil

    sys.path.insert(0, str(Path(bench.ROOT) / "tools"))
    import write_traffic as wt

    head = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
            "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value",
            "Start_Timestamp", "End_Timestamp"]

    def write(sub, rows):
        d = tmp_path / "prof" / sub / "runc"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(head)
            for disp, kernel, counter, value in rows:
                w.writerow([disp, disp, "Agent 2", 1, 1, 1, 65536, 3, kernel, 256, 0, 0, 512, 0, 96, counter, value, 0, 1])

    k = "conch_gemm1w_fp8_bf16"
    write("pmc_fetch", [(1, k, "FETCH_SIZE", 100000.0), (2, k, "FETCH_SIZE", 110000.0), (3, "some_other_kernel", "FETCH_SIZE", 5.0)])
    write("pmc_write", [(1, k, "WRITE_SIZE", 88000.0), (1, k, "TCC_HIT_sum", 900.0), (1, k, "TCC_MISS_sum", 100.0)])
    rec = wt.entry("c3", tmp_path / "prof", "abc1234")
    assert rec["fetch_kib"] == 105000.0 and rec["write_kib"] == 88000.0 and rec["tcc_hit_rate"] == 0.9
    assert rec["hbm_bytes_per_launch"] == 1024 * (2 * 105000 + 88000) and rec["commit"] == "abc1234"
    assert rec["sources_sha256"] == wt.sources_sha256("c3") and rec["algorithmic_bytes"] == 152103936

    # bench.load_traffic against a repo copy whose table holds this entry: accepted; after the kernel source changes: refused
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    for rel in wt.WORKLOADS["c3"][2]:
        (root / rel).parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(Path(bench.ROOT) / rel, root / rel)
    (root / "profiles" / "traffic.json").write_text(json.dumps({"c3": rec, "c1": {"hbm_bytes_per_launch": 1}}))
    monkeypatch.setattr(bench, "ROOT", root)
    assert bench.load_traffic("c3")[0] == rec["hbm_bytes_per_launch"] and bench.traffic_commit("c3") == "abc1234"
    value, why = bench.load_traffic("c1")
    assert value is None and "sources_sha256" in why          # a hand-written entry (no hash) is refused
    with open(root / wt.WORKLOADS["c3"][2][1], "a") as f:
        f.write("\n// changed\n")
    value, why = bench.load_traffic("c3")
    assert value is None and why.startswith("refused") and "abc1234" in why
    assert bench.load_traffic("c2") == (None, None)             # no entry at all


def test_committed_traffic_table_matches_the_kernel_sources():
    """Every entry of the committed profiles/traffic.json was measured on the kernel sources of this tree (else bench.py would
    report traffic null on the driver's box): a kernel edit without a new counter pass fails HERE."""
    import json

    table = json.loads((Path(bench.ROOT) / "profiles" / "traffic.json").read_text())
    for name in ("c1", "c2", "c3", "c4", "c4readme"):
        value, why = bench.load_traffic(name)
        assert value == table[name]["hbm_bytes_per_launch"], (name, why)
