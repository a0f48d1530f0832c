"""world_size-2 gloo test of the N-sharded GEMM wrapper (runs on CPU).

The HIP kernel cannot run here, so the local GEMM is injected (the CPU oracle stands in for the
device kernel); what is under test is the sharding arithmetic, the collective and the block ->
row-major layout: the gathered C must equal the unsharded result bit-for-bit on every rank.
"""

from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conch_amd.distributed import NShardedScaledGemm, shard_bounds


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, in_dtype_name: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)  # same full problem on every rank
        m, k, n = 48, 128, 96
        if in_dtype_name == "int8":
            a = torch.randint(-32, 32, (m, k), dtype=torch.int8)
            b = torch.randint(-32, 32, (n, k), dtype=torch.int8).T
        else:
            a = (0.25 * torch.rand(m, k)).to(torch.float8_e4m3fn)
            b = (0.25 * torch.rand(n, k)).to(torch.float8_e4m3fn).T
        sa, sb = 0.25 * torch.rand(m, 1), 0.25 * torch.rand(n, 1)
        bias = torch.rand(n, dtype=torch.bfloat16)
        full = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
        lo, hi = shard_bounds(n, world, rank)
        for panels in (1, 3):  # one gather, and three row panels of 16 (gather of panel p behind the GEMM of panel p+1)
            op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cpu"), gemm_fn=oracle.scaled_gemm_ref, panels=panels)
            assert (op.lo, op.hi) == (lo, hi)
            for _ in range(2):  # buffers are reused across calls
                got = op(a, b[:, lo:hi], sa, sb[lo:hi], bias[lo:hi])
                assert got.shape == (m, n)
                assert torch.equal(got.view(torch.int16), full.view(torch.int16)), f"rank {rank}: gathered C differs"
            pan = op.gathered_panels(a, b[:, lo:hi], sa, sb[lo:hi], bias[lo:hi])
            assert pan.shape == (panels, world, m // panels, n // world)
            for p in range(panels):
                rows = slice(p * (m // panels), (p + 1) * (m // panels))
                for g in range(world):
                    glo, ghi = shard_bounds(n, world, g)
                    assert torch.equal(pan[p, g].view(torch.int16), full[rows, glo:ghi].contiguous().view(torch.int16))
            # the un-panelised contract: blocks[g] is rank g's [M, N/G] block whatever `panels` is
            for _ in range(2):
                blocks = op.gathered_blocks(a, b[:, lo:hi], sa, sb[lo:hi], bias[lo:hi])
                assert blocks.shape == (world, m, n // world)
                for g in range(world):
                    glo, ghi = shard_bounds(n, world, g)
                    assert torch.equal(blocks[g].view(torch.int16), full[:, glo:ghi].contiguous().view(torch.int16))
        # scalar scale_a is not sliced per panel
        op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cpu"), gemm_fn=oracle.scaled_gemm_ref, panels=2)
        s1 = torch.tensor([[0.5]])
        want = oracle.scaled_gemm_ref(a, b, s1, sb, torch.bfloat16, None)
        assert torch.equal(op(a, b[:, lo:hi], s1, sb[lo:hi]).view(torch.int16), want.view(torch.int16))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("in_dtype_name", ["int8", "fp8"])
def test_nsharded_gemm_world2_gloo(in_dtype_name):
    mp.spawn(_worker, args=(2, _free_port(), in_dtype_name), nprocs=2, join=True)


def test_nsharded_gemm_world4_gloo():
    """Four ranks (24 columns each): the same sharding arithmetic, collective and block -> row-major unpack as the two-rank test,
    with more than one FOREIGN block on either side of a rank's own (ranks 1 and 2) -- the layout the 4- and 8-GPU runs use."""
    mp.spawn(_worker, args=(4, _free_port(), "int8"), nprocs=4, join=True)


def test_shard_bounds():
    assert shard_bounds(11008, 8, 3) == (4128, 5504)
    assert shard_bounds(28672, 8, 7) == (25088, 28672)
    with pytest.raises(ValueError):
        shard_bounds(100, 8, 0)


def test_world1_writes_row_major_result_directly():
    """No process group: the local GEMM lands in the row-major buffer, gathered_blocks() is a view of it."""
    from conch_amd.distributed import default_panels

    torch.manual_seed(0)
    m, k, n = 32, 64, 48
    a = torch.randint(-32, 32, (m, k), dtype=torch.int8)
    b = torch.randint(-32, 32, (n, k), dtype=torch.int8).T
    sa, sb = 0.25 * torch.rand(m, 1), 0.25 * torch.rand(n, 1)
    full = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, None)
    op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cpu"), gemm_fn=oracle.scaled_gemm_ref, panels=2)
    got = op(a, b, sa, sb)
    assert torch.equal(got.view(torch.int16), full.view(torch.int16))
    pan = op.gathered_panels(a, b, sa, sb)
    assert pan.shape == (2, 1, 16, 48) and torch.equal(pan[1, 0].view(torch.int16), full[16:].view(torch.int16))
    blocks = op.gathered_blocks(a, b, sa, sb)
    assert blocks.shape == (1, m, n) and torch.equal(blocks[0].view(torch.int16), full.view(torch.int16))
    # C5 on 8 GPUs: 32 x 14 tiles per rank -> two panels of 224 tiles; C3 on one GPU: no split
    assert default_panels(8192, 3584) == 2
    assert default_panels(256, 3584) == 1


def test_default_panels_only_returns_divisors_of_m():
    """the heuristic counted tile rows (ceil(M / 256)) and could return a panel count M is not divisible by."""
    from conch_amd.distributed import default_panels

    assert default_panels(515, 65536) == 1     # 3 tile rows, 515 % 3 != 0
    assert default_panels(1001, 28672) == 1
    assert default_panels(768, 65536) == 3     # whole tile rows per panel
    for m in (1, 255, 257, 1001, 4096, 8192, 12288, 999999):
        for n_local in (64, 3584, 28672, 65536):
            p = default_panels(m, n_local)
            assert p >= 1 and m % p == 0
    # with the problem's K and dtype the rule counts the tiles of the kernel the dispatcher will launch: the
    # C5 shard of 3584 columns runs 256 x 224 assembly tiles -- 16 tile columns, not the 14 of the 256-wide model
    from conch_amd.distributed import planned_tile_columns

    assert planned_tile_columns(8192, 3584, 8192, torch.float8_e4m3fn) == 224
    assert planned_tile_columns(4096, 11008, 4096, torch.float8_e4m3fn) == 352
    assert planned_tile_columns(4096, 4096, 8192, torch.int8) == 256      # one full round of 256 x 256 HIP tiles
    assert planned_tile_columns(4096, 4096, 8192, torch.float16) == 256   # not a scaled-GEMM input dtype: the default
    assert default_panels(8192, 3584, cus=256, k=8192, in_dtype=torch.float8_e4m3fn) == 2   # two panels of 16 x 16 tiles = one full round each
    assert default_panels(8192, 3584, cus=256) == 2
    for m in (255, 4096, 8192, 12288):
        p = default_panels(m, 3584, cus=256, k=4096, in_dtype=torch.float8_e4m3fn)
        assert p >= 1 and m % p == 0
    # default arguments must construct for any M (one rank: no panels are used at all)
    for m, n in ((1001, 28672), (515, 65536), (7, 8)):
        op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cpu"), gemm_fn=oracle.scaled_gemm_ref)
        assert op.panels == 1


def _odd_m_worker(rank: int, world: int, port: int) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1)
        m, k, n = 77, 64, 8192  # odd M, wide N: default panels must fall back to 1, not raise
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8)
        b = torch.randint(-32, 32, (n, k), dtype=torch.int8).T
        sa, sb = 0.25 * torch.rand(m, 1), 0.25 * torch.rand(n, 1)
        full = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, None)
        op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cpu"), gemm_fn=oracle.scaled_gemm_ref)
        lo, hi = op.lo, op.hi
        assert torch.equal(op(a, b[:, lo:hi], sa, sb[lo:hi]).view(torch.int16), full.view(torch.int16))
    finally:
        dist.destroy_process_group()


def test_nsharded_gemm_world2_odd_m_default_panels():
    mp.spawn(_odd_m_worker, args=(2, _free_port()), nprocs=2, join=True)


def _one_rank_worker(rank: int, world: int, port: int) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(2)
        m, k, n = 64, 128, 96
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8)
        b = torch.randint(-32, 32, (n, k), dtype=torch.int8).T
        sa, sb = 0.25 * torch.rand(m, 1), 0.25 * torch.rand(n, 1)
        full = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, None)
        for panels in (1, 4):
            op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cpu"), gemm_fn=oracle.scaled_gemm_ref, panels=panels, force_collective=True)
            assert op._stage is not None  # the exchange path, not the one-rank shortcut
            for _ in range(2):
                assert torch.equal(op(a, b, sa, sb).view(torch.int16), full.view(torch.int16))
            assert torch.equal(op.gathered_blocks(a, b, sa, sb)[0].view(torch.int16), full.view(torch.int16))
            pan = op.gathered_panels(a, b, sa, sb)
            assert pan.shape == (panels, 1, m // panels, n)
            assert torch.equal(pan[panels - 1, 0].view(torch.int16), full[m - m // panels:].view(torch.int16))
    finally:
        dist.destroy_process_group()


def test_force_collective_runs_the_exchange_path_with_one_rank():
    """`force_collective=True` sends a one-rank group through the staging buffers and the all-gather (the GPU suite does the same
    through a one-rank RCCL group: the exchange path on the real backend, as far as a one-GPU box allows)."""
    mp.spawn(_one_rank_worker, args=(1, _free_port()), nprocs=1, join=True)


_CHILD = '''
import json, os, sys
import torch
import torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
t = torch.tensor([rank + 1.0])
dist.all_reduce(t)
print("library chatter on stdout, rank", rank, flush=True)
if "--fail" in sys.argv and rank == 1:
    sys.exit(7)
dist.barrier()
if rank == 0 and "--silent" not in sys.argv:
    print(json.dumps({"metric": "m", "value": t.item(), "n_gpus": world, "argv": sys.argv[1:]}), flush=True)
dist.destroy_process_group()
'''


def test_bench_starts_its_own_ranks_when_called_bare(tmp_path, capfd):
    """`python bench.py --gpus N` without a torchrun environment (how the driver calls N = 1) must start the N ranks itself --
    as child processes of a parent that never touches the GPU -- relay the ONE JSON line and the ranks' return code.  The
    ranks here are a stand-in script on gloo (bench.py proper needs a GPU); the launcher under test is bench.spawn_ranks."""
    import json
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench

    child = tmp_path / "child.py"
    child.write_text(_CHILD)
    assert bench.spawn_ranks(2, ["--gpus", "2", "--steps", "3"], script=child) == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert len(out) == 1, out  # the ranks' other stdout lines are not relayed
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["value"] == 3.0 and line["argv"] == ["--gpus", "2", "--steps", "3"]
    assert bench.spawn_ranks(2, ["--fail"], script=child) != 0  # a failing rank is a failing bench
    capfd.readouterr()
    assert bench.spawn_ranks(2, ["--silent"], script=child) == 1  # rc 0 without a line is not a result
    assert capfd.readouterr().out.strip() == ""


_RANK_SCRIPT = '''
"""Stand-in rank for bench.py's launcher and timing loop (gloo, CPU): the functions under test are bench.py's own."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, os.environ["CONCH_REPO_ROOT"])
import torch
import torch.distributed as dist
import bench

rank = int(os.environ["RANK"])
bench.arm_stack_dumps(rank)
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo")
world = dist.get_world_size()
bench.stall_if_asked(rank)
calls = [0]
buf = torch.zeros(4)

def step_with_a_collective():
    calls[0] += 1
    dist.all_reduce(buf)

if rank == 1:
    time.sleep(0.11)  # rank 1 enters the load phase late: its own wall clock would run out one batch after rank 0's
for _ in range(3):  # three timed regions back to back, as nshard_c5 runs them
    bench.timed_region(step_with_a_collective, 3, 2, world, torch.device("cpu"), ramp_s=0.05)
seen = [None] * world
dist.all_gather_object(seen, calls[0])
if rank == 0:
    print(json.dumps({"calls": seen}), flush=True)
dist.destroy_process_group()
'''


def _rank_script(tmp_path, monkeypatch):
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    monkeypatch.setenv("CONCH_REPO_ROOT", str(root))
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    return script


def _import_bench():
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench

    return bench


def test_timed_region_ranks_make_the_same_number_of_calls(tmp_path, monkeypatch, capfd):
    """Root cause of round 5's intermittent two-rank stall: the load phase in front of a timed region was a loop bounded by
    each rank's OWN wall clock around a step that holds collectives, so a rank that entered late made one more batch of
    all-gathers than its peer -- which by then sat in a barrier.  The ranks now vote after every batch.  Two gloo ranks, rank 1
    entering 0.11 s late into a 0.05 s load phase: the run must end, and both ranks must have made the same number of calls."""
    import json

    bench = _import_bench()
    rc = bench.spawn_ranks(2, [], script=_rank_script(tmp_path, monkeypatch), launch_timeout_s=60)
    cap = capfd.readouterr()
    assert rc == 0, cap.err[-3000:]
    calls = json.loads(cap.out.strip().splitlines()[-1])["calls"]
    assert calls[0] == calls[1] and calls[0] >= 3 * (20 + 2 + 3), calls


def test_launcher_ends_a_stalled_run_with_stacks_and_leaves_no_rank_behind(tmp_path, monkeypatch, capfd):
    """A rank that stops taking part (CONCH_BENCH_STALL_RANK: the switch tests/test_gpu_distributed.py uses on the device) must
    cost ONE launch timeout: the launcher collects every rank's stacks (SIGUSR1 -> faulthandler, which answers from inside a
    blocked collective too), kills the exact PIDs it started, returns 124 and prints no JSON line."""
    import re
    import time

    bench = _import_bench()
    monkeypatch.setenv("CONCH_BENCH_STALL_RANK", "1")
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, [], script=_rank_script(tmp_path, monkeypatch), launch_timeout_s=12)
    took = time.monotonic() - t0
    cap = capfd.readouterr()
    assert rc == 124 and took < 40, (rc, took, cap.err[-2000:])
    assert cap.out.strip() == ""
    assert "stall_if_asked" in cap.err, cap.err[-3000:]  # rank 1: where it sits
    assert "rank 0" in cap.err and "rank 1" in cap.err
    pids = [int(p) for p in re.search(r"bench.py ranks: ([\d ]+)", cap.err).group(1).split()]
    assert len(pids) == 2 and not any(os.path.exists(f"/proc/{p}") for p in pids), pids


def _direct_worker(rank: int, world: int, port: int, results) -> None:
    """Direct mode: every rank's row-major result is visible to every rank (here: CPU tensors in shared memory standing in for
    peer-mapped HBM), the product is stored into ALL of them by ONE call per panel (here: the oracle, injected), and the only
    collective is the completion barrier."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        m, k, n = 48, 128, 96
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8)
        b = torch.randint(-32, 32, (n, k), dtype=torch.int8).T
        sa, sb = 0.25 * torch.rand(m, 1), 0.25 * torch.rand(n, 1)
        bias = torch.rand(n, dtype=torch.bfloat16)
        full = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16, bias)
        calls = []

        def gemm_multi(outs, a_, b_, sa_, sb_, bias_):
            calls.append(len(outs))
            block = oracle.scaled_gemm_ref(a_, b_, sa_, sb_, torch.bfloat16, bias_)
            assert all(o.shape == block.shape and o.stride() == outs[0].stride() for o in outs)
            for o in outs:  # the kernel's epilogue stores every tile into every destination
                o.copy_(block)

        for panels in (1, 3):
            op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cpu"), panels=panels, direct=True,
                                    peer_results=(results, lambda: dist.barrier()), gemm_multi_fn=gemm_multi)
            lo, hi = op.lo, op.hi
            for _ in range(2):  # results are reused across calls
                calls.clear()
                got = op(a, b[:, lo:hi], sa, sb[lo:hi], bias[lo:hi])
                assert got is results[rank] and calls == [world] * panels  # ONE multi-destination product per panel
                assert torch.equal(got.view(torch.int16), full.view(torch.int16)), f"rank {rank}: direct result differs"
                dist.barrier()
            with pytest.raises(RuntimeError):
                op.gathered_blocks(a, b[:, lo:hi], sa, sb[lo:hi], bias[lo:hi])
    finally:
        dist.destroy_process_group()


def test_nsharded_gemm_direct_mode_world2_layout():
    """the epilogue writes its column block into every peer's row-major [M, N] result -- no staging, no
    all-gather, no unpack.  World size 2 on gloo with shared-memory results and an injected multi-destination product: the
    wrapper's layout logic (which view of which result, panel rows, barriers) bit for bit against the unsharded product."""
    results = [torch.zeros((48, 96), dtype=torch.bfloat16).share_memory_() for _ in range(2)]
    mp.spawn(_direct_worker, args=(2, _free_port(), results), nprocs=2, join=True)
