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
        op = NShardedScaledGemm(m, n, torch.bfloat16, torch.device("cpu"), gemm_fn=oracle.scaled_gemm_ref)
        lo, hi = shard_bounds(n, world, rank)
        assert (op.lo, op.hi) == (lo, hi)
        got = op(a, b[:, lo:hi], sa, sb[lo:hi], bias[lo:hi])
        assert got.shape == (m, n)
        assert torch.equal(got.view(torch.int16), full.view(torch.int16)), f"rank {rank}: gathered C differs"
        blocks = op.gathered_blocks(a, b[:, lo:hi], sa, sb[lo:hi], bias[lo:hi])
        for g in range(world):
            glo, ghi = shard_bounds(n, world, g)
            assert torch.equal(blocks[g].view(torch.int16), full[:, glo:ghi].contiguous().view(torch.int16))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("in_dtype_name", ["int8", "fp8"])
def test_nsharded_gemm_world2_gloo(in_dtype_name):
    mp.spawn(_worker, args=(2, _free_port(), in_dtype_name), nprocs=2, join=True)


def test_shard_bounds():
    assert shard_bounds(11008, 8, 3) == (4128, 5504)
    assert shard_bounds(28672, 8, 7) == (25088, 28672)
    with pytest.raises(ValueError):
        shard_bounds(100, 8, 0)
