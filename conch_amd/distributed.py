"""N-sharded scaled GEMM across the GPUs of one node (BASELINE config C5).

The reference has no distributed code at all; this is the multi-GPU form of the same hot path.
Output column j depends only on B[:, j], scale_b[j] and bias[j], so N is partitioned into
`world_size` contiguous column blocks: every rank holds all of A / scale_a and its block of
B / scale_b / bias, computes its [M, N/G] block of C with the single-GPU kernel (no reduction, so
every element is bit-identical to the single-GPU result), and ONE collective -- an all-gather of the
blocks over RCCL/xGMI -- assembles C.

Layout: an all-gather concatenates along dim 0, so the gathered tensor is
[G, M, N/G] ("column-block major").  `gathered_blocks()` returns that zero-copy view for consumers
that are themselves column-parallel; `__call__` returns the row-major [M, N] tensor (one permuting
copy).  One process per GPU; backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""

from __future__ import annotations

from collections.abc import Callable

import torch
import torch.distributed as dist


def shard_bounds(n: int, world_size: int, rank: int) -> tuple[int, int]:
    """Column range [lo, hi) of `rank`; N must divide evenly (the all-gather needs equal blocks)."""
    if n % world_size:
        raise ValueError(f"N={n} is not divisible by world_size={world_size}")
    per = n // world_size
    return rank * per, (rank + 1) * per


def _default_gemm(a, b, scale_a, scale_b, output_dtype, bias):
    from conch_amd.ops.quantization.gemm import scaled_gemm

    return scaled_gemm(a, b, scale_a, scale_b, output_dtype, bias)


class NShardedScaledGemm:
    """scaled_gemm with B, scale_b and bias sharded on N and C all-gathered.

    Buffers are allocated once (288 GB of HBM per GPU: the gathered C of config C5 is 470 MB) and
    reused by every call, so a call is: one GEMM launch + one all-gather (+ one permuting copy for
    the row-major form).
    """

    def __init__(
        self,
        m: int,
        n: int,
        output_dtype: torch.dtype,
        device: torch.device,
        group: dist.ProcessGroup | None = None,
        gemm_fn: Callable | None = None,
    ) -> None:
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.m, self.n = m, n
        self.lo, self.hi = shard_bounds(n, self.world_size, self.rank)
        self.output_dtype = output_dtype
        self.gemm_fn = gemm_fn or _default_gemm
        self._blocks = torch.empty((self.world_size, m, self.hi - self.lo), dtype=output_dtype, device=device)

    def local_gemm(self, a, b_shard, scale_a, scale_b_shard, bias_shard=None) -> torch.Tensor:
        """This rank's [M, N/G] block of C (b_shard: [K, N/G], any strides)."""
        return self.gemm_fn(a, b_shard, scale_a, scale_b_shard, self.output_dtype, bias_shard)

    def gathered_blocks(self, a, b_shard, scale_a, scale_b_shard, bias_shard=None) -> torch.Tensor:
        """[G, M, N/G]: block g holds columns [g*N/G, (g+1)*N/G) of C."""
        c_loc = self.local_gemm(a, b_shard, scale_a, scale_b_shard, bias_shard)
        if self.world_size == 1:
            self._blocks[0].copy_(c_loc)
        else:
            dist.all_gather_into_tensor(self._blocks.view(-1, self.hi - self.lo), c_loc.contiguous(), group=self.group)
        return self._blocks

    def __call__(self, a, b_shard, scale_a, scale_b_shard, bias_shard=None) -> torch.Tensor:
        """Row-major [M, N] C, identical on every rank."""
        blocks = self.gathered_blocks(a, b_shard, scale_a, scale_b_shard, bias_shard)
        return blocks.permute(1, 0, 2).reshape(self.m, self.n)
