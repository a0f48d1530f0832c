"""Dry run of NShardedScaledGemm's DIRECT mode with two ranks on ONE GPU (development aid): torch symmetric memory between two
processes that share device 0, the multi-destination tile kernel storing into its own and the peer's result.  Functional only --
same-device "peers" say nothing about xGMI.  Rank 0 prints ONE JSON line with every rank's verdicts.
usage: python -c "import bench, sys; sys.exit(bench.spawn_ranks(2, [], script='tools/try_direct_mode.py', launch_timeout_s=120))"
(or python -m torch.distributed.run --nproc-per-node 2 tools/try_direct_mode.py)"""
import json
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd.distributed import NShardedScaledGemm  # noqa: E402
from conch_amd.ops.quantization.gemm import scaled_gemm  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import bench

    bench.arm_stack_dumps(rank)  # the launcher's timeout collects every rank's stacks
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo")
    torch.manual_seed(0)
    m, k, n = 1024, 512, 1024
    a = (0.25 * torch.rand((m, k), device=dev)).to(torch.float8_e4m3fn)
    bt = (0.25 * torch.rand((n, k), device=dev)).to(torch.float8_e4m3fn)
    sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
    full = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    try:
        op = NShardedScaledGemm(m, n, torch.bfloat16, dev, direct=True, panels=2)
    except Exception as exc:  # noqa: BLE001
        print(f"rank {rank}: direct mode unavailable here: {exc!r}", file=sys.stderr, flush=True)
        dist.destroy_process_group()
        sys.exit(3)
    lo, hi = op.lo, op.hi
    mine = []
    for it in range(3):
        got = op(a, bt[lo:hi].T, sa, sb[lo:hi])
        torch.cuda.synchronize()
        mine.append(bool(torch.equal(got, full)))
        print(f"rank {rank} call {it}: direct result equals the unsharded product: {mine[-1]}", file=sys.stderr, flush=True)
    seen = [None] * world
    dist.all_gather_object(seen, mine)
    if rank == 0:
        print(json.dumps({"direct_equals_unsharded": seen, "scheme": op.direct_scheme}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
