"""Two ranks on ONE GPU: the direct mode of the N-sharded scaled GEMM end to end on the device (round 4).

Each rank's row-major [M, N] result is mapped into the other rank's address space over HIP IPC, and the tile kernel's epilogue
(conch_scaled_gemm_multi) stores every tile into both results: no all-gather, no staging, no unpack.  Same-device "peers" say nothing
about xGMI bandwidth; they do exercise everything else -- handle exchange, the multi-destination kernel writing into ANOTHER
process's allocation, the completion barriers, buffer reuse across calls."""

from __future__ import annotations

import os
import re
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_direct_mode_two_ranks_share_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    # a CHILD process tree (this interpreter has initialised the GPU: it must not exec into another program)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(ROOT / "tools" / "try_direct_mode.py")],
                         env=env, capture_output=True, text=True, timeout=600, check=False)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-3000:]
    # the two ranks write to one pipe: their lines may interleave without a newline between them, so count matches, not lines
    verdicts = re.findall(r"rank (\d) call (\d): direct result equals the unsharded product: (True|False)", out)
    assert sorted(verdicts) == [(str(r), str(c), "True") for r in range(2) for c in range(3)], out[-3000:]  # 2 ranks x 3 calls
