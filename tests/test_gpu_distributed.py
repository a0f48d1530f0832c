"""Two ranks on ONE GPU: the direct mode of the N-sharded scaled GEMM end to end on the device (round 4).

Each rank's row-major [M, N] result is mapped into the other rank's address space over HIP IPC, and the tile kernel's epilogue
(conch_scaled_gemm_multi) stores every tile into both results: no all-gather, no staging, no unpack.  Same-device "peers" say nothing
about xGMI bandwidth; they do exercise everything else -- handle exchange, the multi-destination kernel writing into ANOTHER
process's allocation, the completion barriers, buffer reuse across calls."""

from __future__ import annotations

import json
import os
import re
import signal
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


@pytest.mark.parametrize("direct", [False, True])
def test_bench_two_ranks_on_device0(direct):
    """`bench.py --gpus 2` end to end with both ranks on cuda:0 (gloo rendezvous; a one-GPU box has no second device): the
    launcher's child process tree, the N-sharded GEMM, the gather + unpack path -- and with --c5-direct the epilogue-writes-to-peer
    form over HIP IPC -- and the exchange check bench.py runs before it times anything: every rank recomputes a band of every
    other rank's block from its seed and compares the gathered columns bit for bit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
           "--all-ranks-on-device0", "--c5-shape", "2048,1024,4096"]
    if direct:
        cmd.append("--c5-direct")
    # a CHILD process tree (this interpreter has initialised the GPU: it must not exec into another program), in its own session so
    # that a stuck rendezvous can be killed as a group.  One retry: in round 5 this command hung ONCE inside a full-suite run
    # (900 s, no output) and then passed 20 times in a row, alone, after other tests and with two processes hammering the GPU
    # (tools/stress_two_procs.sh) -- a gloo / torchrun rendezvous stall, not a kernel; a second stall fails the test.
    # (Two of the round's seven full passes stalled here again, both attempts; since then gloo is pinned to loopback in bench.py, a
    # stalled rank dumps its stacks after 100 s -- CONCH_BENCH_WATCHDOG_S -- and the dump is part of the failure message.)
    env["CONCH_BENCH_WATCHDOG_S"] = "100"
    res, stalled = None, []
    for attempt in range(3):
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = proc.communicate(timeout=150)
            res = subprocess.CompletedProcess(cmd, proc.returncode, out, err)
            break
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            out, err = proc.communicate()
            stalled.append((out or "")[-1500:] + (err or "")[-4000:])
    assert res is not None, "bench.py --gpus 2 stalled three times (150 s each); last output:\n" + stalled[-1]
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["detail"]["exchange_check_mismatches"] == 0
    assert line["value"] > 0
    if direct:
        assert "direct_error" not in line["detail"], line["detail"].get("direct_error")
        assert line["detail"]["direct_equals_allgather_result"] is True
