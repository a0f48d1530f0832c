"""HIP-graph capture of a decode step (include/conch_amd.h, "Library scratch"): after conch_reserve_scratch on the capture
stream every entry point is allocation-free and synchronisation-free, so a serving stack can capture its decode step once and
replay it; without the reservation a call that must grow a scratch slot inside a capture fails cleanly instead of breaking it."""

from __future__ import annotations

import pytest
import torch

from conch_amd import _C
from conch_amd.ops.quantization.gemm import mixed_precision_gemm, scaled_gemm
from conch_amd.ops.quantization.int8 import scaled_int8_quant
from conch_amd.third_party.vllm.quant_utils import pack_rows
from conch_amd.third_party.vllm.utils import seed_everything

pytestmark = pytest.mark.gpu


def _int4_weights(k, n, group=128):
    q = torch.randint(0, 16, (k, n), dtype=torch.int32)
    packed = pack_rows(q, 4, k, n)
    scales = (0.01 + 0.05 * torch.rand((k // group, n))).to(torch.float16)
    return packed.cuda(), scales.cuda()


def _decode_step(x, scale_x, sa, w1t, sb1, w2, s2):
    """quantise the activations, an int8 projection at decode batch size (split-K kernel: library scratch), an int4 GEMV-like
    projection (decode kernel: library scratch)."""
    q, _ = scaled_int8_quant(x, scale_x)
    y1 = scaled_gemm(q, w1t.T, sa, sb1, torch.bfloat16)
    y2 = mixed_precision_gemm(x, w2, s2, None, 4, 8, 128)
    return q, y1, y2


@pytest.mark.parametrize("m", [8, 128])
def test_decode_step_replays_from_a_hip_graph(m):
    seed_everything(11)
    k, n = 4096, 4096
    dev = torch.device("cuda")
    x = (torch.rand((m, k), device=dev) - 0.5).to(torch.float16)
    scale_x = torch.tensor([0.004], dtype=torch.float32, device=dev)
    sa = torch.full((m, 1), 0.004, dtype=torch.float32, device=dev)
    w1t = torch.randint(-32, 32, (n, k), dtype=torch.int8, device=dev)
    sb1 = 0.25 * torch.rand((n, 1), device=dev)
    w2, s2 = _int4_weights(k, n)
    stream = torch.cuda.Stream()
    lib = _C.load()
    need = max(lib.conch_scaled_gemm_workspace_bytes(m, n, k), lib.conch_mixed_precision_gemm_workspace_bytes(m, n, k))
    with torch.cuda.stream(stream):
        _C.reserve_scratch(need)  # keyed by (device, stream): the capture stream
        _decode_step(x, scale_x, sa, w1t, sb1, w2, s2)  # eager warm-up on that stream
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        outs = _decode_step(x, scale_x, sa, w1t, sb1, w2, s2)
    for trial in range(3):
        x.copy_((torch.rand((m, k), device=dev) - 0.5).to(torch.float16))
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in outs]
        want = _decode_step(x, scale_x, sa, w1t, sb1, w2, s2)
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            assert torch.equal(g.view(torch.uint8), w.view(torch.uint8)), f"trial {trial}: replayed step differs from the eager one"


def test_scratch_growth_inside_a_capture_is_refused_not_fatal():
    """A fresh stream has no scratch: the split-K path must fail with the documented error (-> ValueError) and leave the capture
    usable; after the reservation the same call captures."""
    seed_everything(12)
    m, k, n = 16, 2048, 1024
    dev = torch.device("cuda")
    a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device=dev)
    bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device=dev)
    sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
    want = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        graph.capture_begin()
        try:
            with pytest.raises(ValueError, match="reserve"):
                scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
        finally:
            graph.capture_end()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        _C.reserve_scratch(_C.load().conch_scaled_gemm_workspace_bytes(m, n, k))
    stream.synchronize()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=stream):
        out = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    graph2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


@pytest.mark.parametrize("special", [False, True])
def test_e4m3fnuz_gemm_captures_after_the_reservation_alone(special):
    """The e4m3fnuz flow (flag memset, scan, gated fp8 kernel, gated expansion + bf16 kernel: capi.hip run_scaled) is stream work
    only, and conch_reserve_scratch covers its flag word and its expansion buffers: it captures on a stream that never ran it
    eagerly, and a replay takes whichever branch the operands of THAT replay ask for."""
    seed_everything(14)
    m, k, n = 512, 1024, 768
    dev = torch.device("cuda")
    fnuz = torch.float8_e4m3fnuz
    a = (torch.rand((m, k), device=dev) - 0.5).to(fnuz)
    bt = (torch.rand((n, k), device=dev) - 0.5).to(fnuz)
    sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        _C.reserve_scratch(_C.load().conch_scaled_gemm_workspace_bytes(m, n, k))
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        out = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    for trial in range(2):
        if special and trial == 1:  # a code the fp8 MFMA reads differently (0x7F: 240 in fnuz, NaN in fn): the other branch
            a.view(torch.uint8)[3, 5] = 0x7F
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        want = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), f"trial {trial}"
        assert torch.isfinite(got.float()).all()


@pytest.mark.parametrize(("m", "k", "n"), [(256, 8192, 512), (64, 28672, 256), (512, 11008, 384)])
def test_split_k_forms_capture_after_the_reservation_alone(m, k, n):
    """The workspace query covers the slabs of every split-K form the dispatcher may pick (the 128 x 128 tiles' split, several passes
    per slice in the skinny kernel): a fresh stream, the reservation, a capture without an eager warm-up, replays equal to eager."""
    seed_everything(15)
    dev = torch.device("cuda")
    a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device=dev)
    bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device=dev)
    sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
    want = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        _C.reserve_scratch(_C.load().conch_scaled_gemm_workspace_bytes(m, n, k))
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        out = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)


@pytest.mark.parametrize(("m", "k", "n"), [(64, 4096, 11008), (128, 4096, 11008), (200, 8192, 8192)])
def test_mixed_k_split_strip_forms_capture_after_the_reservation_alone(m, k, n):
    """Batched-decode sizes of mixed_precision_gemm (round 5: the column-strip kernel on 64- / 128- / 256-row tiles, K split into
    fp32 slabs in library scratch + the slab reduce): conch_mixed_precision_gemm_workspace_bytes covers the slabs -- a fresh
    stream, the reservation, a capture without an eager warm-up, replays equal to the eager call."""
    seed_everything(16)
    dev = torch.device("cuda")
    x = (torch.rand((m, k), device=dev) - 0.3).to(torch.float16)
    w, s = _int4_weights(k, n)
    want = mixed_precision_gemm(x, w, s, None, 4, 8, 128)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        _C.reserve_scratch(_C.load().conch_mixed_precision_gemm_workspace_bytes(m, n, k))
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        out = mixed_precision_gemm(x, w, s, None, 4, 8, 128)
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)


def test_reset_scratch_is_harmless_between_one_launch_calls():
    """conch_reset_scratch zeroes the arrival counters of the current stream with a memset ordered ON that stream: between two
    one-launch split-K calls it changes nothing (the counters are zero there by construction); on a stream that never used a
    one-launch form it is a no-op."""
    seed_everything(13)
    m, k, n = 16, 4096, 4096  # M <= 32: the one-launch split-K form
    dev = torch.device("cuda")
    a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device=dev)
    bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device=dev)
    sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
    want = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    for _ in range(3):
        _C.reset_scratch()
        got = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
        assert torch.equal(got, want)
    fresh = torch.cuda.Stream()
    with torch.cuda.stream(fresh):
        _C.reset_scratch()
        _C.reset_scratch()
    fresh.synchronize()


def test_assembly_kernel_replays_from_a_hip_graph():
    """A prefill-size fp8 projection routed to the one-wave-per-SIMD assembly kernel (loaded with hipModuleLoadData, launched with
    hipModuleLaunchKernel: csrc/gemm_asm.hip) is captured and replayed like any other launch: the kernel-argument block is copied
    at capture, so replays see new operand CONTENTS through the same pointers, and the result equals the eager call bit for bit."""
    seed_everything(5)
    m, k, n = 2048, 1024, 2816  # 8 x 8 tiles of 256 x 352: the cost rule takes the assembly kernel
    dev = torch.device("cuda")
    a = (0.25 * torch.rand((m, k), device=dev)).to(torch.float8_e4m3fn)
    bt = (0.25 * torch.rand((n, k), device=dev)).to(torch.float8_e4m3fn)
    sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
    _C.set_gemm_variant(_C.VARIANT_MFMA_ASM1W)
    try:
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)  # eager warm-up on the capture stream (loads the module)
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            out = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
        for _ in range(3):
            a.copy_((0.25 * torch.rand((m, k), device=dev)).to(torch.float8_e4m3fn))
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, scaled_gemm(a, bt.T, sa, sb, torch.bfloat16))
    finally:
        _C.set_gemm_variant(_C.VARIANT_AUTO)


_COLD_CAPTURE = '''
import sys
sys.path.insert(0, sys.argv[1])
import torch
from conch_amd import _C
from conch_amd.ops.quantization.gemm import scaled_gemm

torch.manual_seed(5)
m, k, n = 2048, 1024, 2816  # 8 x 8 tiles of 256 x 352: the automatic dispatcher takes the assembly kernel
dev = torch.device("cuda")
a = (0.25 * torch.rand((m, k), device=dev)).to(torch.float8_e4m3fn)
bt = (0.25 * torch.rand((n, k), device=dev)).to(torch.float8_e4m3fn)
sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    _C.reserve_scratch(_C.load().conch_scaled_gemm_workspace_bytes(m, n, k))  # the ONLY library call in front of the capture
stream.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=stream):
    out = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)  # VARIANT_AUTO: first launch of the assembly kernel in this process
graph.replay()
torch.cuda.synchronize()
_C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2)
want = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
assert torch.equal(out, want), "captured assembly launch differs from the eager HIP kernel"
print("cold capture ok")
'''


def test_cold_capture_of_the_automatic_assembly_pick(tmp_path):
    """the assembly kernel's first use loads a code object (hipModuleLoadData: an allocation + upload), which must not
    happen inside a capture.  conch_reserve_scratch -- the call in front of every capture -- loads the modules for the device, so a
    FRESH process may capture a C3-like fp8 scaled_gemm under the automatic dispatcher with no eager warm-up at all.  A child
    process (this one has long loaded the module), outputs in files, bounded wait."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    script = tmp_path / "cold_capture.py"
    script.write_text(_COLD_CAPTURE)
    with open(tmp_path / "out", "wb") as fo, open(tmp_path / "err", "wb") as fe:
        proc = subprocess.Popen([sys.executable, str(script), str(root)], stdout=fo, stderr=fe, stdin=subprocess.DEVNULL)
        try:
            rc = proc.wait(timeout=180)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait(timeout=15)
            rc = -9
    out, err = (tmp_path / "out").read_text(errors="replace"), (tmp_path / "err").read_text(errors="replace")
    assert rc == 0 and "cold capture ok" in out, f"rc {rc}\n{out[-1500:]}\n{err[-3000:]}"
