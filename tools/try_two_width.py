"""Experiment (round 6): what would an out-of-step two-width launch of C3 buy?

No kernel change: the assembly scaled GEMM is launched TWICE at the same time on two streams, each capped at 128 persistent
workgroups (conch_debug_gemm1w_grid) -- columns [0, 5632) as 16 x 352-column tiles (two per workgroup) beside columns
[5632, 11008) as 24 x 224-column tiles (three per workgroup): 256 workgroups, one per CU, an exact cover of N = 11008, tile ends
of the two halves at different times.  Compared, interleaved, with the one-launch form (512 tiles of 256 x 352, two per CU) under
the same fork / join harness.  usage: python tools/try_two_width.py [--iters 200] [--rounds 5]"""
import argparse
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization.gemm import scaled_gemm_launcher  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--split", type=int, default=5632, help="first column of the 224-wide half (a multiple of 352)")
    args = ap.parse_args()
    lib = _C.load()
    for fn in (lib.conch_debug_gemm1w_width, lib.conch_debug_gemm1w_grid):
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int]
    dev = torch.device("cuda")
    m, k, n = 4096, 4096, 11008
    torch.manual_seed(0)
    a = (0.25 * torch.rand((m, k), device=dev)).to(torch.float8_e4m3fn)
    bt = (0.25 * torch.rand((n, k), device=dev)).to(torch.float8_e4m3fn)
    sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
    out1 = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    out2 = torch.zeros((m, n), dtype=torch.bfloat16, device=dev)
    s = args.split
    md_full = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
    md_l = create_scaled_metadata(a, bt[:s].T, sa, sb[:s], torch.bfloat16)
    md_r = create_scaled_metadata(a, bt[s:].T, sa, sb[s:], torch.bfloat16)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    _C.set_gemm_variant(_C.VARIANT_MFMA_ASM1W)

    def one_launch() -> None:
        lib.conch_debug_gemm1w_width(11)
        lib.conch_debug_gemm1w_grid(0)
        with torch.cuda.stream(s1):
            scaled_gemm_launcher(out1, a, bt.T, sa, sb, md_full)

    def two_widths() -> None:
        lib.conch_debug_gemm1w_grid(128)
        lib.conch_debug_gemm1w_width(11)
        with torch.cuda.stream(s1):
            scaled_gemm_launcher(out2[:, :s], a, bt[:s].T, sa, sb[:s], md_l)
        lib.conch_debug_gemm1w_width(7)
        with torch.cuda.stream(s2):
            scaled_gemm_launcher(out2[:, s:], a, bt[s:].T, sa, sb[s:], md_r)

    def timed(fn, iters: int) -> float:
        """mean us per iteration; every iteration forks from and joins into the default stream (both forms pay the same)"""
        main_s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_s)
        for _ in range(iters):
            s1.wait_stream(main_s)
            s2.wait_stream(main_s)
            fn()
            main_s.wait_stream(s1)
            main_s.wait_stream(s2)
        e1.record(main_s)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    def free_running(two: bool, iters: int) -> float:
        """us per iteration when nothing joins the streams between iterations: each stream runs its launches back to back (the
        sustained, power-limited regime of the benchmark; the per-iteration fork / join above leaves the chip idle for ~28 us of
        every 134).  Two widths: the halves drift against each other freely; the time is that of the stream that ends last."""
        main_s = torch.cuda.current_stream()
        e0 = torch.cuda.Event(enable_timing=True)
        ends = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
        e0.record(main_s)
        s1.wait_stream(main_s)
        s2.wait_stream(main_s)
        if two:
            lib.conch_debug_gemm1w_grid(128)
            for _ in range(iters):
                lib.conch_debug_gemm1w_width(11)
                with torch.cuda.stream(s1):
                    scaled_gemm_launcher(out2[:, :s], a, bt[:s].T, sa, sb[:s], md_l)
                lib.conch_debug_gemm1w_width(7)
                with torch.cuda.stream(s2):
                    scaled_gemm_launcher(out2[:, s:], a, bt[s:].T, sa, sb[s:], md_r)
        else:
            lib.conch_debug_gemm1w_width(11)
            lib.conch_debug_gemm1w_grid(0)
            with torch.cuda.stream(s1):
                for _ in range(iters):
                    scaled_gemm_launcher(out1, a, bt.T, sa, sb, md_full)
        ends[0].record(s1)
        ends[1].record(s2)
        torch.cuda.synchronize()
        return max(e0.elapsed_time(ends[0]), e0.elapsed_time(ends[1])) / iters * 1e3, [round(e0.elapsed_time(e) / iters * 1e3, 2) for e in ends]

    try:
        one_launch()
        two_widths()
        torch.cuda.synchronize()
        print("two-width result equals the one-launch result:", torch.equal(out1, out2))
        for _ in range(3):  # load: clocks settle
            timed(one_launch, args.iters)
        for r in range(args.rounds):
            t1 = timed(one_launch, args.iters)
            t2 = timed(two_widths, args.iters)
            print(f"round {r}: one launch (512 tiles of 256x352) {t1:7.2f} us   two widths at once (256 of 352 | 384 of 224, 128 workgroups each) {t2:7.2f} us")
        for _ in range(2):
            free_running(False, args.iters)
        for r in range(args.rounds):
            t1, _ = free_running(False, 2 * args.iters)
            t2, per = free_running(True, 2 * args.iters)
            print(f"free-running round {r}: one launch {t1:7.2f} us   two widths on two streams {t2:7.2f} us (352-wide stream {per[0]}, 224-wide stream {per[1]} us per launch)")
    finally:
        lib.conch_debug_gemm1w_width(0)
        lib.conch_debug_gemm1w_grid(0)
        _C.set_gemm_variant(_C.VARIANT_AUTO)


if __name__ == "__main__":
    main()
