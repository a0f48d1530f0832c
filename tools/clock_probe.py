"""In-kernel clock of the GEMM K loops under sustained load (development aid; MI355X_MICROARCH.md DVFS item 6).

Loads the DIAGNOSTIC twin of the library (`python -m conch_amd._build --probe`, built on the CPU side so that it
travels to the GPU box), launches a workload back to back for a few seconds on random data and reads the
s_memtime / s_memrealtime stamps thread 0 of every workgroup took around its K loop:
    clock = d(memtime) / d(memrealtime) x 100 MHz,  loop time = d(memrealtime) / 100 MHz.
Prints the median over workgroups, the K-loop share of the launch, and the MFMA-pipe occupancy of the loop at THAT
clock (flops / (CUs x flops-per-clock-per-CU x clock x loop time)).

usage: python tools/clock_probe.py [seconds]
"""
import ctypes
import statistics
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from conch_amd import _C, _build  # noqa: E402

import os  # noqa: E402

# CONCH_PROBE_LIB=<path>: another diagnostic build (python -m conch_amd._build --variant X -DCONCH_CLOCK_PROBE -D...)
_C.LIB_PATH = Path(os.environ["CONCH_PROBE_LIB"]) if os.environ.get("CONCH_PROBE_LIB") else _build.PROBE_LIB
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata, create_scaled_metadata  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 2.5
N_CU = 256


def read_probe(fn_name: str, blocks: int):
    lib = _C.load()
    fn = getattr(lib, fn_name)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    blocks = min(blocks, 4096)
    buf = (ctypes.c_ulonglong * (8 * blocks))()
    rc = fn(buf, blocks)
    assert rc == 0, rc
    clocks, loops, spans = [], [], []
    for b in range(blocks):
        t0, r0, t1, r1, _, r_in, _, r_out = buf[8 * b:8 * b + 8]
        if r1 > r0:
            clocks.append((t1 - t0) / (r1 - r0) * 0.1)  # GHz
            loops.append((r1 - r0) * 0.01)  # us
            spans.append((r0, r1, r_in, r_out))
    if "--persist" in sys.argv:
        # tile b + 256 follows tile b on the same workgroup (persistent) or roughly the same CU slot (plain launch)
        raw = {b: tuple(buf[8 * b + i] for i in (1, 3, 5, 7)) for b in range(blocks) if buf[8 * b + 3] > buf[8 * b + 1]}
        gaps = sorted((raw[b + N_CU][0] - raw[b][1]) * 0.01 for b in raw if b + N_CU in raw)
        tails = sorted((raw[b + N_CU][0] - raw[b][3]) * 0.01 for b in raw if b + N_CU in raw and raw[b][3] > raw[b][1])
        if gaps:
            print(f"    K loop end of tile b -> K loop start of tile b+{N_CU}: median {statistics.median(gaps):.2f} us  p10 {gaps[len(gaps) // 10]:.2f}  p90 {gaps[len(gaps) * 9 // 10]:.2f}")
        if tails:
            print(f"    last store issued of tile b -> K loop start of tile b+{N_CU}: median {statistics.median(tails):.2f} us")
        first_steps = None
    if "--classes" in sys.argv and loops:
        # two-width tile schedules: the K loops fall into a short (192-column) and a long (256-column) class
        ls = sorted(loops)
        cut = (ls[0] + ls[-1]) / 2
        short, long_ = [v for v in ls if v < cut], [v for v in ls if v >= cut]
        for name, grp in (("short", short), ("long", long_)):
            if grp:
                print(f"    K loops, {name} class: n={len(grp)}  median {statistics.median(grp):.2f} us  min {grp[0]:.2f}  max {grp[-1]:.2f}")
        epi = sorted((sp[3] - sp[1]) * 0.01 for sp in spans if sp[3] > sp[1])
        pro = sorted((sp[0] - sp[2]) * 0.01 for sp in spans if sp[0] > sp[2]) or [0.0]
        if epi:
            print(f"    K loop end -> stores issued: median {statistics.median(epi):.2f} us  p10 {epi[len(epi) // 10]:.2f}  p90 {epi[len(epi) * 9 // 10]:.2f};  "
                  f"entry -> K loop: median {statistics.median(pro):.2f} us")
        first, last = min(sp[2] for sp in spans), max(sp[3] for sp in spans)
        print(f"    first entry -> last exit {(last - first) * 0.01:.1f} us; sum of K loops / CU {sum(loops) / N_CU:.1f} us")
    if "--timeline" in sys.argv and spans:
        # K-loop start / end of the workgroups of the LAST launch, in start order, relative to the first start
        spans.sort()
        base = spans[0][0]
        picks = sorted({0, len(spans) // 8, len(spans) // 4, 255, 256, 300, 400, 511, 512, 600, len(spans) - 1} & set(range(len(spans))))
        print("    timeline (us since the first K loop started): " + "  ".join(
            f"#{i}: {(spans[i][0] - base) * 0.01:.1f}-{(spans[i][1] - base) * 0.01:.1f}" for i in picks))
        if any(sp[3] > sp[1] for sp in spans):  # kernels that also stamp entry (2) and exit (3)
            pro = statistics.median((s[0] - s[2]) * 0.01 for s in spans)
            epi = statistics.median((s[3] - s[1]) * 0.01 for s in spans if s[3] > s[1])
            first_in = min(s[2] for s in spans)
            print(f"    entry -> K loop (prologue) {pro:.2f} us median; K loop end -> epilogue stores issued {epi:.2f} us median; "
                  f"first entry at {(first_in - base) * 0.01:.1f} us")
            outs = sorted(s[3] for s in spans if s[3] > s[1])
            ins = sorted(s[2] for s in spans)
            if len(ins) > 300:
                print(f"    workgroup #256 enters {(ins[256] - outs[0]) * 0.01:.2f} us after the first workgroup left")
        ends = sorted(s[1] for s in spans)
        print(f"    last K loop ends at {(ends[-1] - base) * 0.01:.1f} us; starts span {(spans[-1][0] - base) * 0.01:.1f} us")
    return statistics.median(clocks), statistics.median(loops), max(loops)


def sustained(launch, seconds):
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            launch()
        n += 50
        torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e6  # us per launch (host clock, includes launch gaps)


def scaled_case(m, k, n, dtype, flops_per_clk_cu, zeros=False):
    torch.manual_seed(0)
    if zeros:  # all-zero operands: the same instruction stream at the lowest switching activity (DVFS check)
        a = torch.zeros((m, k), device="cuda").to(dtype)
        b = torch.zeros((n, k), device="cuda").to(dtype).T
    elif dtype == torch.int8:
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
        b = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda").T
    else:
        a = (0.25 * torch.rand((m, k), device="cuda")).to(dtype)
        b = (0.25 * torch.rand((n, k), device="cuda")).to(dtype).T
    sa = 0.25 * torch.rand((m, 1), device="cuda")
    sb = 0.25 * torch.rand((n, 1), device="cuda")
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    md = create_scaled_metadata(a, b, sa, sb, out.dtype)
    us = sustained(lambda: kg.scaled_gemm_launcher(out, a, b, sa, sb, md), SECONDS)
    tiles = -(-m // 256) * -(-n // 256)
    ghz, loop_us, loop_max = read_probe("conch_debug_probe_scaled", -(-m // 256) * -(-n // 192) if "--sched" in sys.argv else tiles)
    rounds = -(-tiles // N_CU)
    # per-workgroup occupancy of its CU's matrix pipes during its own K loop
    occ = (2.0 * 256 * 256 * k) / (flops_per_clk_cu * ghz * 1e9 * loop_us * 1e-6)
    print(f"scaled {str(dtype)[6:]:14s}{' ZEROS' if zeros else ''} {m}x{k}x{n}: {us:7.1f} us/launch  clock {ghz:.2f} GHz  K loop {loop_us:.1f} us median "
          f"({loop_max:.1f} max) x {rounds} rounds  pipe occupancy in the loop {occ * 100:.0f} %  "
          f"({2.0 * m * n * k / us / 1e6:.0f} TFLOP/s)", flush=True)


def mixed_stalls(blocks=256):
    """Diagnostic counters of the mixed kernel: wave 0's cycles at the step-top s_waitcnt and at the barrier."""
    lib = _C.load()
    fn = lib.conch_debug_probe_mixed
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * (8 * blocks))()
    assert fn(buf, blocks) == 0
    rows = [(buf[8 * b + 2] - buf[8 * b], buf[8 * b + 4], buf[8 * b + 5], buf[8 * b + 6], buf[8 * b + 7]) for b in range(blocks)
            if buf[8 * b + 2] > buf[8 * b]]
    tot = statistics.median(r[0] for r in rows)
    if not any(r[1] for r in rows):  # the twin was built without -DCONCH_CLOCK_PROBE_STEPS (the default: the stamps perturb the loop)
        return
    print(f"    wave 0 of a workgroup: K loop {tot:.0f} cycles, of which {statistics.median(r[1] for r in rows) / tot * 100:.1f} % at the "
          f"step-top s_waitcnt and {statistics.median(r[2] for r in rows) / tot * 100:.1f} % at the barrier; wave 4 (late VMEM issue): "
          f"{statistics.median(r[3] for r in rows) / tot * 100:.1f} % / {statistics.median(r[4] for r in rows) / tot * 100:.1f} %")


def mixed_case(m, k, n, dtype, bits):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    pf = 32 // bits
    wq = torch.randint(-2**31, 2**31 - 1, (k // pf, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    out = torch.empty((m, n), dtype=dtype, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, None, bits, 8 if bits == 4 else 128, 128)
    us = sustained(lambda: kg.mixed_precision_gemm_launcher(out, x, wq, ws, None, md), SECONDS)
    nt = _C.load().conch_get_tuning(1) or None
    ghz, loop_us, loop_max = read_probe("conch_debug_probe_mixed", 4096)
    print(f"mixed int{bits} {str(dtype)[6:]:9s} {m}x{k}x{n}: {us:7.1f} us/launch  clock {ghz:.2f} GHz  K loop {loop_us:.1f} us median "
          f"({loop_max:.1f} max)  ({2.0 * m * n * k / us / 1e6:.0f} TFLOP/s)", flush=True)
    mixed_stalls()


if __name__ == "__main__":
    if "--mixed-diag" in sys.argv:  # timing experiments of the mixed K loop (WRONG results): tuning key 7, see gemm_mixed.hip mixed_step
        names = {0: "full loop", 1: "no MFMAs", 2: "no fragment reads", 3: "no dequantisation (VALU + ds_write)", 4: "no LDS-DMA of X", 5: "no weight loads"}
        _C.set_tuning(1, 4)
        for diag in range(6):
            _C.set_tuning(7, diag)
            print(f"-- {names[diag]}")
            mixed_case(4096, 8192, 4096, torch.float16, 4)
        _C.set_tuning(7, 0)
        sys.exit(0)
    if "--mixed" in sys.argv:  # mixed kernel only, tile shapes given as a comma list (tuning key 1: 0 auto, 2..4, 5 = 512 x 128)
        for nt in (int(v) for v in sys.argv[sys.argv.index("--mixed") + 1].split(",")):
            _C.set_tuning(1, nt)
            print(f"-- CONCH_TUNE_MIXED_TILE_NT = {nt}")
            mixed_case(4096, 8192, 4096, torch.float16, 4)
            mixed_case(1024, 4096, 11008, torch.float16, 4)
        sys.exit(0)
    if "--persist" in sys.argv:  # tile-boundary study: plain launch against persistent workgroups
        for mode in (1, 2):
            _C.set_tuning(5, mode)
            print(f"-- CONCH_TUNE_PERSISTENT = {mode}")
            scaled_case(4096, 4096, 11008, torch.float8_e4m3fn, 8138.0)
        sys.exit(0)
    if "--epi" in sys.argv:  # CONCH_TUNE_EPILOGUE for every case below (1 = direct stores, 2 = row-major through LDS)
        _C.set_tuning(6, int(sys.argv[sys.argv.index("--epi") + 1]))
    if "--sched" in sys.argv:  # tile-schedule study of the headline shape only
        for sched in (int(v) for v in sys.argv[sys.argv.index("--sched") + 1].split(",")):
            _C.set_tuning(4, sched)
            print(f"-- CONCH_TUNE_TILE_SCHEDULE = {sched}")
            scaled_case(4096, 4096, 11008, torch.float8_e4m3fn, 8138.0)
        sys.exit(0)
    # dense peak per CU per clock: 5e15 / 256 / 2.4e9 (fp8, int8), half that for fp16 / bf16
    scaled_case(4096, 4096, 11008, torch.float8_e4m3fn, 8138.0)
    if "--zeros" in sys.argv:
        scaled_case(4096, 4096, 11008, torch.float8_e4m3fn, 8138.0, zeros=True)
        scaled_case(8192, 8192, 8192, torch.float8_e4m3fn, 8138.0, zeros=True)
    scaled_case(8192, 8192, 8192, torch.float8_e4m3fn, 8138.0)
    scaled_case(4096, 4096, 11008, torch.int8, 8138.0)
    mixed_case(1024, 4096, 11008, torch.float16, 4)
    mixed_case(4096, 8192, 4096, torch.float16, 4)
    mixed_case(8192, 8192, 8192, torch.float16, 4)
