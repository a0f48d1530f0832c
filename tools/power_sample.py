"""Board power and clocks as the SMI tools report them while C3 loops on (a) the benchmark's data, (b) all-zero operands -- the direct
reading that goes with tools/c3_data_dependence.py's in-kernel clock.  The parent never touches the GPU: it starts one child per data
pattern (the child loops `conch_time_scaled_gemm` for --seconds) and polls `rocm-smi` / `amd-smi` from outside.
-> stdout (profiles/r04/c3_power_sample.txt)"""
import argparse
import ctypes
import re
import shutil
import statistics
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
M, K, N = 4096, 4096, 11008


def child(pattern: str, seconds: float) -> None:
    import torch

    sys.path.insert(0, str(ROOT))
    from conch_amd import _C
    from conch_amd.kernels.quantization import gemm as kg
    from conch_amd.ops.quantization.gemm import create_scaled_metadata

    torch.manual_seed(0)
    f8 = torch.float8_e4m3fn
    if pattern.startswith("vendor"):  # torch.matmul (hipBLASLt / rocBLAS) on fp16 weights dequantised beforehand, README shape
        m, k, n = 4096, 8192, 4096
        x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
        if pattern == "vendor_zero":
            x.zero_()
        w = ((torch.randint(0, 16, (k, n), device="cuda") - 8).float() * (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).repeat_interleave(128, 0)).to(torch.float16)
        out = torch.empty((m, n), dtype=torch.float16, device="cuda")
        torch.matmul(x, w, out=out)
        torch.cuda.synchronize()
        print("READY", flush=True)
        t_end = time.perf_counter() + seconds
        last = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        while time.perf_counter() < t_end:
            e0.record()
            for _ in range(100):
                torch.matmul(x, w, out=out)
            e1.record()
            e1.synchronize()
            last.append(e0.elapsed_time(e1) / 100)
        tail = sorted(last[len(last) // 2:])
        print(f"LAUNCH_US {tail[len(tail) // 2] * 1e3:.1f}", flush=True)
        return
    if pattern.startswith("mixed"):  # the reference's README shape, int4 x fp16: the benchmark's data / all-zero activations
        from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata

        m, k, n = 4096, 8192, 4096
        x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
        if pattern == "mixed_zero":
            x.zero_()
        wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
        ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
        md = create_mixed_precision_metadata(x, wq, ws, None, 4, 8, 128)
        out = torch.empty((m, n), dtype=torch.float16, device="cuda")
        ms = ctypes.c_float()
        torch.cuda.synchronize()
        print("READY", flush=True)
        t_end = time.perf_counter() + seconds
        last = []
        while time.perf_counter() < t_end:
            _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (100, ctypes.byref(ms))), "time")
            last.append(ms.value)
        tail = sorted(last[len(last) // 2:])
        print(f"LAUNCH_US {tail[len(tail) // 2] * 1e3:.1f}", flush=True)
        return
    if pattern == "bench":
        a, bt = (0.25 * torch.rand((M, K), device="cuda")).to(f8), (0.25 * torch.rand((N, K), device="cuda")).to(f8)
    elif pattern == "zero":
        a, bt = torch.zeros((M, K), device="cuda").to(f8), torch.zeros((N, K), device="cuda").to(f8)
    elif pattern == "bytes":
        a = torch.randint(0, 0x7E, (M, K), device="cuda", dtype=torch.uint8)
        bt = torch.randint(0, 0x7E, (N, K), device="cuda", dtype=torch.uint8)
        a, bt = (a | (torch.randint(0, 2, (M, K), device="cuda", dtype=torch.uint8) << 7)).view(f8), (bt | (torch.randint(0, 2, (N, K), device="cuda", dtype=torch.uint8) << 7)).view(f8)
    else:  # idle: hold the context, launch nothing
        torch.cuda.synchronize()
        print("READY", flush=True)
        time.sleep(seconds)
        return
    sa, sb = 0.25 * torch.rand((M, 1), device="cuda"), 0.25 * torch.rand((N, 1), device="cuda")
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
    ms = ctypes.c_float()
    lib = _C.load()
    torch.cuda.synchronize()
    print("READY", flush=True)
    t_end = time.perf_counter() + seconds
    last = []
    while time.perf_counter() < t_end:
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (200, ctypes.byref(ms)), lib=lib), "time")
        last.append(ms.value)
    tail = sorted(last[len(last) // 2:])
    print(f"LAUNCH_US {tail[len(tail) // 2] * 1e3:.1f}", flush=True)


def smi_sample() -> str:
    """One line with whatever the SMI tools print for power / clocks (text kept verbatim; units are the tools')."""
    parts = []
    if shutil.which("rocm-smi"):
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showtemp"], capture_output=True, text=True, timeout=20)
        for line in r.stdout.splitlines():
            if re.search(r"Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)", line):
                parts.append(re.sub(r"\s+", " ", line.strip()))
    return " | ".join(parts)


def amd_smi_sample() -> str:
    if not shutil.which("amd-smi"):
        return ""
    r = subprocess.run(["amd-smi", "metric", "-p", "-c", "-g", "0"], capture_output=True, text=True, timeout=30)
    keep = [re.sub(r"\s+", " ", l.strip()) for l in r.stdout.splitlines() if re.search(r"SOCKET_POWER|THROTTLE|GFX_\d+:|CLK:|POWER", l)]
    return " ; ".join(keep[:14])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--patterns", default="idle,bench,zero,bytes,bench,mixed,mixed_zero,mixed")
    ap.add_argument("--raw", action="store_true", help="print every SMI sample")
    args = ap.parse_args()
    if args.child:
        child(args.child, args.seconds)
        return
    if shutil.which("amd-smi"):
        r = subprocess.run(["amd-smi", "static", "-l", "-g", "0"], capture_output=True, text=True, timeout=30)
        print("# amd-smi static -l:", " ; ".join(re.sub(r"\s+", " ", l.strip()) for l in r.stdout.splitlines() if re.search(r"POWER_LIMIT", l))[:300])
    print("# C3 (fp8 4096x4096x11008) looping back to back on: idle = context only; bench = the benchmark's data (0.25 x rand); zero = all-zero operands; "
          "bytes = full-range random bytes; mixed / mixed_zero = mixed_precision_gemm int4 x fp16 4096x8192x4096 on the benchmark's data / zero activations; launch = median time of a launch (200 per timing)")
    for pattern in args.patterns.split(","):
        p = subprocess.Popen([sys.executable, __file__, "--child", pattern, "--seconds", str(args.seconds)], stdout=subprocess.PIPE, text=True)
        assert p.stdout is not None
        while True:
            line = p.stdout.readline()
            if not line or line.startswith("READY"):
                break
        time.sleep(1.5)
        t0 = time.perf_counter()
        power, gfx, sclk, temp = [], [], [], []
        while p.poll() is None and time.perf_counter() - t0 < args.seconds - 2.5:
            r, a = smi_sample(), amd_smi_sample()
            if args.raw:
                print(f"  t+{time.perf_counter() - t0 + 1.5:4.1f}s  {r}\n           amd-smi: {a}", flush=True)
            power += [int(x) for x in re.findall(r"SOCKET_POWER: (\d+) W", a)]
            gfx += [int(x) for x in re.findall(r" CLK: (\d+) MHz", a)]
            sclk += [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", r)]
            temp += [float(x) for x in re.findall(r"junction\) \(C\): ([\d.]+)", r)]
        rest = p.stdout.read()
        p.wait()
        launch = [l.split()[1] for l in rest.splitlines() if l.startswith("LAUNCH_US")]
        med = lambda v: statistics.median(v) if v else float("nan")  # noqa: E731
        print(f"{pattern:10s}: socket power {min(power, default=0)}-{max(power, default=0)} W (median {med(power):.0f}, {len(power)} samples), "
              f"XCD clocks median {med(gfx):.0f} MHz (amd-smi), sclk {med(sclk):.0f} MHz (rocm-smi), junction {med(temp):.0f} C"
              + (f", launch {launch[0]} us" if launch else ""), flush=True)


if __name__ == "__main__":
    main()
