"""Build recipe for libconch_amd.so (hipcc, gfx950 only, in-tree output)."""

from __future__ import annotations

import os
import subprocess
import sys
import threading
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libconch_amd.so"
SOURCES = ["capi.hip", "gemm_asm.hip", "quant.hip", "quant_dynamic.hip", "gemm_generic.hip", "repack.hip", "gemm_mfma.hip", "gemm_mid.hip", "gemm_skinny.hip", "gemm_mixed.hip", "gemm_mixed_strip.hip", "gemm_mixed_skinny.hip", "gemm_modes.hip", "bnb.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-fast-math",
    "-ffp-contract=off",  # the epilogues rely on separately rounded multiplies (bit parity)
    f"-I{ROOT / 'include'}",
    f"-I{CSRC}",
]


# every hipcc process of this interpreter -- the product build, the diagnostic twin and experiment variants may run at the same
# time (__graft_entry__.build) -- takes a slot here first
JOBS = max(1, int(os.environ.get("CONCH_BUILD_JOBS", "4")))
_COMPILE_SLOTS = threading.BoundedSemaphore(JOBS)


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


PROBE_LIB = PKG / "libconch_amd_probe.so"
def _llvm_bin() -> Path:
    """Directory of the ROCm LLVM tools that assemble and link the hand-written kernels (clang, ld.lld): $CONCH_LLVM_BIN, the usual
    place under the ROCm root, else wherever hipcc's own clang lives (`hipcc --print-prog-name=clang`: hipcc is a clang driver)."""
    env = os.environ.get("CONCH_LLVM_BIN")
    if env:
        return Path(env)
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root and (Path(root) / "lib" / "llvm" / "bin" / "clang").exists():
            return Path(root) / "lib" / "llvm" / "bin"
    try:
        out = subprocess.run([HIPCC, "--print-prog-name=clang"], capture_output=True, text=True, check=True, timeout=60).stdout.strip()
        if out and Path(out).exists():
            return Path(out).resolve().parent
    except (OSError, subprocess.SubprocessError):
        pass
    return Path("/opt/rocm/lib/llvm/bin")


LLVM_BIN = _llvm_bin()
ASM_GENERATORS = {"gemm1w": CSRC / "asm" / "gen_gemm1w.py", "mixed1w": CSRC / "asm" / "gen_mixed1w.py"}  # name -> script that writes NAME.s (hand-allocated gfx950 assembly)


def build_asm(objdir: Path, force: bool = False, verbose: bool = False) -> list[Path]:
    """The hand-written assembly kernels: generator script -> NAME.s -> (clang -x assembler, gfx950) -> NAME.o -> (ld.lld
    -shared) -> NAME.hsaco -> NAME_hsaco.inc, a comma-separated byte list that csrc/gemm_asm.hip embeds.  Returns the .inc
    files (dependencies of that translation unit)."""
    incs = []
    for name, script in ASM_GENERATORS.items():
        inc = objdir / f"{name}_hsaco.inc"
        incs.append(inc)
        if not (force or _stale(inc, [script, Path(__file__)])):
            continue
        src, obj, hsaco = objdir / f"{name}.s", objdir / f"{name}.o", objdir / f"{name}.hsaco"
        cmds = [[sys.executable, str(script), str(src)],
                [str(LLVM_BIN / "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(obj)],
                [str(LLVM_BIN / "ld.lld"), "-shared", str(obj), "-o", str(hsaco)]]
        for cmd in cmds:
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        data = hsaco.read_bytes()
        inc.write_text(",\n".join(",".join(str(b) for b in data[i:i + 32]) for i in range(0, len(data), 32)) + "\n")
    return incs


def build(force: bool = False, verbose: bool = False, probe: bool = False, variant: str | None = None, defines: tuple[str, ...] = (),
          only: tuple[str, ...] = ()) -> Path:
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path.

    `probe=True` builds the DIAGNOSTIC twin libconch_amd_probe.so (-DCONCH_CLOCK_PROBE: in-kernel clock stamps
    around the GEMM K loops, read by tools/clock_probe.py); the product library never contains them.
    `variant="x", defines=("-DFOO",)` builds libconch_amd_x.so with extra macros, for interleaved A/B runs of an experiment
    against the product library in one process (tools/ab_lib.py); never loaded by the package itself.  `only=("gemm_mfma.hip",)`
    recompiles just those sources with the macros and links the product build's objects for the rest (the macros of an
    experiment live in one or two files; a full rebuild per variant costs minutes).
    """
    headers = sorted(CSRC.glob("*.hpp")) + [ROOT / "include" / "conch_amd.h", Path(__file__)]
    objdir = PKG / ("build_probe" if probe else "build")
    flags = [*FLAGS, "-DCONCH_CLOCK_PROBE"] if probe else FLAGS
    lib = PROBE_LIB if probe else LIB
    if variant:
        objdir = PKG / f"build_{variant}"
        flags = [*FLAGS, *defines]
        lib = PKG / f"libconch_amd_{variant}.so"
    objdir.mkdir(exist_ok=True)
    asm_incs = build_asm(objdir, force=force, verbose=verbose)

    def compile_one(src: str) -> Path:
        s = CSRC / src
        if variant and only and src not in only:
            return PKG / "build" / (s.stem + ".o")  # the product build's object (built below if missing)
        o = objdir / (s.stem + ".o")
        deps = [s, *headers, *(asm_incs if src == "gemm_asm.hip" else [])]
        if force or _stale(o, deps):
            cmd = [HIPCC, *flags, f"-I{objdir}", "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            with _COMPILE_SLOTS:
                subprocess.run(cmd, check=True)
        return o

    if variant and only:
        build(verbose=verbose)  # the objects the variant borrows
    with ThreadPoolExecutor(max_workers=min(JOBS, len(SOURCES))) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    if force or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return lib


MICRO_LIB = PKG / "libconch_micro.so"


def build_micro(force: bool = False, verbose: bool = False) -> Path:
    """libconch_micro.so: the achievable-peak microbenchmark bench.py reports beside the datasheet peak
    (csrc_diag/micro_peak.hip; a measurement helper, never on a product path)."""
    src = PKG / "csrc_diag" / "micro_peak.hip"
    if force or _stale(MICRO_LIB, [src, Path(__file__)]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", str(src), "-o", str(MICRO_LIB)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        with _COMPILE_SLOTS:
            subprocess.run(cmd, check=True)
    return MICRO_LIB


HOST_SHIM = PKG / "_conch_host.so"


def build_host_shim(force: bool = False, verbose: bool = False) -> Path:
    """Compile conch_amd/csrc_host/host_shim.cpp (g++, against this interpreter's torch): the C++ host path of the hot ops
    (checks + allocation + C-ABI call; kernels/quantization/_fast.py uses it when it is there).  ~30 s."""
    import sysconfig

    import pybind11
    from torch.utils import cpp_extension as ce

    src = PKG / "csrc_host" / "host_shim.cpp"
    import torch

    # stale when the shim's source, the C ABI it is typed from, the library it binds or the torch it links changed
    deps = [src, Path(__file__), ROOT / "include" / "conch_amd.h", Path(torch.__file__)]
    if LIB.exists():
        deps.append(LIB)
    if not (force or _stale(HOST_SHIM, deps)):
        return HOST_SHIM
    inc = [str(ROOT / "include"), *ce.include_paths(), pybind11.get_include(), sysconfig.get_paths()["include"], "/opt/rocm/include"]
    libdir = ce.library_paths()[0]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           "-DTORCH_EXTENSION_NAME=_conch_host", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", str(src), "-o", str(HOST_SHIM),
           *[f"-I{i}" for i in inc], f"-L{libdir}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ltorch_python",
           f"-Wl,-rpath,{libdir}", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    with _COMPILE_SLOTS:
        subprocess.run(cmd, check=True)
    return HOST_SHIM


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m conch_amd._build --variant NAME -DMACRO ...
        name = sys.argv[sys.argv.index("--variant") + 1]
        only = tuple(sys.argv[sys.argv.index("--only") + 1].split(",")) if "--only" in sys.argv else ()
        print(build(force="--force" in sys.argv, verbose=True, variant=name, defines=tuple(a for a in sys.argv if a.startswith("-D")), only=only))
        sys.exit(0)
    # both libraries by default: the diagnostic twin must export the same C ABI as the product library (bench.py opens it
    # through the same ctypes declarations); --probe / --product build one of them only
    if "--product" not in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, probe=True))
    if "--probe" not in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, probe=False))
        print(build_host_shim(force="--force" in sys.argv, verbose=True))
        print(build_micro(force="--force" in sys.argv, verbose=True))
