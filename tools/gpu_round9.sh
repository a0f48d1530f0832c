#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r02i}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "prepack or gelu or mixed" > "$OUT/t_new.log" 2>&1; echo "new rc=$?" >> "$OUT/status.txt"
timeout 600 python tools/ab_prepack.py > "$OUT/ab_prepack.log" 2>&1
tail -n 5 "$OUT/t_new.log"; cat "$OUT/status.txt"; grep -v amdgpu.ids "$OUT/ab_prepack.log"
