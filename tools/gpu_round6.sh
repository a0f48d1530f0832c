#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r02f}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_bnb.py -x -q -m gpu > "$OUT/t_bnb.log" 2>&1; echo "bnb rc=$?" >> "$OUT/status.txt"
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "modes or mixed or row_major or persistent" > "$OUT/t_new.log" 2>&1; echo "new rc=$?" >> "$OUT/status.txt"
tail -n 6 "$OUT/t_bnb.log" "$OUT/t_new.log"; cat "$OUT/status.txt"
