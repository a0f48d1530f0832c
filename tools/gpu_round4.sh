#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r02d}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/clock_probe.py 2.0 --persist --classes > "$OUT/probe_persist.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "modes or world1 or mixed" > "$OUT/t_new.log" 2>&1; echo "new rc=$?" >> "$OUT/status.txt"
grep -v amdgpu.ids "$OUT/probe_persist.log"; tail -n 5 "$OUT/t_new.log"; cat "$OUT/status.txt"
