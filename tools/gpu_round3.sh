#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r02c}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "persistent or modes or world1" > "$OUT/t_new.log" 2>&1; echo "new rc=$?" >> "$OUT/status.txt"
for w in c3 c3i8 sq8k c5shard 4096x8192x4096:int8 2048x4096x11008:fp8; do
  timeout 300 python tools/ab_tuning.py $w persist=1,2 11 >> "$OUT/ab_persist.log" 2>&1
done
timeout 300 python tools/clock_probe.py 2.0 --sched 1 --classes --timeline > "$OUT/probe_persist.log" 2>&1
grep -v amdgpu.ids "$OUT/ab_persist.log" "$OUT/probe_persist.log"; tail -n 5 "$OUT/t_new.log"; cat "$OUT/status.txt"
