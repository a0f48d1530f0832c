#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r02b}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/ab_tuning.py c3 sched=1,2,3 11 > "$OUT/ab_c3.log" 2>&1
timeout 300 python tools/clock_probe.py 2.0 --sched 1,2,3 --classes --timeline > "$OUT/probe_sched.log" 2>&1
# split-K launch forms over decode shapes
for shape in 16x4096x4096 32x4096x4096 64x4096x4096 128x4096x4096 256x4096x4096 16x4096x11008 32x4096x11008 128x4096x11008 16x8192x8192 64x8192x8192 128x8192x8192 32x4096x28672 128x8192x28672 256x8192x3584; do
  timeout 120 python tools/ab_tuning.py ${shape}:int8 skinny=1,3 11 --variant 4 >> "$OUT/splitk_modes.log" 2>&1
done
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "schedule or splitk or c2_config or every_kernel" > "$OUT/t_new.log" 2>&1; echo "new rc=$?" >> "$OUT/status.txt"
timeout 300 python bench.py --workload c2 --no-cpu-baseline > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
grep -v amdgpu.ids "$OUT/ab_c3.log" "$OUT/probe_sched.log" "$OUT/splitk_modes.log"; tail -n 3 "$OUT/t_new.log"; cat "$OUT/status.txt" "$OUT/bench_c2.json"
