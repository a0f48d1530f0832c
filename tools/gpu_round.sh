#!/bin/bash
# One gpurun session of round 2: new parity tests, A/B of the new kernels, headline bench.  Output -> gpurun_out/r02a/
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r02a}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_quant.py -x -q -m gpu -k "dynamic" > "$OUT/t_dyn.log" 2>&1; echo "dyn rc=$?" >> "$OUT/status.txt"
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "schedule or splitk or small_scale or c2_config" > "$OUT/t_new.log" 2>&1; echo "new rc=$?" >> "$OUT/status.txt"
timeout 300 python tools/ab_tuning.py c3 sched=1,2 > "$OUT/ab_c3.log" 2>&1
timeout 300 python tools/ab_tuning.py c3i8 sched=1,2 >> "$OUT/ab_c3.log" 2>&1
timeout 300 python tools/ab_tuning.py c5shard sched=1,2 >> "$OUT/ab_c3.log" 2>&1
timeout 300 python tools/ab_tuning.py sq8k sched=1,2 >> "$OUT/ab_c3.log" 2>&1
timeout 300 python tools/ab_tuning.py c2 skinny=1,2,3 25 > "$OUT/ab_c2.log" 2>&1
timeout 300 python tools/ab_tuning.py c2fp8 skinny=1,2,3 25 >> "$OUT/ab_c2.log" 2>&1
timeout 300 python tools/ab_tuning.py 32x8192x8192:int8 skinny=1,2,3 25 --variant 4 >> "$OUT/ab_c2.log" 2>&1
timeout 300 python tools/ab_tuning.py 64x4096x11008:fp8 skinny=1,2,3 25 --variant 4 >> "$OUT/ab_c2.log" 2>&1
timeout 600 python bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
timeout 300 python bench.py --workload c2 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/t_all.log" 2>&1; echo "all rc=$?" >> "$OUT/status.txt"
tail -3 "$OUT/t_dyn.log" "$OUT/t_new.log" "$OUT/t_all.log"; cat "$OUT/ab_c3.log" "$OUT/ab_c2.log" "$OUT/status.txt"; cat "$OUT/bench_c3.json" "$OUT/bench_c2.json"
