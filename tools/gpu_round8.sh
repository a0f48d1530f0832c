#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r02h}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_bnb.py -x -q -m gpu -k "gelu or dequantize_blockwise or matmul_4bit or golden" > "$OUT/t_new.log" 2>&1; echo "new rc=$?" >> "$OUT/status.txt"
for w in c3 c2 c4 c1; do
  timeout 900 bash tools/gpu_profile.sh r02_$w --workload $w
  cp "$ROOT/gpurun_out/prof_r02_$w/summary.txt" "$OUT/${w}_rocprofv3_summary.txt"
  tail -n 1 "$ROOT/gpurun_out/prof_r02_$w/trace.log" > "$OUT/${w}_trace_bench.json"
done
timeout 600 python bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
timeout 300 python bench.py --workload c2 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
timeout 300 python bench.py --workload c4 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
timeout 300 python bench.py --workload c1 > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err"
tail -n 4 "$OUT/t_new.log"; cat "$OUT/status.txt"; head -n 6 "$OUT/c3_rocprofv3_summary.txt"; cat "$OUT/bench_c3.json"
