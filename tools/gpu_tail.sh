#!/bin/bash
# last pass of a round after a late kernel change: full GPU suite, default bench, one workload's profile -> gpurun_out/$1
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-tail}"
mkdir -p "$OUT"; cd "$ROOT"
( time timeout 2700 python -m pytest tests -x -q -m gpu ) > "$OUT/t_all.log" 2>&1; echo "pytest rc=$?" > "$OUT/status.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/status.txt"
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" >> "$OUT/status.txt"
for w in ${2:-c2}; do
  timeout 600 python bench.py --workload $w --steps 200 --warmup 50 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
  timeout 900 tools/gpu_profile.sh "${1:-tail}_$w" --workload $w > /dev/null 2>&1
done
grep -a "passed\|failed" "$OUT/t_all.log" | tail -1; cat "$OUT/status.txt"; head -c 400 "$OUT/bench_default.json"
