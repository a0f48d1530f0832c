#!/bin/bash
# end-of-round pass at HEAD, everything on ONE box (so that the kept rocprofv3 summaries and the bench line come from the same
# part): full GPU suite, smoke, default bench, per-workload bench lines, rocprofv3 profiles (C1-C4 + the README shape) whose
# counter passes also write gpurun_out/prof_<tag>_<w>/traffic_entry.json (merge them with tools/write_traffic.py merge).
# usage: [CONCH_COMMIT=<short id>] tools/gpu_final.sh <tag>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-final}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 -p no:cacheprovider > "$OUT/t_all.log" 2>&1; echo "pytest rc=$?" > "$OUT/status.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/status.txt"
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?" >> "$OUT/status.txt"
for w in c1 c2 c4; do
  timeout 600 python bench.py --workload $w --steps 200 --warmup 50 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
timeout 600 python tools/time_variants.py > "$OUT/time_variants.log" 2>&1
for w in c3 c4 c2 c1 c4readme; do
  timeout 900 tools/gpu_profile.sh "${1:-final}_$w" --workload $w > /dev/null 2>&1
done
tail -n 4 "$OUT/t_all.log"; cat "$OUT/status.txt"; tail -n 2 "$OUT/smoke.log"; cat "$OUT/bench_default.json"
