#!/bin/bash
# `bench.py --gpus 2` (both ranks on cuda:0, gloo) N times in a row -- half of them beside a process that keeps
# the GPU busy with the assembly kernel -- counting runs that do not end with rc 0 and one JSON line.  The launcher's own
# timeout (60 s) bounds every run; a stalled run prints its ranks' stacks into the log.
# usage: tools/two_rank_loop.sh [runs=50] [outdir=gpurun_out/two_rank_loop]
N=${1:-50}
OUT=${2:-gpurun_out/two_rank_loop}
mkdir -p "$OUT"
bad=0
t0=$(date +%s)
for i in $(seq 1 "$N"); do
  extra=""
  if [ $((i % 2)) -eq 0 ]; then extra="--c5-direct"; fi
  hammer=0
  if [ "$i" -gt $((N / 2)) ]; then
    timeout 60 python tools/try_asm1w.py --time-only --rounds 2 > "$OUT/hammer_$i.log" 2>&1 &
    hammer=$!
  fi
  python bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --all-ranks-on-device0 --c5-shape 2048,1024,4096 --launch-timeout 60 $extra \
    > "$OUT/run_$i.json" 2> "$OUT/run_$i.err"
  rc=$?
  lines=$(grep -c '^{' "$OUT/run_$i.json")
  if [ "$hammer" -ne 0 ]; then wait "$hammer"; fi
  if [ "$rc" -ne 0 ] || [ "$lines" -ne 1 ]; then
    bad=$((bad + 1))
    echo "run $i: rc $rc, $lines JSON line(s) $extra"
    tail -40 "$OUT/run_$i.err"
  else
    rm -f "$OUT/run_$i.err" "$OUT/hammer_$i.log"
  fi
done
echo "two-rank bench runs: $N, stalled or failed: $bad, wall $(( $(date +%s) - t0 )) s (runs $((N / 2 + 1))..$N beside a second process loading the GPU)"
