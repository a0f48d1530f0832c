#!/bin/bash
# two processes hammering the assembly kernel on ONE GPU at the same time
for round in 1 2 3; do
  timeout 120 python tools/try_asm1w.py --time-only --rounds 3 > gpurun_out/stress_a_$round.log 2>&1 &
  PA=$!
  timeout 120 python tools/try_asm1w.py --time-only --rounds 3 > gpurun_out/stress_b_$round.log 2>&1 &
  PB=$!
  wait $PA; RA=$?
  wait $PB; RB=$?
  echo "round $round: rc $RA $RB"
done
tail -3 gpurun_out/stress_a_3.log
