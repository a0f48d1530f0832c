#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r02g}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/t_all.log" 2>&1; echo "all rc=$?" >> "$OUT/status.txt"
timeout 600 python tools/time_variants.py > "$OUT/time_variants.log" 2>&1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --all-ranks-on-device0 --steps 3 --warmup 1 > "$OUT/bench_2rank_gloo.json" 2> "$OUT/bench_2rank_gloo.err"
timeout 300 python bench.py --workload c5 --steps 10 --warmup 3 > "$OUT/bench_c5_1gpu.json" 2> "$OUT/bench_c5_1gpu.err"
tail -n 5 "$OUT/t_all.log"; cat "$OUT/status.txt"; grep -v amdgpu.ids "$OUT/time_variants.log"; cat "$OUT/bench_2rank_gloo.json"; tail -n 5 "$OUT/bench_2rank_gloo.err"; cat "$OUT/bench_c5_1gpu.json"
