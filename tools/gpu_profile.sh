#!/bin/bash
# rocprofv3 passes for one bench workload (run on the GPU box through gpurun).
# usage: [CONCH_COMMIT=<short id>] tools/gpu_profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
# With `--workload W` among the bench args the counter passes are turned into gpurun_out/prof_<tag>/traffic_entry.json
# (tools/write_traffic.py entry); `python tools/write_traffic.py merge gpurun_out/prof_*/traffic_entry.json` in the build container
# then writes profiles/traffic.json (the GPU box has no .git: the commit id travels in $CONCH_COMMIT).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="$1"; shift
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# kernel-trace pass: the bench's sustained leg (>= 2 s of back-to-back launches) dominates the per-kernel average, so the
# summary's avg_ns is the steady-state launch duration bench.py reports as kernel_sustained_ms / frac_sustained
TRACE=("$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-cold --no-probe --no-c5-base --no-side-legs --no-peak "$@")
# counter passes: few launches (counter collection serialises dispatches)
BENCH=("$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --quick "$@")
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "${TRACE[@]}" > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$OUT/pmc_sq" -- python3 "${BENCH[@]}" > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 "${BENCH[@]}" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 "${BENCH[@]}" > "$OUT/pmc_write.log" 2>&1
python3 "$ROOT/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
WL=c3
prev=""
for arg in "$@"; do if [ "$prev" = "--workload" ]; then WL="$arg"; fi; prev="$arg"; done
python3 "$ROOT/tools/write_traffic.py" entry "$WL" "$OUT" --commit "${CONCH_COMMIT:-unknown}" > "$OUT/traffic_entry.log" 2>&1
# keep only the small summaries
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*.csv" -size +1M -delete
