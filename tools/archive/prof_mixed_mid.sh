#!/bin/bash
# per-kernel time of mixed_precision_gemm at batched-decode sizes (strip kernel + slab reduce): rocprofv3 kernel trace, csv
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd /tmp; export TMPDIR=/tmp
for s in "64 4096 11008" "128 4096 11008" "64 8192 28672"; do
  d="$ROOT/gpurun_out/prof_mid_${s// /_}"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$ROOT/tools/archive/run_mixed_once.py" $s 2000 > /dev/null 2>&1
  echo "== $s"
  for f in "$d"/*/*kernel_stats.csv; do [ -f "$f" ] && head -4 "$f" | cut -c1-220; done
done
