#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-r02e}"
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "row_major_epilogue or modes or world1" > "$OUT/t_new.log" 2>&1; echo "new rc=$?" >> "$OUT/status.txt"
for w in c3 c3i8 sq8k c5shard 4096x8192x4096:int8 2048x4096x11008:fp8; do
  timeout 300 python tools/ab_tuning.py $w epi=1,2 11 >> "$OUT/ab_epi.log" 2>&1
done
python - > "$OUT/probe_epi.log" 2>&1 <<'PY'
import subprocess, sys
for epi in (1, 2):
    print(f"-- CONCH_TUNE_EPILOGUE = {epi}", flush=True)
    subprocess.run([sys.executable, "-c", f"import sys; sys.argv=['clock_probe.py','2.0','--sched','1','--classes'];\nimport runpy\nfrom conch_amd import _C, _build\n_C.LIB_PATH=_build.PROBE_LIB\n_C.set_tuning(6,{epi})\nrunpy.run_path('tools/clock_probe.py', run_name='__main__')"])
PY
grep -v amdgpu.ids "$OUT/ab_epi.log" "$OUT/probe_epi.log"; tail -n 5 "$OUT/t_new.log"; cat "$OUT/status.txt"
