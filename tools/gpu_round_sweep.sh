#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "decode or down_projection or ragged or (mixed and (fused or gelu or silu))" 2>&1 | tail -8 > gpurun_out/ragged_tests.txt
cat gpurun_out/ragged_tests.txt
timeout 1200 python tools/dispatch_cold_sweep.py "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/dispatch_cold_sweep.txt
grep -c "pick costs" gpurun_out/dispatch_cold_sweep.txt; grep "pick costs\| 192x\| 256x\| 384x" gpurun_out/dispatch_cold_sweep.txt | grep "mixed\|pick" | head -60
