#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "split_k" 2>&1 | tail -6 > gpurun_out/splitk_tests.txt
cat gpurun_out/splitk_tests.txt
timeout 1200 python tools/dispatch_cold_sweep.py --mixed-mid 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee gpurun_out/mixed_mid_sweep.txt
