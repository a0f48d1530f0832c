#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_graph.py -x -q -m gpu -k "decode or graph" 2>&1 | tail -5 > gpurun_out/decode_tests.txt
cat gpurun_out/decode_tests.txt
timeout 600 python tools/time_decode_modes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mixed_decode_one_launch.txt
