#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "several_workgroups" 2>&1 | tail -5 | tee gpurun_out/decode_tests.txt
