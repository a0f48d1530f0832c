#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "split_k_on_every" 2>&1 | tail -12 | tee gpurun_out/sk_tests.txt
