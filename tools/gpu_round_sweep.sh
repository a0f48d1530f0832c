#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_bnb.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/bnb_tests.txt
timeout 600 python tools/time_variants.py 2>&1 | grep -i "bnb q" | tee gpurun_out/bnb_times.txt
