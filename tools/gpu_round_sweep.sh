#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_graph.py -x -q -m gpu -k "one_launch or splitk or every_kernel_variant or decode or graph or down_projection or static_quant or c2_config or fused" 2>&1 | tail -5 | tee gpurun_out/ol_tests.txt
