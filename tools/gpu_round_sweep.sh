#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/graph_tests.txt
cat gpurun_out/graph_tests.txt
