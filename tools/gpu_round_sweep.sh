#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee gpurun_out/full_tests.txt
