#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -14 | tee gpurun_out/durations.txt
