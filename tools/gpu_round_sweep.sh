#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python tools/ab_lib.py vpin 5 --scaled 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_vpin.txt
