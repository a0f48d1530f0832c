#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 tools/gpu_profile.sh r02i_c3 --workload c3 > /dev/null 2>&1
head -3 gpurun_out/prof_r02i_c3/summary.txt
tail -c 1200 gpurun_out/prof_r02i_c3/trace.log | grep -o '"kernel_sustained_ms": [0-9.]*'
