#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python tools/time_fused_quant.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fused_quant_times.txt
