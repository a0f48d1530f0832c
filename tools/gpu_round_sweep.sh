#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python tools/time_bnb_decode.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bnb_decode_times.txt
