#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1500 python tools/fuzz_dispatch.py "$@" 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/fuzz.txt
