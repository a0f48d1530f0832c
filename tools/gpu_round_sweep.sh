#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python tools/vendor_compare.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/vendor_compare.txt
