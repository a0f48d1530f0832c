#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
PYTHONPATH=. timeout 600 python tools/_tmp/prof_host.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prof_host.txt
