#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1500 python tools/dispatch_cold_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/dispatch_cold_sweep.txt
grep -c "pick costs" gpurun_out/dispatch_cold_sweep.txt; grep "pick costs" gpurun_out/dispatch_cold_sweep.txt | cut -c1-300 | head -20; wc -l gpurun_out/dispatch_cold_sweep.txt
