#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rm -f gpurun_out/ab_lib.txt
for v in "$@"; do
  echo "== variant $v (ratios > 1: the product library is faster)" | tee -a gpurun_out/ab_lib.txt
  timeout 600 python tools/ab_lib.py $v 5 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab_lib.txt
done
