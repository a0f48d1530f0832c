cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-r04d}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "strip" 2>&1 | tail -3 > $OUT/t_strip.log
timeout 600 python tools/ab_mixed_strip.py 5 2>&1 | grep -v amdgpu.ids > $OUT/ab_mixed_strip.txt
timeout 600 python tools/ab_strip_variants.py sBASE sNOMFMA sNODQ sNOREAD sNOXDMA sNOBAR 2>&1 | grep -v amdgpu.ids > $OUT/strip_diag.txt
cat $OUT/t_strip.log $OUT/ab_mixed_strip.txt $OUT/strip_diag.txt
