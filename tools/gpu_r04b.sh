cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "strip" > $OUT/t_strip.log 2>&1; echo "strip rc=$?" > $OUT/status.txt
timeout 600 python tools/ab_mixed_strip.py 5 > $OUT/ab_mixed_strip.txt 2>&1
( time timeout 2700 python -m pytest tests -q -m gpu ) > $OUT/t_all.log 2>&1; echo "pytest rc=$?" >> $OUT/status.txt
cat $OUT/status.txt; tail -n 5 $OUT/t_strip.log; cat $OUT/ab_mixed_strip.txt; tail -n 12 $OUT/t_all.log
