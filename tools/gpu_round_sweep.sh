#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}/benchmarks"
OUT=../gpurun_out/benchmarks_cli.txt
mkdir -p ../gpurun_out; : > $OUT
run() { echo "\$ $*" >> $OUT; timeout 300 "$@" 2>&1 | grep -v "amdgpu.ids" >> $OUT; echo >> $OUT; }
T="--iteration-time-ms 1500 --warmup-time-ms 300"
run python scaled_gemm_benchmark.py --input-dtype fp8 --m-dim 4096 --k-dim 4096 --n-dim 11008 $T
CONCH_BENCH_ENABLE_ALL_REF=1 run python scaled_gemm_benchmark.py --input-dtype fp8 --m-dim 4096 --k-dim 4096 --n-dim 11008 $T
run python scaled_gemm_benchmark.py --input-dtype int8 --m-dim 128 --k-dim 4096 --n-dim 4096 $T
run python scaled_gemm_benchmark.py --input-dtype int8 --m-dim 16 --k-dim 11008 --n-dim 4096 $T
run python mixed_precision_gemm_benchmark.py --m-dim 1024 --k-dim 4096 --n-dim 11008 $T
run python mixed_precision_gemm_benchmark.py --m-dim 1024 --k-dim 4096 --n-dim 11008 --prepack $T
run python mixed_precision_gemm_benchmark.py $T
run python mixed_precision_gemm_benchmark.py --prepack $T
run python mixed_precision_gemm_benchmark.py --m-dim 1 --k-dim 11008 --n-dim 4096 $T
run python static_scaled_int8_quant_benchmark.py $T
run python static_scaled_int8_quant_benchmark.py --dynamic $T
run python static_scaled_fp8_quant_benchmark.py $T
run python bnb_quantize_blockwise_benchmark.py $T
run python bnb_dequantize_blockwise_benchmark.py $T
tail -n 120 $OUT
