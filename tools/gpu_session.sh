mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_gemm.py -q -m gpu -n 2 -k "static_quant_scaled_gemm" 2>&1 | tail -2
python tools/time_fused_quant.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3a/fused_quant_times.txt
