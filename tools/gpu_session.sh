mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "silu or gelu or mixed_row_major" -n 2 2>&1 | tail -3
python tools/time_fused_ffn.py 7 2>&1 | grep -v amdgpu | tee gpurun_out/r3a/ab10_fused_ffn.log
