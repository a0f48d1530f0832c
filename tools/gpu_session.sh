mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_gemm.py -q -m gpu -n 2 -x -k "scaled" 2>&1 | tail -2
{
for w in c3 sq8k c5shard c3i8; do python tools/ab_scaled_lib.py $w plainrsrc 9; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/ab11_waterfall.log
