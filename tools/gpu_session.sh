mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_gemm.py -q -m gpu -n 2 -x -k "persistent" 2>&1 | tail -2
{
python tools/ab_tuning.py c3 persist=1,2 9
python tools/ab_tuning.py sq8k persist=1,2 7
python tools/ab_tuning.py c5shard persist=1,2 7
python tools/ab_tuning.py c3i8 persist=1,2 7
echo "### probe persist"; python tools/clock_probe.py 2 --persist
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/ab14_persist.log
