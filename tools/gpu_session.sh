mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -x -q -m gpu -n 2 > gpurun_out/r3a/t_all2.log 2>&1; tail -3 gpurun_out/r3a/t_all2.log
{
for w in 256x4096x11008:int8 512x4096x4096:fp8 1024x4096x4096:fp8 128x4096x28672:int8; do
  python tools/ab_tuning.py $w epi=1,2 7 --variant 6
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/ab9_mid_epi.log
