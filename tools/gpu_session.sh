mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "mixed_row_major" -n 2 2>&1 | tail -2
{
python tools/ab_mixed_tuning.py c4 6=1,2 9
python tools/ab_mixed_tuning.py readme 6=1,2 9
python tools/ab_mixed_tuning.py sq8k 6=1,2 7
python tools/ab_mixed_tuning.py 4096x4096x11008 6=1,2 7
python tools/ab_mixed_tuning.py readme 6=1,2 7 --nt 5
for w in c3 sq8k c5shard; do python tools/ab_tuning.py $w epi=1,2 7; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/ab8_mixed_epi.log
