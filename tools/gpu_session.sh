mkdir -p gpurun_out/r3a
{
for w in c3 sq8k c5shard; do python tools/ab_scaled_lib.py $w aux0,aux2 9; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/ab13_aux_after_wf.log
