python -m pytest tests/test_gpu_gemm.py -q -m gpu -n 2 -x 2>&1 | tail -2
{
echo "### entry stamp BEFORE the setup arithmetic"; python tools/clock_probe.py 2 --sched 1 --classes
echo "### entry stamp AFTER the setup arithmetic"; CONCH_PROBE_LIB=conch_amd/libconch_amd_probeas.so python tools/clock_probe.py 2 --sched 1 --classes
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/probe_setup2.log
