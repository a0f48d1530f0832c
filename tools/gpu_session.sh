mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -x -q -m gpu -n 2 > gpurun_out/r3a/t_all.log 2>&1; tail -3 gpurun_out/r3a/t_all.log
python bench.py > gpurun_out/r3a/bench_c3.json 2> gpurun_out/r3a/bench_c3.err; cat gpurun_out/r3a/bench_c3.json
