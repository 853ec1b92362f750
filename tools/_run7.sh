mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_corridor.py -x -q 2>&1 | tail -15 > gpurun_out/t_corr.log
timeout 300 python tools/bench_corridor.py 1024 80 8 > gpurun_out/corr_bench.log 2>&1
cat gpurun_out/t_corr.log; grep -v amdgpu gpurun_out/corr_bench.log | tail -5
