#!/bin/bash
# round 4, third GPU call: chain graph + smoother tests again, the stream kernel after the reciprocal diet
set -x
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_smoothers.py tests/test_gpu_stream.py -m gpu -x -q > $O/pytest_subset.log 2>&1; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" $O/pytest_subset.log | tail -8
timeout 600 python tools/bench_stream.py --batches 8192,16384,32768,65536 --steps 4 --oracle 32 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" > $O/bench_stream.txt; cat $O/bench_stream.txt
