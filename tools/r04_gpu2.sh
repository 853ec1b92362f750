#!/bin/bash
# round 4, second GPU call: the chain as a hipGraph (test + bench), the exact smoother kernels at 500 points / layers, the shim tests
set -x
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_smoothers.py tests/test_cpp_shim.py -m gpu -x -q > $O/pytest_subset.log 2>&1; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" $O/pytest_subset.log | tail -15
for flags in "--exact-smoothers --moving" "--exact-smoothers --moving --graph" "--exact-smoothers" "--exact-smoothers --graph" "--exact-smoothers --carry" "--exact-smoothers --carry --graph" "--exact-smoothers --inflight-2 --moving" "--exact-smoothers --inflight-2 --moving --graph"; do
  echo "$flags" >> $O/full_chain.txt
  timeout 300 python tools/bench_full_chain.py 1024 8 30 $flags 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" >> $O/full_chain.txt
done
cat $O/full_chain.txt | grep "scenarios/s\|^--" | cut -c1-120
