#!/bin/bash
O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for rep in 1 2; do
for fm in 2 1 0; do
  echo "PQP_CHAIN_FENCE=$fm"; PQP_CHAIN_FENCE=$fm timeout 120 python tools/bench_full_chain.py 1024 8 40 --exact-smoothers --moving --graph 2>&1 | grep "scenarios/s" | cut -c1-130
done
echo "plain"; timeout 120 python tools/bench_full_chain.py 1024 8 40 --exact-smoothers --moving 2>&1 | grep "scenarios/s" | cut -c1-130
done 2>&1 | grep -v "$F" | tee $O/chain_fence.txt
