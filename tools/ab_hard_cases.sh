#!/bin/bash
# Rare extreme QPs under variants of the rounds' policy: tools/ab_hard_cases.sh "<variant names under build_variants/>" [seeds=4]
for v in $1; do lib=$PWD/build_variants/libpqp_$v.so; [ $v = base ] && lib=$PWD/path_optimizer_2_amd/csrc/libpqp_hip.so; echo "== $v"
  PQP_LIB=$lib timeout 100 python tools/stragglers.py 2048 300 varied seed=1004 2>&1 | grep -v amdgpu | head -2 | tail -1
  PQP_LIB=$lib timeout 100 python tools/stragglers.py 2048 200 uniform seed=1002 2>&1 | grep -v amdgpu | head -2 | tail -1
  PQP_LIB=$lib timeout 300 python tools/robustness_sweep.py ${2:-4} 8192 1000 2>&1 | grep -v amdgpu | grep "^n " | cut -c1-150
done
