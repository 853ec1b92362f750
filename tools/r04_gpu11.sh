#!/bin/bash
# round 4: lean kernel - GPU equality test against the general kernel, robustness sweep without equilibration, seed sweep of the headline
O=gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lean or long_paths" 2>&1 | grep -v "$F" | tail -8
timeout 900 python tools/robustness_sweep.py 8 8192 1000 0 2>&1 | grep -v "$F" > $O/robustness_scaling0.txt; cat $O/robustness_scaling0.txt
timeout 900 python tools/robustness_sweep.py 8 8192 1000 2>&1 | grep -v "$F" > $O/robustness_scaling4.txt; cat $O/robustness_scaling4.txt
L=path_optimizer_2_amd/csrc/libpqp_hip.so
for seed in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  bash tools/ab.sh "--steps 300 --seed $seed" $L
  bash tools/ab.sh "--steps 300 --seed $seed --scaling 0" $L
done 2>&1 | grep -v "$F" | tee $O/seed_sweep_lean.txt
