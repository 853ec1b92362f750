#!/bin/bash
# round 4, seventh GPU call: the two-waypoints-per-lane context (PQP_PAIR=1): parity tests + bench A/B
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
PQP_PAIR=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -15 > $O/pytest_pair.log; tail -6 $O/pytest_pair.log
L=path_optimizer_2_amd/csrc/libpqp_hip.so
{
for args in "--steps 200" "--steps 200 --inflight 1" "--config 3 --steps 40" "--config 2 --steps 30" "--batch 512 --n 200 --steps 100"; do
  PQP_PAIR=0 bash tools/ab.sh "$args" $L
  PQP_PAIR=1 bash tools/ab.sh "$args" $L
done
} 2>&1 | grep -v "$F" | tee $O/pair_ab.txt
