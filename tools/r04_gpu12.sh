#!/bin/bash
# round 4: the equilibration evaluated on one waypoint's blocks (scaling < 0) against the full Ruiz passes
O=gpurun_out/r04l; mkdir -p $O
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
L=path_optimizer_2_amd/csrc/libpqp_hip.so
{
for args in "--steps 400" "--steps 400 --inflight 1" "--config 3 --steps 60" "--config 2 --steps 40" "--batch 512 --n 200 --steps 150"; do
  bash tools/ab.sh "$args" $L
  bash tools/ab.sh "$args --scaling -4" $L
done
} 2>&1 | grep -v "$F" | tee $O/nominal_scaling_ab.txt
