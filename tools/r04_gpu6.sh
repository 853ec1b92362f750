#!/bin/bash
# round 4, sixth GPU call: two wavefronts per SIMD for the lane-per-waypoint kernel, measured (VERDICT round 3 task 1)
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
L=path_optimizer_2_amd/csrc/libpqp_hip.so; V=build_variants
{
for args in "--steps 400" "--steps 400 --inflight 1" "--config 3 --steps 60"; do
  bash tools/ab.sh "$args" $L $V/libpqp_occ1g.so $V/libpqp_occ2.so $V/libpqp_occ2g.so
done
echo "== Ruiz passes 0 / 2 (production default 4)"
bash tools/ab.sh "--steps 400 --scaling 0" $L
bash tools/ab.sh "--steps 400 --scaling 2" $L
} 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tee $O/occupancy2_ab.txt
