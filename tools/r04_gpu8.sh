#!/bin/bash
# round 4, eighth GPU call: cold operations out of line (PQP_NO_MONOLITH) at one and two wavefronts per SIMD
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
L=path_optimizer_2_amd/csrc/libpqp_hip.so; V=build_variants
{
for args in "--steps 200" "--steps 200 --inflight 1" "--config 3 --steps 40"; do
  bash tools/ab.sh "$args" $L $V/libpqp_nm1.so $V/libpqp_nm2.so
done
} 2>&1 | grep -v "$F" | tee $O/no_monolith_ab.txt
