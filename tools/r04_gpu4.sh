#!/bin/bash
# round 4, fourth GPU call: A/B of the lane-per-QP kernel (reciprocal diet vs the previous build), workspace block padding (HBM channel spread)
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
run() { echo "== $1 PAD=${2:-0}" >> $O/ab_stream.txt; PQP_LIB=$1 PQP_STREAM_PAD=${2:-0} timeout 300 python tools/bench_stream.py --skip-old --batches $3 --steps 6 --oracle 0 2>&1 | grep -v "$F" | grep "stream" >> $O/ab_stream.txt; }
NEW=path_optimizer_2_amd/csrc/libpqp_hip.so; OLD=build_variants/libpqp_lqold.so
for rep in 1 2; do run $NEW 0 16384,65536; run $OLD 0 16384,65536; done
for pad in 1 3 17 33 129 257; do run $NEW $pad 65536; done
cat $O/ab_stream.txt
