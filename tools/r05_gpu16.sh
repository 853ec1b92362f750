#!/bin/bash
# Round 5, GPU call 16: guarded active-set rounds of the lane-per-QP kernel (long paths whose rounds cycle): GPU suite, the kernel before / after at the sizes it serves, 512 waypoints.
o=gpurun_out/r05u; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|FAILED\|rror" | tee ${o}_pytest.log
for rep in 1 2; do for lib in r05t new; do
  if [ $lib = new ]; then unset PQP_LIB; else export PQP_LIB=$PWD/ab/libpqp_$lib.so; fi
  timeout 300 python tools/bench_stream.py --n 80 --batches 16384,65536 --steps 10 --skip-old --oracle 0 2>&1 | grep -v "$F" | grep "batch" | sed "s/^/$lib /"
  timeout 300 python tools/bench_stream.py --n 120 --profile varied --batches 65536 --steps 6 --skip-old --oracle 0 2>&1 | grep -v "$F" | grep "batch" | sed "s/^/$lib /"
done; done | tee ${o}_stream_before_after.txt
unset PQP_LIB
timeout 300 python tools/bench_stream.py --n 512 --profile varied --batches 16384 --steps 2 --skip-old --oracle 0 2>&1 | grep -v "$F" | tee ${o}_n512.txt | tail -3
