#!/bin/bash
# Round 5, GPU call 17: start curvature projected in the lane-per-waypoint kernel too: GPU suite, the kernels before (ab/libpqp_r05t.so) / after on the bench workloads.
o=gpurun_out/r05v; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|FAILED\|rror" | tee ${o}_pytest.log
Q="--no-cpu-baseline --no-secondary --pmc off --sustain 0"
for rep in 1 2 3; do for lib in r05t new; do
  if [ $lib = new ]; then unset PQP_LIB; else export PQP_LIB=$PWD/ab/libpqp_$lib.so; fi
  for a in "--steps 400" "--config 3 --steps 60" "--batch 8192 --n 64 --steps 60" "--batch 512 --n 200 --steps 200" "--config 2 --steps 40"; do
  timeout 200 python bench.py $a $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('$lib  %-32s %9.0f /s  step %.4f ms solved %d sha %s' % ('$a', d['value'], d['ms_per_step'], d['solved'], d['out_sha1']))"
  done; done; done | tee ${o}_ab.txt
