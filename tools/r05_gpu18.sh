#!/bin/bash
# Round 5, GPU call 18: pqp_kernels.hip compiled with LLVM scheduler / register-allocator options, against the production build.
o=gpurun_out/r05w; mkdir -p gpurun_out
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-secondary --pmc off --sustain 0"
for rep in 1 2; do for lib in new ilp mem trump bias0 revloc sspeed; do
  if [ $lib = new ]; then unset PQP_LIB; else export PQP_LIB=$PWD/ab/libpqp_$lib.so; fi
  for a in "--steps 400" "--config 3 --steps 60" "--batch 8192 --n 64 --steps 60"; do
  timeout 200 python bench.py $a $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('$lib  %-32s %9.0f /s  step %.4f ms solved %d sha %s' % ('$a', d['value'], d['ms_per_step'], d['solved'], d['out_sha1']))"
  done; done; done | tee ${o}_compile_flags.txt
