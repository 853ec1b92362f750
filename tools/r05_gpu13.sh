#!/bin/bash
# Round 5, GPU call 13: a fourth one-wavefront workgroup per CU (paths of up to 64 waypoints) - the order histogram out of static LDS.
o=gpurun_out/r05l; mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in lds3 new; do
for nn in 48 60 64 80; do
  if [ $lib = new ]; then unset PQP_LIB; else export PQP_LIB=$PWD/ab/libpqp_$lib.so; fi
  timeout 200 python bench.py --batch 8192 --n $nn --steps 40 --no-cpu-baseline --no-secondary --pmc off --sustain 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('$lib batch 8192 N = %3d: %9.0f paths/s  step %.3f ms  kkt %.1f  factorisations %.1f sha %s' % ($nn, d['value'], d['ms_per_step'], d['kkt_solves']['mean'], d['factorisations']['mean'], d['out_sha1']))"
done; done | tee ${o}_fourth_workgroup_per_cu.txt
unset PQP_LIB
for lib in lds3 new; do
  if [ $lib = new ]; then unset PQP_LIB; else export PQP_LIB=$PWD/ab/libpqp_$lib.so; fi
  timeout 200 python bench.py --batch 1024 --n 60 --steps 400 --no-cpu-baseline --no-secondary --pmc off --sustain 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('$lib batch 1024 N = 60: %9.0f paths/s  step %.3f ms' % (d['value'], d['ms_per_step']))"
done | tee -a ${o}_fourth_workgroup_per_cu.txt
unset PQP_LIB
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3)
