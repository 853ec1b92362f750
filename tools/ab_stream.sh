#!/bin/bash
# A/B of library builds on the lane-per-QP (stream) kernel's lines in ONE GPU call:  LIBS="name=path ..." tools/ab_stream.sh <tag>
tag=$1; o=gpurun_out/${tag}; mkdir -p gpurun_out; root=$PWD
Q="--no-cpu-baseline --no-secondary --pmc off --sustain 0"
line() { python -c "
import json,sys
ls=[l for l in sys.stdin.readlines() if l.startswith('{\"metric\"')]
if not ls: print('$1: no bench line'); sys.exit(0)
d=json.loads(ls[-1])
k, f = d.get('kkt_solves') or d.get('riccati_sweeps') or {}, d.get('factorisations') or d.get('active_set_rounds') or {}
print('%-60s %9.0f paths/s  step %.4f ms  solved %d  sweeps %.1f (max %.0f)  rounds %.1f (max %.0f)  sha %s' % ('$1', d['value'], d['ms_per_step'], d['solved'], k.get('mean', 0), k.get('max', 0), f.get('mean', 0), f.get('max', 0), d['out_sha1'][:10]))"; }
for rep in $(seq 1 ${REPS:-2}); do for nl in $LIBS; do name=${nl%%=*}; lib=${nl#*=}
  for a in "--config 3 --batch 65536 --steps 40" "--config 3 --batch 65536 --steps 20 --inflight 1" "--batch 32768 --n 120 --steps 20" ${AB_EXTRA:+"$AB_EXTRA"}; do
    PQP_LIB=$lib timeout 300 python bench.py $a $Q 2>/dev/null | line "$name | $a"
  done; done; done | tee ${o}_ab_stream.txt
