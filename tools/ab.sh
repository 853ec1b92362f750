#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh "<bench args>" lib1.so lib2.so ...   -> one summary line per build
args="$1"; shift
for lib in "$@"; do
  PQP_LIB=$lib python bench.py --no-cpu-baseline --no-secondary --pmc off --sustain 0 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
k=d.get('kkt_solves',{}).get('mean',0)
print('%-26s %-44s value %9.0f paths/s  step %.4f ms kernel %.4f ms  kkt %.1f  sha %s solved %d' % ('$lib'.split('/')[-1], '$args', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], k, d['out_sha1'], d['solved']))"
done
