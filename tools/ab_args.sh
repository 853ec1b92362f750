#!/bin/bash
# A/B of bench settings on the GPU box with one library: tools/ab_args.sh "<common bench args>" "<variant args 1>" "<variant args 2>" ...
common="$1"; shift
for v in "$@"; do
  python bench.py --no-cpu-baseline --no-secondary --pmc off --sustain 0 $common $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
k=d.get('kkt_solves',{}).get('mean',0)
print('%-34s %-30s value %9.0f paths/s  step %.4f ms kernel %.4f ms  kkt %.1f  sha %s solved %d' % ('$common', '$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], k, d['out_sha1'], d['solved']))"
done
