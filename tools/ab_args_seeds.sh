#!/bin/bash
# A/B of bench settings over scenario seeds, two batches in flight and one launch at a time: tools/ab_args_seeds.sh "<seeds>" "<args 1>" "<args 2>" ...
seeds="$1"; shift
for v in "$@"; do
  for seed in $seeds; do
    s=""; [ $seed != default ] && s="--seed $seed"
    a=$(timeout 100 python bench.py --no-cpu-baseline --no-secondary --pmc off --sustain 0 --steps 200 --warmup 8 $s $v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f M kkt %.1f max %d' % (d['value']/1e6, d['kkt_solves']['mean'], d['kkt_solves']['max']))")
    b=$(timeout 100 python bench.py --no-cpu-baseline --no-secondary --pmc off --sustain 0 --steps 200 --warmup 8 --inflight 1 $s $v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f M' % (d['value']/1e6))")
    echo "[$v] seed $seed: $a | one at a time $b"
  done
done
