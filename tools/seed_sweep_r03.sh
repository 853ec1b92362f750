#!/bin/bash
# configs[1] over 16 scenario seeds: the headline (8 jittered variants, two batches in flight, cold first solves), one launch at a time,
# and both with the first solve of a cycle carried from the previous cycle.  Usage: tools/seed_sweep_r03.sh > profiles/r03n_seed_sweep.txt
for seed in default 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  s=""; [ $seed != default ] && s="--seed $seed"
  python bench.py --no-cpu-baseline --steps 400 --warmup 8 --pmc off --sustain 0 $s 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']
print('seed %-7s value %.3f M | one at a time %.3f M | carried %.3f M | carried, one at a time %.3f M | solved %d | kkt solves mean %.1f max %d factorisations max %d' % ('$seed', d['value']/1e6, s['one_batch_at_a_time']['value']/1e6, s['carry_cycles']['value']/1e6, s['carry_cycles_one_batch_at_a_time']['value']/1e6, d['solved'], d['kkt_solves']['mean'], d['kkt_solves']['max'], d['factorisations']['max']))"
done
