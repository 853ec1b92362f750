#!/bin/bash
# batch-1024 kernel time of the bench workload over several scenario seeds: the batch is as slow as its slowest QP, so the
# headline figure depends on the draw.  Usage: bash tools/seed_sweep.sh [bench args...]  (run on the GPU box)
for seed in default 0 1 2 3 4 5 6 7; do
  s=""; [ $seed != default ] && s="--seed $seed"
  python bench.py --no-cpu-baseline --steps 12 --warmup 2 $s "$@" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('seed $seed: kernel %.3f ms  %d paths/s  solved %d  iters p99 %s max %s' % (d['roofline']['kernel_ms'], d['value'], d['solved'], d['admm_iters']['p99'], d['admm_iters'].get('max')))"
done
