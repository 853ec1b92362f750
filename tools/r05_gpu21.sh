#!/bin/bash
# Round 5, GPU call 21: the interval rule at its ends: 4 / 5 / 6 iterations at 48 ... 64 waypoints, 8 / 10 / 12 at 120 ... 200.
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-secondary --pmc off --sustain 0"
run() { python bench.py $2 --polish-every $1 --check-termination $1 --rho-interval $1 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('k=%-2s  %-40s %9.0f /s  step %.4f ms solved %d kkt %.1f max %.0f fac %.1f max %.0f' % ('$1', '$2', d['value'], d['ms_per_step'], d['solved'], d['kkt_solves']['mean'], d['kkt_solves']['max'], d['factorisations']['mean'], d['factorisations']['max']))"; }
for rep in 1 2; do
for a in "--batch 8192 --n 48 --steps 60" "--batch 8192 --n 64 --steps 60" "--batch 1024 --n 60 --steps 400" "--batch 1024 --n 40 --steps 400"; do for k in 4 5 6; do run $k "$a"; done; done
for a in "--config 2 --steps 40" "--batch 8192 --n 128 --steps 40" "--batch 512 --n 200 --steps 200" "--batch 2048 --n 256 --steps 60"; do for k in 8 10 12; do run $k "$a"; done; done
done
