#!/bin/bash
# Round 5, GPU call 19: polish_every = check_termination = adaptive_rho_interval = 5 against the production setting's 8: scenario seeds, the other configs, one launch at a time.
o=gpurun_out/r05x; mkdir -p gpurun_out
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-secondary --pmc off --sustain 0"
run() { python bench.py $2 --polish-every $1 --check-termination $1 --rho-interval $1 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('k=$1  %-44s %9.0f /s  step %.4f ms solved %d kkt %.1f max %.0f fac %.1f max %.0f' % ('$2', d['value'], d['ms_per_step'], d['solved'], d['kkt_solves']['mean'], d['kkt_solves']['max'], d['factorisations']['mean'], d['factorisations']['max']))"; }
for s in 1 2 3 4 5 6 7 8; do for k in 8 5; do run $k "--steps 400 --seed $s"; done; done
for a in "--steps 400 --inflight 1" "--config 2 --steps 40" "--batch 512 --n 200 --steps 200" "--config 4 --steps 200" "--batch 8192 --n 64 --steps 60" "--batch 8192 --n 128 --steps 40" "--batch 1024 --n 60 --steps 400"; do for k in 8 5 6; do run $k "$a"; done; done
