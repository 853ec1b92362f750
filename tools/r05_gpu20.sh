#!/bin/bash
# Round 5, GPU call 20: the production intervals by path length (5 up to 100 waypoints, 8 beyond): GPU suite, robustness sweep, the bench workloads against an explicit 8, the threshold (N = 96, 100, 110).
o=gpurun_out/r05aa; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|FAILED\|rror" | tee ${o}_pytest.log
timeout 600 python tools/robustness_sweep.py 16 8192 2>&1 | grep -v "$F" | tee ${o}_robustness_sweep_both_kernels.txt | cut -c1-230 | grep "^n \|not" 
Q="--no-cpu-baseline --no-secondary --pmc off --sustain 0"
run() { python bench.py $2 $3 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('$1  %-40s %9.0f /s  step %.4f ms solved %d kkt %.1f max %.0f fac %.1f max %.0f sha %s' % ('$2', d['value'], d['ms_per_step'], d['solved'], d['kkt_solves']['mean'], d['kkt_solves']['max'], d['factorisations']['mean'], d['factorisations']['max'], d['out_sha1']))"; }
E8="--polish-every 8 --check-termination 8 --rho-interval 8"
E5="--polish-every 5 --check-termination 5 --rho-interval 5"
for rep in 1 2; do for a in "--steps 400" "--config 3 --steps 60" "--config 2 --steps 40" "--batch 512 --n 200 --steps 200" "--batch 8192 --n 96 --steps 40" "--batch 8192 --n 100 --steps 40" "--batch 8192 --n 110 --steps 40"; do run "auto" "$a" ""; run "all8" "$a" "$E8"; run "all5" "$a" "$E5"; done; done | tee ${o}_intervals.txt
