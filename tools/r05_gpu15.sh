#!/bin/bash
# Round 5, GPU call 15: polish_final_refine on long paths: robustness sweep, GPU suite, configs[4] / N = 200, and the N = 80 kernel before / after (80 against 62 spilled VGPRs).
o=gpurun_out/r05q; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
timeout 600 python tools/robustness_sweep.py 16 8192 2>&1 | grep -v "$F" | tee ${o}_robustness_sweep.txt | grep "n 200\|n 300\|lane-per-QP" | tail -5
python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | grep "passed\|failed\|FAILED\|rror" | tee ${o}_pytest.log
Q="--no-cpu-baseline --no-secondary --pmc off --sustain 0"
for rep in 1 2; do for lib in notail new; do
  if [ $lib = new ]; then unset PQP_LIB; else export PQP_LIB=$PWD/ab/libpqp_$lib.so; fi
  for a in "--steps 400" "--config 3 --steps 60" "--batch 512 --n 200 --steps 200" "--config 4 --steps 200"; do
  timeout 200 python bench.py $a $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('$lib  %-32s %9.0f /s  step %.4f ms solved %d sha %s' % ('$a', d['value'], d['ms_per_step'], d['solved'], d['out_sha1']))"
  done; done; done | tee ${o}_ab.txt
