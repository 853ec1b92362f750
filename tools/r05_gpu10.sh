#!/bin/bash
# Round 5, GPU call 10: the ordering routine with several accesses in flight per lane.
o=gpurun_out/r05i; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 600 python -m pytest tests/test_gpu_stream.py -m gpu -q 2>&1 | grep -v "$F" | tail -5) > ${o}_pytest_stream.log 2>&1
tail -3 ${o}_pytest_stream.log
for mode in "" "--no-cost-order"; do
  echo "bench_stream $mode"; (timeout 300 python tools/bench_stream.py --n 80 --batches 32768,49152,57344,65536,98304,131072 --steps 6 --oracle 0 --skip-old $mode 2>&1 | grep -v "$F") | cut -c1-170
done | tee ${o}_stream_ordered.txt
for fl in 1 2; do
  timeout 400 python bench.py --config 3 --batch 65536 --steps 40 --inflight $fl --no-cpu-baseline --no-secondary --sustain 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
r=d['roofline']
print('config 3 whole batch, jittered variants, inflight $fl: %9.0f paths/s  step %.3f ms kernel %.3f ms  traffic/algorithmic %s  hbm_measured_frac %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r.get('traffic_over_algorithmic'), r.get('hbm_measured_frac')))"
done | tee ${o}_bench_stream_jittered.txt
