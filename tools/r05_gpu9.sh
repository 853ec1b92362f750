#!/bin/bash
# Round 5, GPU call 9: wavefronts of similar work in path_stream_kernel (PQP_OPT_ORDER_BY_COST on the lane-per-QP kernel).
o=gpurun_out/r05h; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -15) > ${o}_pytest.log 2>&1
tail -4 ${o}_pytest.log
for mode in "" "--no-cost-order"; do
  echo "bench_stream $mode"; (timeout 300 python tools/bench_stream.py --n 80 --batches 32768,49152,65536,98304 --steps 6 --oracle 0 --skip-old $mode 2>&1 | grep -v "$F") | cut -c1-170
done | tee ${o}_stream_ordered.txt
for fl in 1 2; do
  timeout 400 python bench.py --config 3 --batch 65536 --steps 40 --inflight $fl --no-cpu-baseline --no-secondary --sustain 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
r=d['roofline']
print('config 3 whole batch, jittered variants, inflight $fl: %9.0f paths/s  step %.3f ms kernel %.3f ms  sweeps %s  traffic/algorithmic %s  hbm_measured_frac %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], d.get('riccati_sweeps'), r.get('traffic_over_algorithmic'), r.get('hbm_measured_frac')))"
done | tee ${o}_bench_stream_jittered.txt
(time timeout 900 python bench.py > ${o}_bench_n1.json) 2> ${o}_bench_n1.err
python - <<PY
import json
d = json.loads(open("${o}_bench_n1.json").read().strip().splitlines()[-1])
print("value %.4g" % d["value"], "sustained %.4g" % d["sustained"]["value"])
w = d["secondary"]["configs3_whole_batch_one_gpu"]["lane_per_qp_stream_kernel"]
print({k: (round(v["value"]) if isinstance(v, dict) and "value" in v else v) for k, v in w.items() if k not in ("roofline",)})
print({k: w["roofline"].get(k) for k in ("frac", "traffic", "traffic_over_algorithmic", "hbm_measured_frac", "algorithmic_bytes_per_launch")})
PY
