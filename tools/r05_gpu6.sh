#!/bin/bash
# Round 5, GPU call 6: the final build once more - GPU suite, the default bench line exactly as the driver runs it (K = 20), smoke().
o=gpurun_out/r05f; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -8) > ${o}_pytest.log 2>&1
tail -3 ${o}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2
(time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > ${o}_bench_as_the_driver_runs_it.json) 2> ${o}_bench.err
tail -3 ${o}_bench.err
python - <<PY
import json
d = json.loads(open("${o}_bench_as_the_driver_runs_it.json").read().strip().splitlines()[-1])
print("value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "steps", d["steps"], "sustained %.4g" % d["sustained"]["value"], "lanes", d["lanes_active_frac"]["value"], "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
for k, v in (d.get("secondary") or {}).items():
    if v and "value" in v: print("     ", k, "%.4g" % v["value"], v.get("steps"))
    elif v and "lane_per_qp_stream_kernel" in v: print("     ", k, "%.4g" % v["lane_per_qp_stream_kernel"]["value"], "vs %.4g" % v["lane_per_waypoint_kernel"]["value"])
    elif v and "error" in v: print("     ", k, v)
PY
