#!/bin/bash
# Round 5, GPU call 4: full GPU suite; PQP_OPT_CARRY_CYCLES = k (tails) on configs[1], configs[3]'s shard and configs[2].
o=gpurun_out/r05d; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -25) > ${o}_pytest.log 2>&1
tail -6 ${o}_pytest.log
(time timeout 600 python bench.py --no-cpu-baseline --pmc off > ${o}_bench_n1.json) 2> ${o}_bench_n1.err
timeout 400 python bench.py --config 3 --steps 100 --no-cpu-baseline --pmc off > ${o}_bench_config3_shard.json 2> /dev/null
timeout 400 python bench.py --config 2 --steps 60 --no-cpu-baseline --pmc off > ${o}_bench_config2.json 2> /dev/null
python - <<PY
import json
for f in ("${o}_bench_n1.json", "${o}_bench_config3_shard.json", "${o}_bench_config2.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "solved", d["solved"], "sha", d["out_sha1"], "kkt", d.get("kkt_solves"))
        for k, v in (d.get("secondary") or {}).items():
            if v and "value" in v: print("     ", k, "%.4g" % v["value"], v.get("kkt_solves_mean"), v.get("kkt_solves_max"))
    except Exception as e:
        print(f, "ERR", e)
PY
