#!/bin/bash
# Round 5, GPU call 2: full GPU suite; lane kernel before / after the carry-tails change (register allocation moved: 48 -> 82 VGPR spills);
# stream kernel scheduler variants; carry tails in the bench line.
o=gpurun_out/r05b; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -15) > ${o}_pytest.log 2>&1
tail -5 ${o}_pytest.log
Q="--no-cpu-baseline --no-secondary --pmc off --sustain 0"
for rep in 1 2; do for lib in s_bias0 carry; do
  for cfgargs in "--steps 400" "--steps 200 --inflight 1" "--config 3 --steps 60"; do
  PQP_LIB=$PWD/ab/libpqp_$lib.so timeout 200 python bench.py $Q $cfgargs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('%-10s %-28s value %9.0f paths/s  step %.4f ms kernel %.4f ms  sha %s solved %d' % ('$lib', '$cfgargs', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['out_sha1'], d['solved']))"
  done; done; done 2>&1 | tee ${o}_lane_ab.txt
for lib in s_ilp s_mem s_o2 s_bias0 carry; do
  (PQP_LIB=$PWD/ab/libpqp_$lib.so timeout 300 python tools/bench_stream.py --n 80 --batches 8192,16384,65536 --steps 6 --oracle 0 --skip-old 2>&1 | grep -v "$F") > ${o}_stream_$lib.txt
  cat ${o}_stream_$lib.txt | cut -c1-150
done
(time timeout 600 python bench.py > ${o}_bench_n1.json) 2> ${o}_bench_n1.err
python - <<PY
import json
for f in ("${o}_bench_n1.json",):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "solved", d["solved"], "sha", d["out_sha1"])
        for k, v in (d.get("secondary") or {}).items():
            if v and "value" in v: print("     ", k, "%.4g" % v["value"], v.get("kkt_solves_mean"), v.get("kkt_solves_max"))
            elif v and "lane_per_qp_stream_kernel" in v: print("     ", k, "%.4g" % v["lane_per_qp_stream_kernel"]["value"], "vs %.4g" % v["lane_per_waypoint_kernel"]["value"], "traffic", v["lane_per_qp_stream_kernel"]["roofline"].get("traffic"))
    except Exception as e:
        print(f, "ERR", e)
PY
