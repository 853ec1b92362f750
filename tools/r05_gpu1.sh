#!/bin/bash
# Round 5, GPU call 1: the GPU suite on the stripped sources; the two path-QP kernels over batch sizes (crossover of PQP_OPT_STREAM_BATCH);
# the stream kernel before / after the compile-time lane count (ab/libpqp_old.so = round 4's build) and with prefetch depth 2.
o=gpurun_out/r05a; mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -8) > ${o}_pytest.log 2>&1
tail -3 ${o}_pytest.log
B=4096,8192,12288,16384,24576,32768,65536
(timeout 300 python tools/bench_stream.py --n 80 --batches $B --steps 6 --oracle 0 2>&1 | grep -v "$F") > ${o}_crossover_n80.txt
(timeout 300 python tools/bench_stream.py --n 120 --profile varied --batches 4096,8192,16384,24576,32768 --steps 6 --oracle 0 2>&1 | grep -v "$F") > ${o}_crossover_n120.txt
for lib in old d2; do
  (PQP_LIB=$PWD/ab/libpqp_$lib.so timeout 300 python tools/bench_stream.py --n 80 --batches 8192,16384,32768,65536 --steps 6 --oracle 0 --skip-old 2>&1 | grep -v "$F") > ${o}_stream_$lib.txt
done
(timeout 200 python tools/bench_stream.py --n 80 --batches 8192,16384,32768,65536 --steps 6 --oracle 0 --skip-old 2>&1 | grep -v "$F") > ${o}_stream_new_again.txt
cat ${o}_crossover_n80.txt ${o}_crossover_n120.txt ${o}_stream_old.txt ${o}_stream_d2.txt ${o}_stream_new_again.txt | cut -c1-200
(time timeout 600 python bench.py > ${o}_bench_n1.json) 2> ${o}_bench_n1.err
timeout 300 python bench.py --batch 512 --n 200 --steps 200 --no-cpu-baseline --pmc off > ${o}_bench_n200_batch512.json 2> /dev/null
PQP_LIB=$PWD/ab/libpqp_old.so timeout 300 python bench.py --batch 512 --n 200 --steps 200 --no-cpu-baseline --pmc off > ${o}_bench_n200_batch512_old.json 2> /dev/null
python - <<PY
import json
for f in ("${o}_bench_n1.json", "${o}_bench_n200_batch512.json", "${o}_bench_n200_batch512_old.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "solved", d["solved"], "sha", d["out_sha1"])
        for k, v in (d.get("secondary") or {}).items():
            if v and "value" in v: print("     ", k, "%.4g" % v["value"])
            elif v and "lane_per_qp_stream_kernel" in v: print("     ", k, "%.4g" % v["lane_per_qp_stream_kernel"]["value"], "vs %.4g" % v["lane_per_waypoint_kernel"]["value"], "traffic", v["lane_per_qp_stream_kernel"]["roofline"].get("traffic"))
    except Exception as e:
        print(f, "ERR", e)
PY
