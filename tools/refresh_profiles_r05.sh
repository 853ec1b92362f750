#!/bin/bash
# Round-5 evidence in one GPU call: the GPU test suite, the default bench line, the other BASELINE configs, the stream kernel's own line,
# rocprofv3 kernel statistics of the two path-QP kernels, the device-resident chain with and without PQP_OPT_CHAIN_GRAPH.
# Writes gpurun_out/r05<tag>_*; copy what is to be judged into profiles/.   Usage: tools/refresh_profiles_r05.sh <tag>
tag=${1:-x}; o=gpurun_out/r05${tag}
mkdir -p gpurun_out
export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "$F" | tail -12) > ${o}_pytest.log 2>&1
(time timeout 900 python bench.py > ${o}_bench_n1.json) 2> ${o}_bench_n1.err
(time timeout 600 python bench.py --config 3 --batch 65536 --steps 40 > ${o}_bench_stream_65536.json) 2> ${o}_bench_stream_65536.err
timeout 300 python bench.py --config 2 --steps 40 --no-cpu-baseline > ${o}_bench_config2.json 2> /dev/null
timeout 300 python bench.py --config 3 --steps 100 --no-cpu-baseline > ${o}_bench_config3_shard.json 2> /dev/null
timeout 300 python bench.py --config 4 --steps 200 --no-cpu-baseline > ${o}_bench_config4.json 2> /dev/null
timeout 300 python bench.py --batch 512 --n 200 --steps 200 --no-cpu-baseline --pmc off > ${o}_bench_n200_batch512.json 2> /dev/null
for nn in 48 60 64 80 96 128; do timeout 200 python bench.py --batch 8192 --n $nn --steps 40 --no-cpu-baseline --no-secondary --pmc off --sustain 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('batch 8192 N = %3d: %9.0f paths/s  step %.3f ms  kkt %.1f  factorisations %.1f' % ($nn, d['value'], d['ms_per_step'], d['kkt_solves']['mean'], d['factorisations']['mean']))"; done > ${o}_n_sweep_batch8192.txt
cat ${o}_n_sweep_batch8192.txt
root=$PWD
(cd /tmp && rm -rf /tmp/rp1 && rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp1 -- python $root/bench.py --no-cpu-baseline --no-secondary --pmc off --sustain 0 > ${root}/${o}_bench_n1_under_rocprof.json 2> /dev/null)
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) ${o}_bench_n1_kernel_stats.csv
(cd /tmp && rm -rf /tmp/rp3 && rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp3 -- python $root/bench.py --config 3 --batch 65536 --steps 20 --inflight 1 --no-cpu-baseline --no-secondary --pmc off --sustain 0 > ${root}/${o}_bench_stream_65536_one_at_a_time_under_rocprof.json 2> /dev/null)
cp $(find /tmp/rp3 -name "*kernel_stats.csv" | head -1) ${o}_bench_stream_65536_one_at_a_time_kernel_stats.csv
for f in "--exact-smoothers --moving" "--exact-smoothers --moving --graph"; do echo "$f"; timeout 120 python tools/bench_full_chain.py 1024 8 30 $f 2>&1 | grep -v "$F"; done > ${o}_full_chain.txt
tail -4 ${o}_pytest.log
python - <<PY
import json
for f in ("${o}_bench_n1.json", "${o}_bench_stream_65536.json", "${o}_bench_config2.json", "${o}_bench_config3_shard.json", "${o}_bench_config4.json", "${o}_bench_n200_batch512.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f.split("r05")[1], "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "solved", d["solved"], "sha", d["out_sha1"], "| frac", r.get("frac"), "measured", r.get("hbm_measured_frac"))
        for k, v in (d.get("secondary") or {}).items():
            if v and "value" in v: print("     ", k, "%.4g" % v["value"])
            elif v and "lane_per_qp_stream_kernel" in v: print("     ", k, "%.4g" % v["lane_per_qp_stream_kernel"]["value"], "vs %.4g" % v["lane_per_waypoint_kernel"]["value"], "traffic", v["lane_per_qp_stream_kernel"]["roofline"].get("traffic"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep "scenarios/s\|^--" ${o}_full_chain.txt | cut -c1-125
