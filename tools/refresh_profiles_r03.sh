#!/bin/bash
# Round-3 evidence in one GPU call: tests, the default bench line, the stream kernel's bench line, the other configs, rocprofv3 kernel
# statistics of both.  Writes gpurun_out/r03<tag>_*; copy what is to be judged into profiles/.   Usage: tools/refresh_profiles_r03.sh <tag>
tag=${1:-x}; o=gpurun_out/r03${tag}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6) > ${o}_pytest.log 2>&1
(time timeout 600 python bench.py > ${o}_bench_n1.json) 2> ${o}_bench_n1.err
(time timeout 600 python bench.py --config 3 --batch 65536 --steps 40 > ${o}_bench_stream_65536.json) 2> ${o}_bench_stream_65536.err
timeout 300 python bench.py --config 2 --steps 40 --no-cpu-baseline > ${o}_bench_config2.json 2> /dev/null
timeout 300 python bench.py --config 3 --steps 100 --no-cpu-baseline > ${o}_bench_config3_shard.json 2> /dev/null
timeout 300 python bench.py --config 4 --steps 200 --no-cpu-baseline > ${o}_bench_config4.json 2> /dev/null
timeout 300 python bench.py --batch 512 --n 200 --steps 200 --no-cpu-baseline --pmc off > ${o}_bench_n200_batch512.json 2> /dev/null
export TMPDIR=/tmp
root=$PWD
(cd /tmp && rm -rf /tmp/rp1 && rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp1 -- python $root/bench.py --no-cpu-baseline --no-secondary --pmc off --sustain 0 > ${root}/${o}_bench_n1_under_rocprof.json 2> /dev/null)
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) ${o}_bench_n1_kernel_stats.csv
(cd /tmp && rm -rf /tmp/rp2 && rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp2 -- python $root/bench.py --config 3 --batch 65536 --steps 40 --no-cpu-baseline --no-secondary --pmc off --sustain 0 > ${root}/${o}_bench_stream_65536_under_rocprof.json 2> /dev/null)
cp $(find /tmp/rp2 -name "*kernel_stats.csv" | head -1) ${o}_bench_stream_65536_kernel_stats.csv
(cd /tmp && rm -rf /tmp/rp3 && rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp3 -- python $root/bench.py --config 3 --batch 65536 --steps 20 --inflight 1 --no-cpu-baseline --no-secondary --pmc off --sustain 0 > ${root}/${o}_bench_stream_65536_one_at_a_time_under_rocprof.json 2> /dev/null)
cp $(find /tmp/rp3 -name "*kernel_stats.csv" | head -1) ${o}_bench_stream_65536_one_at_a_time_kernel_stats.csv
# the smoother QPs (cold and carried), the exact TensionSmoother kernel on distinct lines, the device-resident chain on moving scenarios
{ timeout 200 python tools/bench_smoothers_carry.py; timeout 100 python tools/smoother_rounds.py; for b in 4096 16384; do timeout 100 python tools/bench_smoothers_carry.py $b 80 | grep tension; done; } 2>&1 | grep -v amdgpu.ids > ${o}_smoothers.txt
timeout 300 python tools/tension_fuzz.py 2048 2>&1 | grep -v amdgpu.ids > ${o}_tension_fuzz.txt
for f in "--exact-smoothers --moving" "--exact-smoothers --carry" "--exact-smoothers --tension --moving" "--exact-smoothers --tension --carry" "--exact-smoothers --inflight-2 --moving" "--exact-smoothers --inflight-2 --carry" "--exact-smoothers"; do echo "$f"; timeout 120 python tools/bench_full_chain.py 1024 8 20 $f 2>&1 | grep -v amdgpu.ids; done > ${o}_full_chain.txt
CHAIN_ARGS="--exact-smoothers --carry" timeout 300 bash tools/pmc_chain.sh r03${tag}_chain_carry > /dev/null 2>&1; cp gpurun_out/prof/r03${tag}_chain_carry/chain_pmc.txt ${o}_chain_carry_pmc.txt
tail -3 ${o}_pytest.log
python - <<PY
import json
for f in ("${o}_bench_n1.json", "${o}_bench_stream_65536.json", "${o}_bench_config2.json", "${o}_bench_config3_shard.json", "${o}_bench_config4.json", "${o}_bench_n200_batch512.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f.split("r03")[1], "value %.3g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "solved", d["solved"], "sha", d["out_sha1"], "| frac", r.get("frac"), "measured", r.get("hbm_measured_frac"), r.get("hbm_measured_frac_chip_wide"))
        for k, v in (d.get("secondary") or {}).items():
            if v and "value" in v: print("     ", k, "%.3g" % v["value"])
            elif v and "lane_per_qp_stream_kernel" in v: print("     ", k, "%.3g" % v["lane_per_qp_stream_kernel"]["value"], "vs %.3g" % v["lane_per_waypoint_kernel"]["value"])
            elif v: print("     ", k, str(v)[:200])
        cb = d.get("cpu_baseline")
        if cb: print("      cpu", "%.3g" % cb["value"], cb["cores"], "single %.3g" % cb["single_thread"]["value"], "same-alg", cb["same_algorithm_on_host"].get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
