#!/bin/bash
# Regenerates, on the GPU box, every measurement the documents quote: tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>/...
# (copy what is to be judged into profiles/ afterwards).  About 5 GPU-minutes.
tag=${1:-rxx}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p $out
cd $root
# 1. the driver's command: the headline line (PMC child passes inside bench.py)
timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
# 2. rocprofv3 kernel statistics + PMC passes of the same command (two batches in flight), and one launch at a time
bash tools/profile_bench.sh ${tag}_n1 --no-secondary --pmc off > $out/profile_n1.log 2>&1
cp gpurun_out/prof/${tag}_n1/kernel_stats.csv $out/bench_n1_kernel_stats.csv
cp gpurun_out/prof/${tag}_n1/pmc_per_launch.json $out/bench_n1_pmc_rocprof_passes.json
cp gpurun_out/prof/${tag}_n1/bench_line_under_rocprof.json $out/bench_n1_under_rocprof.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $out/stats1 -- python $root/bench.py --no-cpu-baseline --no-secondary --pmc off --inflight 1 --steps 20 --warmup 3 > $out/stats1.log 2>&1 )
find $out/stats1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_n1_one_at_a_time_kernel_stats.csv
grep '^{"metric"' $out/stats1.log > $out/bench_n1_one_at_a_time_under_rocprof.json
rm -rf $out/stats1
# 3. the other configurations
for spec in "config2:--config 2" "config3:--config 3" "config4:--config 4" "n200_batch512:--config 1 --n 200 --batch 512" "batch65536_one_gpu:--config 3 --batch 65536 --steps 3 --warmup 1"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 900 python bench.py --no-cpu-baseline --pmc off $args > $out/bench_$name.json 2> $out/bench_$name.err
done
# 4. the device-resident chain and the device timeline
{ for a in "1024 8 10 --exact-smoothers" "4096 8 10 --exact-smoothers" "1024 8 10 --exact-smoothers --inflight-2" "4096 8 10 --exact-smoothers --inflight-2" \
          "1024 8 10 --exact-smoothers --default-capacities" "1024 8 10" "1024 8 10 --tension --exact-smoothers" "1024 8 10 --tension"; do python tools/bench_full_chain.py $a; done; } > $out/full_chain.txt 2>&1
# 5. the smoother QPs (reference setting, exact kernels), the exact TensionSmoother kernel's rounds, per-kernel PMC of the chain
{ python tools/bench_smoothers.py 1024 80; python tools/bench_smoothers.py 1024 48; python tools/bench_smoothers.py 512 200; python tools/smoother_rounds.py; } > $out/smoothers.txt 2>&1
CHAIN_ARGS=--exact-smoothers bash tools/pmc_chain.sh ${tag}_chain 1024 > /dev/null 2>&1; cp gpurun_out/prof/${tag}_chain/chain_pmc.txt $out/chain_pmc.txt
python tools/kernel_timeline.py 1024 80 > $out/timeline_batch1024_n80.txt 2>&1
for m in 0x0 0x10 0x2000 0x20; do echo "== PQP_TIMING_MASK=$m (one category per build)"; PQP_TIMING_MASK=$m python tools/kernel_timeline.py 1024 80 2>/dev/null | grep -v " 0.0 us"; done > $out/timeline_selective.txt 2>&1
tail -c 600 $out/bench_n1.json; ls $out
