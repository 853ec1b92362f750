#!/bin/bash
# rocprofv3 evidence for bench.py's default command (run on the GPU box; outputs under gpurun_out/prof/).
#   1. --kernel-trace --stats            -> per-kernel average duration (must agree with bench.py's HIP-event figure)
#   2. separate --pmc passes (no trace domains beside --kernel-trace): HBM bytes, instruction mix, wait/busy cycles
# Usage: bash tools/profile_bench.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof/$tag; mkdir -p $out
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --no-cpu-baseline --steps 20 --warmup 3 $*"
rocprofv3 --kernel-trace --stats -f csv -d $out/stats -- $cmd > $out/stats.log 2>&1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out/pmc_$i -- $cmd > $out/pmc_$i.log 2>&1
done
cd $root
python tools/pmc_aggregate.py $out path_solve_kernel $out/pmc_per_launch.json > /dev/null
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
grep '^{"metric"' $out/stats.log > $out/bench_line_under_rocprof.json
# keep the merged-back payload small
find $out -name "*.csv" -size +2M -delete
head -3 $out/kernel_stats.csv; cat $out/pmc_per_launch.json
