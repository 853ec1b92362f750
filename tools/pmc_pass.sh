#!/bin/bash
# Extra rocprofv3 counter passes over the bench command (GPU box): tools/pmc_pass.sh <tag> "<bench args>" "<counters of pass 1>" "<pass 2>" ...
# One pass per counter group, --kernel-trace only beside --pmc; per-launch averages of path_solve_kernel -> gpurun_out/prof/<tag>/pmc_per_launch.json
tag=$1; bargs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --no-cpu-baseline --no-secondary --pmc off --sustain 0 --prewarm 0 --steps 10 --warmup 3 $bargs"
i=0
for pmc in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out/pmc_$i -- $cmd > $out/pmc_$i.log 2>&1
done
cd $root
python tools/pmc_aggregate.py $out path_solve_kernel $out/pmc_per_launch.json
find $out -name "*.csv" -size +1M -delete
