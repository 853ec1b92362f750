#!/bin/bash
# HBM traffic and issue counters of path_stream_kernel: rocprofv3 --pmc passes (their own runs, --kernel-trace only beside them) of
# tools/bench_stream.py at one batch size.  Usage: tools/pmc_stream.sh <batch> <out prefix>      (run on the GPU box)
batch=${1:-65536}; out=${2:-gpurun_out/pmc_stream}
root=$PWD
export TMPDIR=/tmp
for pass in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE SQ_WAVES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  d=/tmp/pmc_$tag; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -- python $root/tools/bench_stream.py --steps 2 --skip-old --oracle 0 --batches $batch > /dev/null 2>&1)
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" >> ${out}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "path_stream_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k}: mean per launch {sum(v) / len(v):.6g} over {len(v)} launches")
PY
done
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d /tmp/pmc_stats -- python $root/tools/bench_stream.py --steps 4 --skip-old --oracle 0 --batches $batch > /dev/null 2>&1)
cp $(find /tmp/pmc_stats -name "*kernel_stats.csv" | head -1) ${out}_kernel_stats.csv
cat ${out}_pmc.txt; cat ${out}_kernel_stats.csv
