#!/bin/bash
# rocprofv3 evidence for the kernels AROUND the path QP (smoother QPs, spline, reference states, corridor bounds, DP search): kernel
# statistics + separate --pmc passes of tools/bench_full_chain.py, averaged per launch and kernel -> gpurun_out/prof/<tag>/chain_pmc.txt
tag=${1:-chain}; batch=${2:-1024}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $root/tools/bench_full_chain.py $batch 8 5 ${CHAIN_ARGS:-}"
rocprofv3 --kernel-trace --stats -f csv -d $out/stats -- $cmd > $out/stats.log 2>&1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out/pmc_$i -- $cmd > $out/pmc_$i.log 2>&1
done
cd $root
python - $out <<'PY' > $out/chain_pmc.txt
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
dur = {}
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Name"].split("(")[0].replace("void ", "")] = (float(row["AverageNs"]), int(row["Calls"]), float(row["Percentage"]))
print("kernel | calls | avg us | % of GPU time | HBM MB/launch (2 x FETCH_SIZE + WRITE_SIZE) | GB/s | HBM roofline frac (of 8 TB/s) | VALU issue roofline frac (VALU-active cycles / (1024 SIMDs x kernel cycles)) | VALU-active / wave cycles | LDS-active | waiting | VALU instr/launch | LDS instr/launch")
for k, (ns, calls, pct) in sorted(dur.items(), key=lambda kv: -kv[1][2]):
    if k not in acc: continue
    a = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    mb = (2 * a.get("FETCH_SIZE", 0) + a.get("WRITE_SIZE", 0)) * 1024 / 1e6
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    kc = (a.get("GRBM_GUI_ACTIVE", 0) / 8.0) or ns * 2.4
    issue = a.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / (1024 * kc)
    print(f"{k} | {calls} | {ns/1e3:.1f} | {pct:.1f} | {mb:.2f} | {mb*1e6/ns:.0f} | {mb*1e6/ns/8000:.4f} | {issue:.3f} | {a.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} | {a.get('SQ_ACTIVE_INST_LDS',0)/wc:.2f} | {a.get('SQ_WAIT_ANY',0)/wc:.2f} | {a.get('SQ_INSTS_VALU',0):.3g} | {a.get('SQ_INSTS_LDS',0):.3g}")
PY
find $out -name "*.csv" -size +1M -delete
cat $out/chain_pmc.txt
