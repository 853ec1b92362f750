"""Average the rocprofv3 counter CSVs of several --pmc passes per launch of one kernel.
Usage: python tools/pmc_aggregate.py <dir with pmc_*/ subdirs> <kernel name substring> <out.json>"""
import csv, glob, json, os, sys
root, kname, out = sys.argv[1], sys.argv[2], sys.argv[3]
acc, cnt, disp = {}, {}, None
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if kname not in row["Kernel_Name"]:
            continue
        c = row["Counter_Name"]
        acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
        cnt[c] = cnt.get(c, 0) + 1
        if disp is None:
            disp = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                       "Accum_VGPR_Count", "SGPR_Count") if k in row}
res = {c: acc[c] / cnt[c] for c in sorted(acc)}
res["_dispatch"] = disp
res["_launches_per_counter"] = {c: cnt[c] for c in sorted(cnt)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
