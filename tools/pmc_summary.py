#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel.  usage: pmc_summary.py OUT.json DIR [DIR...]"""
import csv, glob, json, os, sys
out, dirs = sys.argv[1], sys.argv[2:]
acc = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            e = acc.setdefault(k, {"launches": set(), "counters": {}, "launch": {}})
            e["launches"].add((d, row["Dispatch_Id"]))
            e["counters"][row["Counter_Name"]] = e["counters"].get(row["Counter_Name"], 0.) + float(row["Counter_Value"])
            for c in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"):
                if c in row: e["launch"][c] = row[c]
res = [{"kernel": k, "launch": v["launch"], "counters": v["counters"]} for k, v in acc.items() if "coop_kernel" in k or "symphony" in k]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
