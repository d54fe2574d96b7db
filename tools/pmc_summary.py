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
res = [{"kernel": k, "launch": v["launch"], "counters": v["counters"]} for k, v in acc.items() if "coop_kernel" in k or "symphony" in k or "group_kernel" in k]
# the launch's own work counters (tools/one_batch.py prints them) next to the hardware counters, so that per-pass and
# per-sample figures -- and bench.py's traffic check -- can be derived from this one file
import re
for d in dirs:
    log = os.path.join(os.path.dirname(d.rstrip("/")), "log1.txt")
    if os.path.exists(log):
        text = open(log).read()
        for pat, faraday in ((r"^kernel ms ([0-9.]+) samples (\d+) passes (\d+) inner_qags (\d+)", False),
                             (r"^faraday kernel ms ([0-9.]+) samples (\d+) passes (\d+) inner_qags (\d+)", True)):
            m = re.search(pat, text, re.M)
            if m:
                for r in res:
                    if ("Heyvaerts" in r["kernel"] or "HeyGroup" in r["kernel"]) == faraday:
                        r["work"] = {"kernel_ms_under_pmc": float(m.group(1)), "samples": int(m.group(2)), "passes": int(m.group(3)), "inner_qags": int(m.group(4))}
        break
# the build the counters were taken on: bench.py only quotes them for that build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rimphony_amd import _build
for r in res:
    r["source_id"] = _build.source_id()
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
