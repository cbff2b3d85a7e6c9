#!/usr/bin/env python3
"""Turn a rocprofv3 --pmc SQ_INSTS_VALU pass of the bench command into profiles/<tag>_valu.json:
executed vector-ALU wave-instructions per launch and per solve for each engine kernel (bench.py's
`valu_issue` prices them at 4 issue cycles per wave64 instruction)."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sbdart_amd._srchash import kernel_source_hash
from collections import defaultdict

root, out, solves_per_launch = sys.argv[1], sys.argv[2], int(sys.argv[3])
nstr, nlyr = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (16, 33)
# only the launches at the bench's own launch size (the kernel's largest grid in the run: the host entry point's
# passes, measured in the same command, are smaller)
rows = []
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "SQ_INSTS_VALU":
            rows.append((row["Kernel_Name"].split("(")[0].strip(), int(row["Grid_Size"]), float(row["Counter_Value"])))
gmax = defaultdict(int)
for k, g, v in rows:
    gmax[k] = max(gmax[k], g)
acc = defaultdict(list)
for k, g, v in rows:
    if g == gmax[k]:
        acc[k].append(v)
res = {"kernel_source_hash": kernel_source_hash(), "solves_per_launch": solves_per_launch, "nstr": nstr, "nlyr": nlyr,
       "note": "SQ_INSTS_VALU summed over the dispatch (all SEs), averaged over the launches at the bench's launch size", "kernels": {}}
for k, v in acc.items():
    if "sbd::" not in k:
        continue
    per_launch = sum(v) / len(v)
    res["kernels"][k] = {"valu_wave_insts_per_launch": per_launch, "valu_wave_insts_per_solve": per_launch / solves_per_launch,
                         "launches_seen": len(v)}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
