#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_traffic.json:
HBM bytes per launch and per solve for each engine kernel.  FETCH_SIZE/WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so the read side is doubled as that guide prescribes."""
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
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            rows.append((row["Kernel_Name"].split("(")[0].strip(), row["Counter_Name"], int(row["Grid_Size"]), float(row["Counter_Value"])))
gmax = defaultdict(int)
for k, c, g, v in rows:
    gmax[k] = max(gmax[k], g)
acc = defaultdict(lambda: defaultdict(list))
for k, c, g, v in rows:
    if g == gmax[k]:
        acc[k][c].append(v)
res = {"kernel_source_hash": kernel_source_hash(), "solves_per_launch": solves_per_launch, "nstr": nstr, "nlyr": nlyr, "note": "FETCH_SIZE doubled (gfx950 64B-per-128B tally); KB*1024", "kernels": {}}
for k, d in acc.items():
    if "sbd::" not in k:
        continue
    rd = 2.0 * 1024.0 * sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
    wr = 1024.0 * sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    res["kernels"][k] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                         "bytes_per_launch": rd + wr, "bytes_per_solve": (rd + wr) / solves_per_launch}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
