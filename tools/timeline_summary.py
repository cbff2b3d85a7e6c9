"""Compact timeline of the LAST host-entry-point step from a rocprofv3 --kernel-trace --memory-copy-trace csv directory."""
import csv, glob, sys, os
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:40], r.get("Queue_Id", "")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
ev.sort()
# last step: find the last setup_kernel group of 6
setups = [i for i, e in enumerate(ev) if "setup_kernel" in e[3]]
i0 = setups[-int(sys.argv[2]) if len(sys.argv) > 2 else -6]
# include copies shortly before
t0 = ev[i0][0]
j = i0
while j > 0 and ev[j - 1][2] == "C" and t0 - ev[j - 1][0] < 3_000_000:
    j -= 1
t0 = ev[j][0]
for e in ev[j:]:
    print(f"{(e[0]-t0)/1e6:8.3f} {(e[1]-t0)/1e6:8.3f} {(e[1]-e[0])/1e3:8.1f}us {e[2]} {e[3]} {e[4]}")
