#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, counter sums over dispatches."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    if "sbd" not in k and "accum" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        print(f"   {c:28s} sum={acc[k][c]:.6g}  dispatches={cnt[k][c]}  per-dispatch={acc[k][c]/cnt[k][c]:.6g}")
