#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("smgpu::", "").replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if k.startswith("__amd") or "k_finish" in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
