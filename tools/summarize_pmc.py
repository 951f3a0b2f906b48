"""Averages rocprofv3 --pmc counter_collection CSVs per kernel name: usage summarize_pmc.py <dir>."""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "taylor" not in k and "presplit" not in k and "wgrad" not in k:
            continue
        acc[k.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v) // 3:]  # skip warm-up launches
        print(f"   {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
