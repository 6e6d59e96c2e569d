#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name.
usage: pmc_summary.py <dir with *counter_collection.csv> [out.csv]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
names = sorted({c for k in acc for c in acc[k]})
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out)
w.writerow(["kernel", "launches"] + names)
for k in sorted(acc):
    n = max(calls[k].values())
    w.writerow([k, n] + ["%.0f" % (acc[k].get(c, 0.0) / max(1, calls[k].get(c, 1))) for c in names])
