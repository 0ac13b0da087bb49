#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per
kernel (and per grid size) over its dispatches.   python profiles/summarize_pmc.py <dir>"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = (r["Kernel_Name"].split("(")[0][:48], r["Grid_Size"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("%s grid=%s" % k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("    %-24s n=%3d mean=%16.1f" % (c, len(v), sum(v) / len(v)))
