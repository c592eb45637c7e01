#!/usr/bin/env python3
"""Mean counter values per kernel from rocprofv3 counter_collection CSVs.  usage: pmc_table.py DIR [kernel-substring]"""
import csv, sys, collections, glob, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in acc.items():
    if sub in k:
        print(k)
        for c, x in sorted(v.items()): print("   %-32s %16.0f  (n=%d)" % (c, sum(x) / len(x), len(x)))
