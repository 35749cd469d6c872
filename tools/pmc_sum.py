#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel: python3 tools/pmc_sum.py <counter_collection.csv> [divide_by]"""
import collections
import csv
import sys

div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print("    %-24s %.1f" % (c, x / div))
