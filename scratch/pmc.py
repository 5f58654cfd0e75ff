#!/usr/bin/env python3
"""Average PMC counter values per kernel from a rocprofv3 counter_collection.csv.  usage: pmc.py file.csv [kernel substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sub in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
