#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv: per kernel calls/step, average us, share of the total.  usage: kstats.py file.csv steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time per step: %.2f ms" % (tot / 1e6 / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-62s %6.1f/step avg %8.1f us  %6.3f ms/step %5.1f%%" % (r['Name'][:62], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3,
                                                                     float(r['TotalDurationNs']) / 1e6 / steps, 100 * float(r['TotalDurationNs']) / tot))
