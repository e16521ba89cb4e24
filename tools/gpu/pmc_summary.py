#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: mean counter value per (kernel, counter)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
        acc[k][0] += float(row.get("Counter_Value", 0)); acc[k][1] += 1
for (kern, ctr), (s, n) in sorted(acc.items()):
    print(f"{kern:60s} {ctr:28s} mean={s / n:16.1f} n={n}")
