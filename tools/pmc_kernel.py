#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv: usage pmc_kernel.py <csv> [name filter]"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").split("(")[0]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in sorted(v.items())})
