#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace CSV and prints, per kernel: launches, mean duration, and the
GPU's busy time (union of kernel intervals) / summed kernel time, i.e. how much the streams overlap.
usage: trace_overlap.py <trace_kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]  # steady state only
agg = defaultdict(lambda: [0, 0])
iv = []
queues = defaultdict(int)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void ", "").split("(")[0]
    agg[n][0] += 1
    agg[n][1] += e - s
    iv.append((s, e))
    queues[r["Queue_Id"]] += e - s
iv.sort()
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = max(e for _, e in iv) - iv[0][0]
total = sum(v[1] for v in agg.values())
print("span %.3f ms  busy(union) %.3f ms  sum of kernels %.3f ms  overlap factor %.2f" % (span / 1e6, busy / 1e6, total / 1e6, total / max(busy, 1)))
print("per queue kernel ms:", {q: round(v / 1e6, 3) for q, v in queues.items()})
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print("%-28s n=%5d mean %9.1f us  total %8.3f ms (%4.1f%%)" % (n[:28], c, t / c / 1e3, t / 1e6, 100.0 * t / total))
