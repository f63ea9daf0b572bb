#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs that tools/gpu_profile.sh left under gpurun_out/<tag>/ into the
committed summaries under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_pmc.json           per kernel: launches, mean FETCH_SIZE / WRITE_SIZE bytes per launch
  profiles/pmc_latest.json          copy of the newest <tag>_pmc.json (bench.py reads `traffic` here)
  profiles/<tag>_sq.json, sq_latest.json   SQ counters per kernel (bench.py reads FAST's instruction count here)

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts a 128-byte
request of a wide (16 B per lane) streaming read as 64 bytes (MI355X_MICROARCH.md, HBM); kernels
listed in WIDE_READERS get the x2 correction, the others are left as counted ("uncalibrated").
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDE_READERS = ("k_knn2_mfma",)  # kernels whose global reads are 16 B per lane streaming loads: the matrix-core
                                  # matcher reads train rows and queries with global_load_dwordx4 (the popcount
                                  # kernel read its rows with scalar loads: 64-byte requests, counted as they are)


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def pmc(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = agg[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"]) * 1024.0
    return agg


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    fetch = pmc(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
    write = pmc(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
    out = {"tag": tag, "note": "bytes per launch; FETCH_SIZE x2 for wide streaming readers (gfx950), others as counted",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fn, fb = fetch.get(k, [0, 0.0])
        wn, wb = write.get(k, [0, 0.0])
        corr = 2.0 if WIDE_READERS and k.startswith(WIDE_READERS) else 1.0
        f_avg = fb / fn * corr if fn else None
        w_avg = wb / wn if wn else None
        out["kernels"][k] = {"launches": fn or wn, "fetch_bytes": f_avg, "write_bytes": w_avg,
                             "fetch_correction": corr,
                             "hbm_bytes": (f_avg or 0.0) + (w_avg or 0.0)}
    p = os.path.join(dst, tag + "_pmc.json")
    json.dump(out, open(p, "w"), indent=1)
    shutil.copy(p, os.path.join(dst, "pmc_latest.json"))
    for k, v in out["kernels"].items():
        print("%-28s n=%4d fetch=%12.0f write=%12.0f" % (k, v["launches"], v["fetch_bytes"] or 0, v["write_bytes"] or 0))
    # SQ pass: per kernel means per launch (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles)
    sq_path = os.path.join(src, "pmc_sq", "sq_counter_collection.csv")
    if os.path.exists(sq_path):
        agg = defaultdict(lambda: defaultdict(list))
        with open(sq_path) as f:
            for row in csv.DictReader(f):
                agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        sq = {"tag": tag, "note": "means per launch; SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* are quad-cycles "
                                  "summed over all waves, SQ_BUSY_CYCLES is summed over the 32 shader engines",
              "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(agg.items())}}
        json.dump(sq, open(os.path.join(dst, tag + "_sq.json"), "w"), indent=1)
        shutil.copy(os.path.join(dst, tag + "_sq.json"), os.path.join(dst, "sq_latest.json"))  # bench.py reads it
        for k in ("k_knn2<8>", "k_fast_nms"):
            if k in sq["kernels"]:
                print(k, {c: round(v) for c, v in sq["kernels"][k].items()})


if __name__ == "__main__":
    main()
