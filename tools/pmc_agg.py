#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_bwd.sh: python tools/pmc_agg.py <outdir>"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "chunk" not in k:
            continue
        k = "bwd12" if "bwd12" in k else ("fwd" if "chunk_fwd" in k else k[:40])
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v) / len(v):16.1f}   (n={len(v)})")
