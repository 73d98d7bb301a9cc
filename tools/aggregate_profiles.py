#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/collect_profiles.sh to the two small files kept under profiles/:
<prefix>_kernel_stats.csv (the --stats table) and <prefix>_pmc.json (per-launch counter averages per kernel, with
the HBM byte totals computed as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are KiB and FETCH_SIZE
counts half of the bytes of a coalesced stream on gfx950)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src, prefix = sys.argv[1], sys.argv[2]
def newest(pattern):
    """gpurun merges a call's output INTO the local gpurun_out/: files of earlier collections may sit beside the new ones -- keep the
    newest file of every directory."""
    by_dir = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


stats = newest(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
if stats:      # keep this library's kernels only (the torch kernels of the input generation have kilobyte-long names)
    with open(stats[0]) as fh, open(prefix + "_kernel_stats.csv", "w") as out_fh:
        for i, line in enumerate(fh):
            if i == 0 or "wkv6" in line or "mask_to_lens" in line:
                out_fh.write(line)
acc = defaultdict(lambda: defaultdict(list))
for f in newest(os.path.join(src, "pmc*", "**", "*counter_collection.csv")):
    with open(f) as fh:
        per_dispatch = defaultdict(float)
        meta = {}
        for row in csv.DictReader(fh):
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            meta[row["Dispatch_Id"]] = row["Kernel_Name"]
        for (d, c), v in per_dispatch.items():
            acc[meta[d]][c].append(v)
out = {"command": "tools/collect_profiles.sh (rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 50 --warmup 10 --no-cpu; "
                  "one run per counter group, FETCH_SIZE and WRITE_SIZE in separate runs)",
       "note": "per-launch averages; FETCH_SIZE/WRITE_SIZE are KiB; read bytes = 2 * FETCH_SIZE * 1024 on gfx950", "kernels": {}}
# several template instantiations share a short name (the self-test launches a tiny state-pass variant): keep the
# instantiation with the most dispatches, i.e. the one the bench loop launches
best = {}
for k, cs in acc.items():
    short = next((n for n in ("chunk_fwd_kernel", "chunk_bwd12k_kernel") if n in k), None)
    if short is None:
        continue
    n = max(len(v) for v in cs.values())
    if short not in best or n > best[short][0]:
        best[short] = (n, k)
for short, (n, k) in best.items():
    e = out["kernels"].setdefault(short, {"instantiation": k[k.index(short):][:60], "dispatches": n, "counters": {}})
    for c, vals in acc[k].items():
        vals = sorted(vals)[len(vals) // 10:]          # drop the smallest tenth (self-test sized launches of the same instantiation)
        e["counters"][c] = round(sum(vals) / len(vals), 1)
for e in out["kernels"].values():
    c = e["counters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_read_bytes"] = int(2 * c["FETCH_SIZE"] * 1024)
        e["hbm_write_bytes"] = int(c["WRITE_SIZE"] * 1024)
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in out["kernels"].items()}, indent=1))
