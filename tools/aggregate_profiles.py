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


def timed_rows(trace_csv, steps):
    """Per-kernel statistics over the TIMED launches of a bench.py run under rocprofv3 --kernel-trace: bench.py brackets its W warm-up + K
    timed steps with two wkv6::pass_marker_kernel launches when it runs under a profiler; the timed launches of a kernel are the last
    `steps` x (launches per pass) of its dispatches between the markers.  Rows in rocprofv3's own --stats format."""
    with open(trace_csv) as fh:
        rows = [r for r in csv.DictReader(fh)]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [int(r["Dispatch_Id"]) for r in rows if "pass_marker_kernel" in r["Kernel_Name"]]
    if len(marks) < 2:
        return None
    lo, hi = marks[-2], marks[-1]
    by_kernel = defaultdict(list)
    for r in rows:
        d = int(r["Dispatch_Id"])
        if lo < d < hi and "wkv6::" in r["Kernel_Name"]:
            by_kernel[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out, total = [], 0
    for k, durs in by_kernel.items():
        per_step = max(1, round(len(durs) / (steps["steps"] + steps["warmup"])))
        durs = durs[-steps["steps"] * per_step:]
        total += sum(durs)
        mean = sum(durs) / len(durs)
        sd = (sum((x - mean) ** 2 for x in durs) / max(1, len(durs) - 1)) ** 0.5
        out.append([k, len(durs), sum(durs), mean, 0.0, min(durs), max(durs), sd])
    for r in out:
        r[4] = 100.0 * r[2] / total
    return sorted(out, key=lambda r: -r[2])


STEPS = {"steps": int(os.environ.get("PROFILE_STEPS", "50")), "warmup": int(os.environ.get("PROFILE_WARMUP", "10"))}   # collect_profiles.sh's command line
stats = newest(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
if stats:      # keep this library's kernels only (the torch kernels of the input generation have kilobyte-long names)
    # Two scopes in one table.  "timed": the K timed steps only (between bench.py's pass markers) -- what bench.py's HIP events and the
    # driver measure.  "all": rocprofv3's own --stats rows over the whole process -- pre-warm launches at ramping clocks and the
    # library's self-test launches (13 us) included; kept for the cross-check, NOT comparable with the bench line.
    trace = newest(os.path.join(src, "stats", "**", "*kernel_trace.csv"))
    timed = timed_rows(trace[0], STEPS) if trace else None
    with open(stats[0]) as fh, open(prefix + "_kernel_stats.csv", "w") as out_fh:
        w = csv.writer(out_fh, quoting=csv.QUOTE_NONNUMERIC)
        for i, row in enumerate(csv.reader(fh)):
            if i == 0:
                w.writerow(["Scope"] + row)
                for r in timed or []:
                    w.writerow([f"timed ({STEPS['steps']} steps between the pass markers)", r[0], r[1], r[2], round(r[3], 1), round(r[4], 2), r[5], r[6], round(r[7], 1)])
            elif "wkv6" in row[0] or "mask_to_lens" in row[0]:
                w.writerow(["all launches of the process (rocprofv3 --stats)"] + row)
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
