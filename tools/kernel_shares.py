#!/usr/bin/env python3
"""Per-kernel shares of a rocprofv3 --kernel-trace --stats run, grouped into what the DP LoRA step consists of:
    python tools/kernel_shares.py <dir with *kernel_stats.csv> [> profiles/...]"""
import csv
import glob
import sys

GROUPS = [("WKV6 forward", ("chunk_fwd_kernel",)), ("WKV6 backward", ("chunk_bwd", "scan_bwd")),
          ("token shift / ddlerp (HIP)", ("ddlerp",)), ("GroupNorm * gate (HIP)", ("gn_gate",)),
          ("channel-mix glue (HIP)", ("sqrelu", "sigmul")),
          ("GEMM (rocBLAS / hipBLASLt)", ("Cijk", "gemm", "Gemm", "GEMM")), ("optimizer", ("adam", "Adam", "multi_tensor")),
          ("RCCL", ("nccl", "rccl")), ("layer norm", ("layer_norm", "LayerNorm", "layernorm")),
          ("softmax / loss", ("softmax", "log_softmax", "nll")), ("embedding / gather / scatter", ("embedding", "index", "gather", "scatter")),
          ("copies / casts / fills", ("copy", "Copy", "fill", "Fill", "cast")), ("reductions", ("reduce", "Reduce", "sum")),
          ("other elementwise (eager torch)", ("elementwise", "vectorized", "Elementwise"))]
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
acc = {}
for r in rows:
    name = r["Name"]
    grp = next((g for g, keys in GROUPS if any(k in name for k in keys)), "other")
    a = acc.setdefault(grp, [0.0, 0, []])
    a[0] += float(r["TotalDurationNs"]); a[1] += int(r["Calls"]); a[2].append((float(r["TotalDurationNs"]), name))
print(f"total kernel time {tot / 1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for grp, (ns, calls, names) in sorted(acc.items(), key=lambda t: -t[1][0]):
    print(f"  {100 * ns / tot:5.1f} %  {ns / 1e6:9.2f} ms  {calls:7d} launches  {grp}")
    for d, n in sorted(names, reverse=True)[:3]:
        print(f"             {100 * d / tot:5.1f} %  {n[:110]}")
