#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>/*/.../*_counter_collection.csv: per kernel and
counter, the mean value per dispatch."""
import csv, glob, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join("gpurun_out", f"pmc_{tag}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        short = "bwd" if "render_bwd" in k else "fwd" if "render_fwd" in k else "merge" if "grad_merge" in k \
            else "fused" if "grad_fused" in k else "compact" if "compact_rows" in k else None
        if short is None:
            continue
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k in ("fwd", "bwd", "merge", "fused", "compact"):
    if k not in acc:
        continue
    print(f"== {k}")
    out[k] = {}
    for c, v in sorted(acc[k].items()):
        print(f"  {c:40s} {sum(v)/len(v):18.1f}   (n={len(v)})")
        out[k][c] = sum(v) / len(v)
if len(sys.argv) > 2:
    import json
    json.dump({"tag": tag, "units": "mean counter value per dispatch; FETCH_SIZE / WRITE_SIZE in KiB",
               "kernels": {"fwd": "render_fwd_kernel<SH,3,9,N2,REC>", "bwd": "render_bwd_kernel<SH,3,9,N2,REPLAY[,GATHER]>",
                           "merge": "grad_merge_kernel<SH,9> (second kernel of the two-kernel backward)",
                           "fused": "grad_fused_kernel<SH,9> (list walk + merge in one kernel; bwd is then the tail-only launch)",
                           "compact": "compact_rows_kernel (128-byte-aligned gradient rows -> dense [M, K])"},
               "counters": out}, open(sys.argv[2], "w"), indent=1)
