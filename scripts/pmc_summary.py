#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>/*/.../*_counter_collection.csv: per kernel and
counter, the mean value per dispatch.   pmc_summary.py <tag> [out.json]"""
import csv, glob, os, sys, collections, json, re

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.join("gpurun_out", f"pmc_{tag}")
SHORT = [("fwd_roles", "roles"), ("fwd_finish", "finish"), ("march_rec", "march"), ("shade_tile", "shade"), ("shade_chan", "shade"), ("render_bwd", "bwd"), ("render_fwd", "fwd"),
         ("grad_merge", "merge"), ("grad_fused", "fused"), ("grad_wide", "wide"), ("compact_rows", "compact"), ("depth_kernel", "depth"), ("exp_table", "exptab"), ("sigma_mask", "mask")]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
names = {}
for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        short = next((s for pat, s in SHORT if pat in k), None)
        if short is None:
            continue
        m = re.search(r"svoxt::(\w+<[^>]*>|\w+)", k)
        names.setdefault(short, set()).add(m.group(1) if m else k[:60])
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k in sorted(acc):
    print(f"== {k}  ({', '.join(sorted(names[k]))})")
    out[k] = {}
    for c, v in sorted(acc[k].items()):
        print(f"  {c:40s} {sum(v)/len(v):18.1f}   (n={len(v)})")
        out[k][c] = sum(v) / len(v)
if len(sys.argv) > 2:
    json.dump({"tag": tag, "units": "mean counter value per dispatch; FETCH_SIZE / WRITE_SIZE in KiB; SQ_WAVE_CYCLES, "
               "SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles (MI355X_MICROARCH.md)",
               "kernels": {k: sorted(v) for k, v in names.items()}, "counters": out}, open(sys.argv[2], "w"), indent=1)
