#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a short text summary:
per-kernel stats (calls, average ns) and per-kernel PMC averages per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    if "k_warp_rows<true" in name or "k_warp_rowsILb1E" in name:
        return "k_warp_rows<stitch>"
    if "k_warp_rows" in name:
        return "k_warp_rows"
    if "k_warp<true>" in name or "k_warpILb1E" in name:
        return "k_warp<stitch>"
    for k in ("k_assemble_valu", "k_assemble_mfma4", "k_assemble_mfma", "k_solve_small", "k_eigen_denorm", "k_invert_cells", "k_cell_lut", "k_warp_coords",
              "k_warp_setup", "k_warp_fast", "k_warp", "k_flatten", "k_weights", "k_blend", "k_eq_hist", "k_eq_lut", "k_eq_apply",
              "k_ransac_hyp", "k_ransac_score", "k_ransac_select"):
        if k in name:
            return k
    return None


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Name"])
        if k:
            print(f"{k:18s} calls {row['Calls']:>5s}  avg {float(row['AverageNs'])/1e3:9.2f} us  "
                  f"min {float(row['MinNs'])/1e3:9.2f}  max {float(row['MaxNs'])/1e3:9.2f}  {row['Percentage']}%")
print("== PMC, average per dispatch ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        vals = "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items()))
        print(f"{k:18s} {vals}")
