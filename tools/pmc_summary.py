#!/usr/bin/env python3
"""Summarise the per-pass counter CSVs written by tools/pmc_passes.sh into one table."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for key in ("warp_ssim_min_fwd", "warp_ssim_min_bwd", "identity_loss", "disp_to_depth_fwd", "disp_to_depth_bwd"):
            if key in name:
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in vals.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print("   %-26s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
