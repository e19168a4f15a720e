#!/usr/bin/env python3
"""Summarise ONE steady-state training step out of a rocprofv3 --kernel-trace CSV of bench.py:
kernels between the last two launches of the fused forward kernel."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "warp_ssim_min_fwd" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
wall = (int(rows[b]["Start_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"]][0] += 1
    agg[r["Kernel_Name"]][1] += d
busy = sum(v[1] for v in agg.values())
out = csv.writer(sys.stdout)
out.writerow(["kernel", "calls", "total_us", "avg_us", "pct_of_busy"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.writerow([k[:150], v[0], round(v[1] / 1e3, 2), round(v[1] / 1e3 / v[0], 2), round(100 * v[1] / busy, 2)])
out.writerow(["TOTAL_BUSY (wall %.3f ms, %d launches)" % (wall, len(step)), len(step), round(busy / 1e3, 2), "", 100])
