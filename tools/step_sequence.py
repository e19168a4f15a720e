#!/usr/bin/env python3
"""Ordered kernel sequence of ONE steady training step out of a rocprofv3 --kernel-trace CSV (between the last two launches
of the fused forward kernel): start offset (us), duration (us), stream/queue id, kernel name.   usage: step_sequence.py trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "warp_ssim_min_fwd" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%9.1f %7.1f q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                     r.get("Queue_Id", "?"), r["Kernel_Name"][:110]))
