#!/usr/bin/env python3
"""GPU busy time of ONE steady training step from a rocprofv3 --kernel-trace CSV of bench.py: the union
of all kernel intervals (kernels on different HIP streams overlap) against the step's wall time, plus the
sum of kernel durations (= what a single stream would need)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "warp_ssim_min_fwd" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
total = sum(e - s for s, e in iv)
streams = sorted({r.get("Stream_Id", r.get("Queue_Id", "?")) for r in step})
print("step wall %.3f ms | union busy %.3f ms (%.1f %%) | sum of kernels %.3f ms | launches %d | queues %s"
      % ((t1 - t0) / 1e6, union / 1e6, 100.0 * union / (t1 - t0), total / 1e6, len(step), streams))
