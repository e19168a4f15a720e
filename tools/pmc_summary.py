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

# optional second argument: write the per-kernel HBM traffic (bytes per launch) as JSON for bench.py
if len(sys.argv) > 2:
    import json
    names = {"warp_ssim_min_fwd": "bbd_warp_ssim_min_fwd", "warp_ssim_min_bwd": "bbd_warp_ssim_min_bwd",
             "identity_loss": "bbd_identity_loss_fwd"}
    out = {"_note": "rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE (tools/pmc_passes.sh); KB -> bytes "
                    "x1024; FETCH_SIZE DOUBLED (MI355X_MICROARCH.md, HBM: gfx950 tallies 128-B requests at 64 B). "
                    "Calibrated on this build's own access widths by tools/microbench/fetch_calib.hip over a 1 GiB "
                    "buffer: 4-, 8- and 16-byte-per-lane streams and the warp kernels' 8-byte texel-pair gather all "
                    "report exactly 0.5000 of their known bytes (profiles/r02/fetch_size_calibration.txt); round 1's "
                    "'not doubled' reading came from a buffer that fitted the Infinity Cache. WRITE_SIZE as reported. "
                    "Fabric-side requests: Infinity-Cache hits are counted.",
           "workload": sys.argv[3] if len(sys.argv) > 3 else "tools/kernel_bench.py"}
    for k, cs in vals.items():
        if k in names and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f = 2.0 * sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024
            w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
            out[names[k]] = {"fetch_bytes": round(f), "write_bytes": round(w), "traffic_bytes": round(f + w)}
            if "SQ_INSTS_VALU" in cs:
                out[names[k]]["valu_wave_instructions"] = round(sum(cs["SQ_INSTS_VALU"]) / len(cs["SQ_INSTS_VALU"]))
                out[names[k]]["waves"] = round(sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"]))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from baseboostdepth_amd.csrc.build import source_sha16
    out["kernel_source_sha16"] = source_sha16()     # bench.py: "kernels_constants_stale" when the shipped source differs
    json.dump(out, open(sys.argv[2], "w"), indent=1)
