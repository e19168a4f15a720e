#!/usr/bin/env python3
"""Fills the @@PLACEHOLDER@@ fields of DESIGN.md / FINDINGS.md / README.md / profiles/r06/README.md from the round's final bench
line (profiles/r06/bench_default_final.json) - the documents quote the driver's command, not numbers typed by hand.
    python tools/fill_round6_docs.py [--check]      (--check: only report placeholders that would stay unfilled)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    line = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_default_final.json")))
    sec = {s["config"]: s for s in line.get("secondary", [])}
    b15, t5 = sec["boosted15_fresh"], sec["trimin5_fresh"]
    p15, p5 = b15["passes"], t5["passes"]
    fz15, fz5 = sec["boosted15"], sec["trimin5"]
    r = lambda a, b: "%.3f" % (a / b)
    slow = lambda p: "%.0f / %s ms" % (p["host_call_ms_median"], "–".join("%.0f" % v for v in (min(p["host_call_ms_slowest3"]), max(p["host_call_ms_slowest3"]))))
    rf = line["roofline"]
    how = rf.get("peak_measured_how", {})
    vals = {
        "GRAPHS15": str(b15.get("step_graphs_in_use")),
        "HOST15": "%.0f" % p15[2]["host_call_ms_median"],
        "FRESH15": "%.3f" % b15["vs_frozen_batch"],
        "WORK15": "%.1f %%" % (100 * (p15[1]["pose_rows_run_mean"] / b15["frozen_batch_pose_rows"] - 1)),
        "COLD15": "%.3f" % b15["vs_frozen_batch_cold_start"],
        "COLD5": "%.3f" % t5["vs_frozen_batch_cold_start"],
        "SLOW15": "%.0f–%.0f" % (min(p15[1]["host_call_ms_slowest3"]), max(p15[1]["host_call_ms_slowest3"])),
        "B15FROZEN": "%.1f" % fz15["ms_per_step"], "T5FROZEN": "%.2f" % fz5["ms_per_step"],
        "PRE15": "%d buckets, %.1f s" % (b15["prewarm"]["buckets"], b15["prewarm"]["seconds"]),
        "PRE5": "%d buckets, %.1f s" % (t5["prewarm"]["buckets"], t5["prewarm"]["seconds"]),
        "B15COLD": "%.1f" % p15[0]["ms_per_step"], "B15COLDR": r(fz15["ms_per_step"], p15[0]["ms_per_step"]), "B15COLDH": slow(p15[0]),
        "B15STEADY": "%.1f" % p15[1]["ms_per_step"], "B15STEADYR": r(fz15["ms_per_step"], p15[1]["ms_per_step"]), "B15STEADYH": slow(p15[1]),
        "B15SEEN": "%.1f / %.1f" % (p15[2]["ms_per_step"], p15[3]["ms_per_step"]),
        "B15SEENR": "%s / %s" % (r(fz15["ms_per_step"], p15[2]["ms_per_step"]), r(fz15["ms_per_step"], p15[3]["ms_per_step"])),
        "T5COLD": "%.2f" % p5[0]["ms_per_step"], "T5COLDR": r(fz5["ms_per_step"], p5[0]["ms_per_step"]), "T5COLDH": slow(p5[0]),
        "T5SEEN": "%.2f" % p5[2]["ms_per_step"], "T5SEENR": r(fz5["ms_per_step"], p5[2]["ms_per_step"]),
        "COPY": " / ".join("%.0f" % v for v in how.get("GBps_by_loads_in_flight", {}).values()) or str(rf.get("peak_measured")),
        "MD2": "%.1f" % line["value"], "MD2MS": "%.2f" % line["ms_per_step"],
        "PEAKM": "%.0f" % rf["peak_measured"], "FRACM": "%.3f" % rf["frac_of_measured"], "FRAC": "%.3f" % rf["frac"],
        "FIB": "%.2f" % rf["frac_of_issue_bound"],
        "BWDUS": "%.1f" % (1e3 * line["kernels"]["bbd_warp_ssim_min_disp_bwd"]["mean_ms"]),
        "FWDUS": "%.1f" % (1e3 * line["kernels"]["bbd_warp_ssim_min_disp_fwd"]["mean_ms"]),
        "IDUS": "%.1f" % (1e3 * line["kernels"]["bbd_identity_loss_fwd"]["mean_ms"]),
        "CPU": "%.2f" % line["cpu_baseline"]["value"], "CPUMODEL": line["cpu_baseline"].get("cpu_model", "?"),
        "CPUCORES": str(line["cpu_baseline"]["cores"]), "CPU1": str(line["cpu_baseline"].get("one_thread_images_per_sec")),
        "EAGER": str(line.get("eager_step_images_per_sec")),
        "BOOSTED": "%.1f" % sec["boosted"]["value"], "BOOSTED15": "%.1f" % fz15["value"], "COHERENT": "%.1f" % sec["boosted15_coherent"]["value"],
        "TRIMIN5": "%.1f" % fz5["value"], "VIT": "%.1f" % sec["vit"]["value"],
        "LOADER": "%.1f" % sec["md2_loader"]["value"], "LOADERR": "%.3f" % sec["md2_loader"]["vs_frozen_batch"],
        "B15FRESH": "%.1f" % b15["value"], "T5FRESH": "%.1f" % t5["value"], "T5FRESHR": "%.3f" % t5["vs_frozen_batch"],
        "SECS": "%.0f" % line.get("secondary_seconds", 0),
    }
    extra = os.path.join(ROOT, "profiles", "r06", "doc_values.json")
    if os.path.isfile(extra):
        vals.update(json.load(open(extra)))
    left = {}
    for name in ("DESIGN.md", "FINDINGS.md", "README.md", os.path.join("profiles", "r06", "README.md")):
        path = os.path.join(ROOT, name)
        if not os.path.isfile(path):
            continue
        text = open(path).read()
        new = re.sub(r"@@([A-Z0-9]+)@@", lambda m: vals.get(m.group(1), m.group(0)), text)
        rest = sorted(set(re.findall(r"@@([A-Z0-9]+)@@", new)))
        if rest:
            left[name] = rest
        if "--check" not in sys.argv and new != text:
            open(path, "w").write(new)
    print("unfilled:", left or "none")


if __name__ == "__main__":
    main()
