#!/usr/bin/env python3
"""Expected padding of the batched pose pass by epoch: rows the batches' frame sets ask for (draws from the loader's offset
distributions, SURVEY 8d / mono_dataset.py:87-109, batch 12, trimin + decomp + incremental + partial) against the rows that
run after rounding up to a row count the shipped MIOpen database has find results for (tuning.POSE_ROW_COUNTS).  CPU only.
    python tools/pose_row_buckets.py [--draws 400] [--without 40 216 ...]"""
import argparse
import os
import random
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseboostdepth_amd import steptables, tuning  # noqa: E402
from baseboostdepth_amd.plan import ReprojectionPlan, STEREO  # noqa: E402

P = {10: [.108, .287, .277, .135, .068, .040, .084], 15: [.050, .050, .077, .094, .139, .142, .448],
     19: [.050, .050, .050, .059, .070, .078, .644]}


def weights(epoch):
    if epoch <= 10:
        return P[10]
    if epoch >= 19:
        return P[19]
    lo, hi = (10, 15) if epoch < 15 else (15, 19)
    t = (epoch - lo) / (hi - lo)
    return [a * (1 - t) + b * t for a, b in zip(P[lo], P[hi])]


def rows_of(ms, maxing):
    ms = sorted(ms, reverse=True)
    plan = ReprojectionPlan([[0, STEREO] if m == 0 else [0, m, -m] for m in ms], True, True)
    M = max(ms)
    frames = list(range(-M, M + 1)) if M else [0]
    if min(ms) < 3:
        frames.append(STEREO)
    fid = sorted(frames, key=lambda f: float("inf") if f == STEREO else abs(f))
    return steptables.PoseSchedule(plan, fid, maxing, maxing, True, 1 << 30).total_rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=400)
    ap.add_argument("--without", type=int, nargs="*", default=[], help="row counts to leave out (what an earlier database had)")
    a = ap.parse_args()
    table = tuple(r for r in tuning.POSE_ROW_COUNTS if r not in a.without)

    def pad(n):
        q = -(-n // 32) * 32
        return next((r for r in table if n <= r <= q), q)
    rnd = random.Random(5)
    print("row counts with find results:", list(table))
    print("%-28s %10s %10s %10s %8s" % ("phase", "rows asked", "rows run", "padding", "buckets"))
    for label, maxing, w, lo in [("epoch 5 (early curriculum)", False, [.062, .573, .366], 0)] + \
            [("epoch %d" % e, True, weights(e), 1) for e in (10, 11, 12, 13, 15, 17, 19)]:
        rr = [rows_of(rnd.choices(range(lo, lo + len(w)), w, k=12), maxing) for _ in range(a.draws)]
        run = [pad(max(r, 1)) for r in rr]
        print("%-28s %10.1f %10.1f %9.1f%% %8d" % (label, statistics.mean(rr), statistics.mean(run),
                                                   100 * (sum(run) / max(sum(rr), 1) - 1), len(set(run))))


if __name__ == "__main__":
    main()
