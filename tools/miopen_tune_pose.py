#!/usr/bin/env python3
"""Record MIOpen find results for the batched pose pass at the row counts `Trainer._pose_pairs` rounds it up to.

Boosted batches change the pass's row count almost every step (mono_dataset.py:87-109); with `pose_pad_rows = 32` the
convolutions of the pose network only ever see multiples of 32 rows.  This runs the pose encoder + decoder forward and
backward at those row counts in MIOpen's find mode, against the database MIOPEN_USER_DB_PATH points to:

    export MIOPEN_USER_DB_PATH=$PWD/baseboostdepth_amd/miopen_db MIOPEN_CUSTOM_CACHE_DIR=$MIOPEN_USER_DB_PATH/cache
    python tools/miopen_tune_pose.py --rows 32 64 192 224 256 288
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[32, 64, 96, 128, 160, 192, 224, 256, 288, 320])
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-find", action="store_true", help="immediate mode: time the first sight of every row count as a run would see it")
    a = ap.parse_args()
    assert "MIOPEN_USER_DB_PATH" in os.environ, "point MIOPEN_USER_DB_PATH at the database to extend"
    from baseboostdepth_amd import networks, ops
    torch.backends.cudnn.benchmark = not a.no_find
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = networks.ResnetEncoder(18, False, num_input_images=2).to(dev).train()
    dec = networks.PoseDecoder(enc.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2).to(dev).train()
    out = {}
    for n in a.rows:
        groups = [12] * (n // 12) + ([n % 12] if n % 12 else [])
        per = []
        for it in range(3):
            x = torch.rand(n, 6, a.height, a.width, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with ops.bn_call_groups(groups[:ops.BN_MAX_GROUPS - 1] + ([sum(groups[ops.BN_MAX_GROUPS - 1:])] if len(groups) >= ops.BN_MAX_GROUPS else [])):
                aa, tt = dec([enc(x)])
            (aa.square().sum() + tt.square().sum()).backward()
            torch.cuda.synchronize()
            per.append(round(time.perf_counter() - t0, 3))
        out[n] = per
        print("rows %4d: first %.1f s, then %.3f s, %.3f s" % (n, per[0], per[1], per[2]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
