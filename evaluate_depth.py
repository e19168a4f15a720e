#!/usr/bin/env python3
"""Depth evaluation with the reference's command line:

    python evaluate_depth.py --eval_mono --load_weights_folder <weights> --kt_path <kitti> --eval_split eigen
"""
from baseboostdepth_amd.evaluation import evaluate
from baseboostdepth_amd.options import MonodepthOptions

if __name__ == "__main__":
    evaluate(MonodepthOptions().parse())
