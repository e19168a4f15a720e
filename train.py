#!/usr/bin/env python3
"""Training entry point with the reference's command line (`python train.py --rand --trimin ...`).

    python train.py --kt_path /data/kitti --rand --trimin --decomp --incremental_skip --partial_skip \
                    --naive_mix --kt --weights_init scratch
    python train.py --synthetic --num_epochs 1          # no dataset: KITTI-shaped synthetic batches

Multi-GPU: `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py ...`
(one process per GPU, gradients averaged over RCCL).
"""
import os
import random

import numpy as np
import torch

from baseboostdepth_amd import Trainer, distributed, synthetic
from baseboostdepth_amd.options import MonodepthOptions


def seed_everything(seed):
    seed = seed or 1
    for fn in (torch.manual_seed, np.random.seed, random.seed):
        fn(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None):
    opts = MonodepthOptions().parse(argv)
    rank, world = 0, int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        rank, local, world = distributed.init_from_env()
        opts.cuda = local
    seed_everything(opts.pytorch_random_seed + rank)
    trainer = Trainer(opts)
    if world > 1:
        distributed.attach(trainer)
    if opts.synthetic:
        steps = int(os.environ.get("BBD_SYNTH_STEPS", "50"))
        trainer.train(lambda epoch: synthetic.synthetic_loader(
            opts.batch_size, steps, opts.height, opts.width, opts.scales, device=trainer.device,
            seed=opts.pytorch_random_seed + rank, trimin=opts.trimin, epoch=epoch))
    else:
        trainer.train()
    return trainer


if __name__ == "__main__":
    main()
