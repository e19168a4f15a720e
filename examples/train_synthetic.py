#!/usr/bin/env python3
"""Minimal training loop on synthetic KITTI-shaped batches (no dataset needed).

    python examples/train_synthetic.py --epochs 1 --steps 20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py

Mirrors the reference's `train.py` -> `Trainer(opts).train()`; `--boosted` turns on the BaseBoostDepth
recipe flags of run.sh (trimin, decomp, incremental + partial pose) and starts at epoch 10.
"""
import argparse
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baseboostdepth_amd import Trainer, distributed  # noqa: E402
from baseboostdepth_amd.synthetic import synthetic_loader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch_size", type=int, default=12)
    ap.add_argument("--boosted", action="store_true")
    ap.add_argument("--ViT", action="store_true", help="MonoViT: MPViT-small encoder + HR decoder (BASELINE configs[4])")
    a = ap.parse_args()
    rank, local, world = distributed.init_from_env()
    torch.cuda.set_device(local)
    opt = types.SimpleNamespace(
        height=192, width=640, batch_size=a.batch_size, scales=[0, 1, 2, 3], frame_ids=[0, -1, 1], min_depth=0.1,
        max_depth=100.0, disparity_smoothness=1e-3, no_ssim=False, rand=True, trimin=a.boosted, decomp=a.boosted,
        pose_error=5.5, incremental_skip=a.boosted, partial_skip=a.boosted, num_layers=18, weights_init="scratch",
        learning_rate=1e-4, no_cuda=False, cuda=local, load_weights_folder="None", log_dir="/tmp/bbd_logs",
        model_name="synthetic", num_epochs=a.epochs, save_frequency=1, ViT=a.ViT)
    trainer = Trainer(opt)
    trainer.epoch = 10 if a.boosted else 0
    distributed.attach(trainer)
    dev = torch.device("cuda", local)
    for epoch in range(trainer.epoch, trainer.epoch + a.epochs):
        trainer.epoch = epoch
        loader = synthetic_loader(a.batch_size, a.steps, device=dev, seed=42 + rank, trimin=a.boosted, epoch=epoch,
                                  scales=[0, 1, 2, 3] if epoch < 10 else [0])
        outputs, losses = trainer.run_epoch(loader)
        torch.cuda.synchronize()
        if rank == 0:
            print("epoch %d: last loss %.5f" % (epoch, float(losses["loss"].detach())))


if __name__ == "__main__":
    main()
