"""baseboostdepth_amd - MI355X-native photometric-reprojection training step.

Drop-in for the hot path of kieran514/baseboostdepth: `Trainer.process_batch` keeps its
signature, `layers.*` / `networks.*` keep their names, and the warp + SSIM/L1 + per-pixel-min
chain runs as hand-written gfx950 HIP kernels behind the C ABI of include/bbd_hip.h.
"""
from . import layers, networks, ops, plan, tuning  # noqa: F401
from .trainer import Trainer  # noqa: F401

__version__ = "0.1.0"
