"""GPU tier of the loader image kernels (bbd_resample_*_u8, bbd_color_jitter_u8, bbd_u8_to_float_chw)
through the C ABI: bit-exact against Pillow, same checks as the CPU host-port tier."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_checks  # noqa: E402
from baseboostdepth_amd import imageops  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pipe():
    return imageops.ImagePipeline("cuda:0")


@pytest.mark.parametrize("h,w,oh,ow", image_checks.RESIZE_CASES)
def test_lanczos_resize_matches_pillow(pipe, h, w, oh, ow):
    image_checks.check_resize(pipe, h, w, oh, ow)


def test_ragged_sizes_in_one_call_and_pyramid(pipe):
    image_checks.check_ragged_and_pyramid(pipe)


def test_color_jitter_sequences_match_pillow(pipe):
    image_checks.check_color_jitter(pipe)
    image_checks.check_color_jitter(pipe, H=192, W=640, seed=3)      # BASELINE size, multi-block reduction


def test_jitter_covers_every_rgb_triple(pipe):
    """All 2^24 RGB values through hue and saturation on the device vs Pillow (a 4096x4096 image)."""
    from oracle import loader_ref
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(1, 4096, 4096, 3)
    seq = [(imageops.HUE, 0.07), (imageops.SATURATION, 1.13), (imageops.BRIGHTNESS, 0.9), (imageops.CONTRAST, 1.1)]
    dst = torch.zeros(1, 3, 4096, 4096, device=pipe.device)
    pipe.jitter_to_float(torch.from_numpy(img).to(pipe.device), [0], [seq], dst, [0])
    pipe.flush()
    want = loader_ref.to_tensor(loader_ref.color_jitter(img[0], seq))
    assert torch.equal(dst[0].cpu(), want)
