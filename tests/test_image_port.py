"""Loader image arithmetic (baseboostdepth_amd/csrc/bbd_image_math.h via the host port) pinned against
the installed Pillow: exhaustively for the per-pixel functions, on random images for the resampler and
the full ColorJitter sequence.  CPU only."""
import ctypes
import os
import sys

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from host_port import HostPortBackend  # noqa: E402
from oracle import loader_ref  # noqa: E402
from baseboostdepth_amd import imageops  # noqa: E402


@pytest.fixture(scope="module")
def port():
    return HostPortBackend()


def _all_triples():
    v = np.arange(1 << 24, dtype=np.uint32)
    return np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)


def test_rgb_hsv_luma_exhaustive(port):
    trip = _all_triples()
    out = np.empty((1 << 24) * 3, np.uint8)
    port.dll.hp_img_rgb2hsv_all(out.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(out.reshape(4096, 4096, 3), np.array(Image.fromarray(trip, "RGB").convert("HSV")))
    port.dll.hp_img_hsv2rgb_all(out.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(out.reshape(4096, 4096, 3), np.array(Image.fromarray(trip, "HSV").convert("RGB")))
    lum = np.empty(1 << 24, np.uint8)
    port.dll.hp_img_luma_all(lum.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(lum.reshape(4096, 4096), np.array(Image.fromarray(trip, "RGB").convert("L")))


def test_blend_all_operand_pairs(port):
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    im1, im2 = Image.fromarray(a, "L"), Image.fromarray(b, "L")
    rng = np.random.default_rng(0)
    out = np.empty(65536, np.uint8)
    port.dll.hp_img_blend_all.argtypes = [ctypes.c_float, ctypes.c_void_p]
    for alpha in list(rng.uniform(0.8, 1.2, 60)) + [0.0, 0.5, 0.8, 1.0, 1.2, 1.7]:
        port.dll.hp_img_blend_all(ctypes.c_float(alpha), out.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(out.reshape(256, 256), np.array(Image.blend(im1, im2, float(alpha)))), alpha


import image_checks  # noqa: E402


@pytest.mark.parametrize("h,w,oh,ow", image_checks.RESIZE_CASES)
def test_lanczos_resize_matches_pillow(port, h, w, oh, ow):
    image_checks.check_resize(imageops.ImagePipeline("cpu", backend=port), h, w, oh, ow)


def test_ragged_sizes_in_one_call_and_pyramid(port):
    image_checks.check_ragged_and_pyramid(imageops.ImagePipeline("cpu", backend=port))


def test_color_jitter_sequences_match_pillow(port):
    image_checks.check_color_jitter(imageops.ImagePipeline("cpu", backend=port))


def test_hue_offset_is_c_cast_wraparound():
    assert imageops.hue_offset(-0.05) == (256 - 12) and imageops.hue_offset(0.05) == 12 and imageops.hue_offset(0.0) == 0
