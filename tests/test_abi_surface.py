"""CPU tier: the C-ABI library loads (no GPU needed) and exports exactly what include/bbd_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bbd_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(bbd_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from baseboostdepth_amd.csrc.build import build
    return build()


def test_header_declares_the_hot_path_entry_points():
    names = declared_functions()
    for must in ("bbd_identity_loss_fwd", "bbd_warp_ssim_min_fwd", "bbd_warp_ssim_min_bwd",
                 "bbd_disp_to_depth_fwd", "bbd_disp_to_depth_bwd"):
        assert must in names


def test_library_exports_every_declared_symbol(lib_path):
    dll = ctypes.CDLL(lib_path)
    for name in declared_functions():
        assert hasattr(dll, name), "libbbd_hip.so lacks %s" % name


def test_binding_signatures_cover_the_header(lib_path):
    from baseboostdepth_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    lib = _lib.get_lib()
    assert lib.tile_w == 64 and lib.tile_h == 16
    assert lib.num_tiles(192, 640) == 120
    assert _lib.POSE_STRIDE == 40 and _lib.MAX_CAND == 20


def test_constants_match_header():
    from baseboostdepth_amd import _lib
    text = open(HEADER).read()
    for macro, val in (("BBD_MAX_FRAME_SLOTS", _lib.MAX_FRAME_SLOTS), ("BBD_MAX_CAND", _lib.MAX_CAND),
                       ("BBD_POSE_STRIDE", _lib.POSE_STRIDE), ("BBD_ABI_VERSION", _lib.ABI_VERSION)):
        assert int(re.search(r"#define\s+%s\s+(\d+)" % macro, text).group(1)) == val


def test_product_refuses_cpu_tensors(lib_path):
    """No CPU fallback: the HIP backend raises on host tensors instead of computing anything."""
    import torch
    from baseboostdepth_amd import ops, _lib
    be = ops.HipBackend()
    with pytest.raises(_lib.BbdError):
        ops.disp_to_depth_fullres(torch.rand(1, 1, 8, 8), 16, 16, 0.1, 100.0, be)


def test_missing_library_is_loud(tmp_path):
    from baseboostdepth_amd import _lib
    with pytest.raises(_lib.BbdError):
        _lib.HipLibrary(str(tmp_path / "nope.so"))
