"""ctypes binding of the C-ABI shared library (include/bbd_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `baseboostdepth_amd/csrc/build.py`
(`hipcc --offload-arch=gfx950`).  There is NO fallback: if the library is missing or a call
fails, a RuntimeError is raised - the product path never silently runs anything else.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: it owns the process's libamdhip64.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
# BBD_HIP_LIB lets the tuning scripts load an alternative build of the SAME sources (tools/variants.sh)
LIB_PATH = os.environ.get("BBD_HIP_LIB", os.path.join(_HERE, "csrc", "libbbd_hip.so"))

MAX_FRAME_SLOTS = 16
MAX_CAND = 20
POSE_STRIDE = 40
PROJ_STRIDE = 24
KIND_WARP, KIND_IDENT, FLAG_NO_POSE_GRAD = 0, 1, 0x100
COMPOSE_STRIDE, COMPOSE_ERROR, COMPOSE_REPLACE = 12, 1, 2
PAIR_SHIFT = 16        # bits 16-23 of bbd_cand_t.kind: 1 + index of the pass partner (hint), 0 = none
ABI_VERSION = 7

_p = ctypes.c_void_p
_i = ctypes.c_int
_d = ctypes.c_double

# name -> argtypes, exactly the prototypes of include/bbd_hip.h
SIGNATURES = {
    "bbd_abi_version": [],
    "bbd_tile_w": [],
    "bbd_tile_h": [],
    "bbd_num_tiles": [_i, _i],
    "bbd_num_tiles_fwd": [_i, _i],
    "bbd_num_tiles_bwd": [_i, _i],
    "bbd_pose_expand": [_p, _p, _i, _p],
    "bbd_identity_loss_fwd": [_p, _p, _p, _i, _p, _i, _i, _i, _p],
    "bbd_identity_loss_grouped_fwd": [_p, _p, _p, _p, _i, _p, _i, _i, _i, _p],
    "bbd_warp_ssim_min_fwd": [_p] * 12 + [_i] * 6 + [_p],
    "bbd_warp_ssim_min_bwd": [_p] * 10 + [_i] * 6 + [_p],
    "bbd_fused_work_items": [_i, _i, _i, _i, _i, _p, _p],
    "bbd_warp_ssim_min_disp_fwd": [_p, _p, _p, _p, _d, _d] + [_p] * 11 + [_i] * 6 + [_p],
    "bbd_warp_ssim_min_disp_bwd": [_p, _p, _p, _p, _d, _d] + [_p] * 9 + [_i] * 6 + [_p],
    "bbd_disp_upsample_adjoint": [_p, _p, _p, _i, _i, _i, _i, _p],
    "bbd_disp_to_depth_fwd": [_p, _p, _i, _i, _i, _i, _i, _d, _d, _p],
    "bbd_disp_to_depth_bwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _d, _p],
    "bbd_pose_matrix_fwd": [_p, _p, _p, _i, _i, _p, _p],
    "bbd_pose_matrix_bwd": [_p, _p, _p, _p, _p, _i, _i, _p, _p],
    "bbd_smooth_loss_multi_fwd": [_p, _p, _p, _p, _p, _i, _i, _p],
    "bbd_smooth_loss_multi_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "bbd_pose_compose_fwd": [_p, _p, _p, _i, _d, _p],
    "bbd_pose_compose_bwd": [_p, _p, _p, _p, _p, _p, _i, _p],
    "bbd_smooth_chunks": [],
    "bbd_smooth_loss_fwd": [_p, _p, _p, _p, _i, _i, _i, _p],
    "bbd_smooth_loss_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "bbd_backproject_fwd": [_p, _p, _p, _i, _i, _i, _p],
    "bbd_project3d_fwd": [_p, _p, _p, _p, _i, _i, _i, _d, _p],
    "bbd_ssim_fwd": [_p, _p, _p, _i, _i, _i, _p],
    "bbd_backproject_bwd": [_p, _p, _p, _i, _i, _i, _p],
    "bbd_project3d_bwd_blocks": [],
    "bbd_project3d_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _d, _p],
    "bbd_ssim_bwd": [_p, _p, _p, _p, _i, _i, _i, _p],
    "bbd_depth_metrics": [_p, _p, _p, _p, _i, _i, _i, _d, _d, _d, _d, _d, _i, _p],
    "bbd_resample_h_u8": [_p, _p, _p, _i, _i, _p, _p, _i, _p],
    "bbd_resample_v_u8": [_p, _p, _p, _i, _i, _i, _p, _p, _i, _p],
    "bbd_color_jitter_u8": [_p, _p, _p, _i, _i, _i, _p, _p],
    "bbd_u8_to_float_chw": [_p, _p, _p, _i, _i, _i, _p],
    "bbd_bn_scratch_doubles": [_i, _i, _i],
    "bbd_bn_act_fwd": [_p] * 11 + [_i, _i, _i, _d, _d, _i, _p],
    "bbd_bn_act_bwd": [_p] * 12 + [_i, _i, _i, _i, _p],
    "bbd_bn_grouped_scratch_doubles": [_i, _i, _i, _i],
    "bbd_bn_act_grouped_fwd": [_p] * 12 + [_i, _i, _i, _i, _i, _d, _d, _i, _p],
    "bbd_bn_act_grouped_bwd": [_p] * 13 + [_i, _i, _i, _i, _i, _p],
    "bbd_bn_act_grouped_dev_fwd": [_p] * 12 + [_i, _i, _i, _i, _i, _d, _d, _i, _p],
    "bbd_bn_act_grouped_dev_bwd": [_p] * 13 + [_i, _i, _i, _i, _i, _i, _p],
    "bbd_reflect_pad1_fwd": [_p, _p, _i, _i, _i, _p],
    "bbd_reflect_pad1_bwd": [_p, _p, _i, _i, _i, _p],
    "bbd_maxpool3s2_fwd": [_p, _p, _p, _i, _i, _i, _p],
    "bbd_maxpool3s2_bwd": [_p, _p, _p, _i, _i, _i, _p],
    "bbd_dispconv_scratch_doubles": [_i],
    "bbd_dispconv_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "bbd_dispconv_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "bbd_upcat_pad1_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "bbd_upcat_pad1_bwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "bbd_bias_elu_scratch_doubles": [_i, _i, _i],
    "bbd_bias_elu_fwd": [_p, _p, _i, _i, _i, _p],
    "bbd_bias_elu_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "bbd_selftest_div": [_i, _i, ctypes.c_uint, _p, _p],
    "bbd_stream_copy": [_p, _p, ctypes.c_long, _i, _p],
    "bbd_gather_pairs": [_p, _p, _p, _p, _i, ctypes.c_long, _d, _d, _p],
    "bbd_dwconv_tokens_fwd": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "bbd_factor_att_supported": [_i, _i],
    "bbd_factor_att_segments": [_i, _i],
    "bbd_factor_att_scratch_floats": [_i, _i, _i, _i],
    "bbd_factor_att_fwd": [_p] * 7 + [_i, _i, _i, _i, _d, _p],
    "bbd_factor_att_bwd": [_p] * 10 + [_i, _i, _i, _i, _d, _p],
    "bbd_colsum_scratch_floats": [ctypes.c_long, _i],
    "bbd_colsum": [_p, _p, _p, ctypes.c_long, _i, _p],
    "bbd_token_ln_supported": [_i],
    "bbd_token_ln_scratch_floats": [_i, _i],
    "bbd_token_ln_fwd": [_p] * 8 + [_i, _i, _i, _d, _p],
    "bbd_token_ln_bwd": [_p] * 11 + [_i, _i, _i, _p],
    "bbd_dwconv_tokens_groups_fwd": [_p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "bbd_dwconv_tokens_groups_wgrad_scratch_floats": [_i, _i, _i, _i, _p, _p],
    "bbd_dwconv_tokens_groups_wgrad": [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "bbd_dwconv_tokens_wgrad_scratch_floats": [_i, _i, _i, _i, _i],
    "bbd_dwconv_tokens_wgrad": [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
}
RESAMPLE_JOB, RESAMPLE_FLIP, JITTER_JOB, CONVERT_JOB = 12, 1, 12, 4
EVAL_DESC, EVAL_OUT = 8, 12
EVAL_PRED_IS_DISP, EVAL_MEDIAN_MIDPOINT, EVAL_NO_MEDIAN_SCALING = 1, 2, 4


class BbdError(RuntimeError):
    pass


class HipLibrary:
    """Thin typed wrapper; every call checks the int status the ABI returns."""

    def __init__(self, path=LIB_PATH):
        if not os.path.isfile(path):
            raise BbdError(
                "HIP extension %s not found - run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
        self.path = path
        self._dll = ctypes.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(self._dll, name)   # AttributeError here = header/library mismatch
            fn.argtypes = argtypes
            fn.restype = ctypes.c_long if name.endswith("_scratch_floats") else _i
        if self._dll.bbd_abi_version() != ABI_VERSION:
            raise BbdError("libbbd_hip.so ABI version mismatch")
        self.smooth_chunks = self._dll.bbd_smooth_chunks()
        self.tile_w = self._dll.bbd_tile_w()
        self.tile_h = self._dll.bbd_tile_h()
        self.project3d_bwd_blocks = self._dll.bbd_project3d_bwd_blocks()

    def num_tiles(self, H, W):
        return self._dll.bbd_num_tiles(H, W)

    def num_tiles_fwd(self, H, W):
        return self._dll.bbd_num_tiles_fwd(H, W)

    def num_tiles_bwd(self, H, W):
        return self._dll.bbd_num_tiles_bwd(H, W)

    def bn_scratch_doubles(self, N, C, HW):
        return self._dll.bbd_bn_scratch_doubles(N, C, HW)

    def bn_grouped_scratch_doubles(self, max_rows, G, C, HW):
        return self._dll.bbd_bn_grouped_scratch_doubles(max_rows, G, C, HW)

    def dwconv_groups_wgrad_scratch_floats(self, B, H, W, n, cn, k):
        return self._dll.bbd_dwconv_tokens_groups_wgrad_scratch_floats(B, H, W, n, cn, k)

    def dwconv_wgrad_scratch_floats(self, B, H, W, C, k):
        return self._dll.bbd_dwconv_tokens_wgrad_scratch_floats(B, H, W, C, k)

    def colsum_scratch_floats(self, rows, C):
        return self._dll.bbd_colsum_scratch_floats(rows, C)

    def token_ln_supported(self, C):
        return bool(self._dll.bbd_token_ln_supported(C))

    def token_ln_scratch_floats(self, rows, C):
        return self._dll.bbd_token_ln_scratch_floats(rows, C)

    def factor_att_supported(self, C, Ch):
        return bool(self._dll.bbd_factor_att_supported(C, Ch))

    def factor_att_scratch_floats(self, B, N, C, Ch):
        return self._dll.bbd_factor_att_scratch_floats(B, N, C, Ch)

    def bias_elu_scratch_doubles(self, N, C, HW):
        return self._dll.bbd_bias_elu_scratch_doubles(N, C, HW)

    def dispconv_scratch_doubles(self, C):
        return self._dll.bbd_dispconv_scratch_doubles(C)

    def call(self, name, *args):
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            raise BbdError("%s failed with status %d" % (name, rc))

    @staticmethod
    def stream_for(tensor):
        # raw handle of torch's CURRENT stream on the tensor's device (the C accessor: this is on the
        # path of every launch, torch.cuda.current_stream() builds a Python Stream object each time)
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(tensor.get_device()))


_LIB = None


def get_lib():
    global _LIB
    if _LIB is None:
        _LIB = HipLibrary()
    return _LIB


def ptr(t):
    """Device (or host, for the test port) address of a contiguous tensor; None -> NULL."""
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_contiguous(), "C-ABI tensors must be contiguous"
    return ctypes.c_void_p(t.data_ptr())
