"""Build + bind the host port of the kernels' arithmetic (tests/host_port/bbd_host_port.cpp).

Test infrastructure only.  `HostPortBackend` plugs into the `backend=` seam of
baseboostdepth_amd.ops so the CPU tier exercises the product's Python plumbing (plan tables,
projection table, autograd wrappers) together with the exact per-pixel math of the HIP kernels.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_port", "bbd_host_port.cpp")
SRC_IMAGE = os.path.join(HERE, "host_port", "bbd_image_port.cpp")
LIB = os.path.join(HERE, "host_port", "libbbd_host_port.so")
DEPS = [SRC, SRC_IMAGE, os.path.join(HERE, "..", "baseboostdepth_amd", "csrc", "bbd_math.h"),
        os.path.join(HERE, "..", "baseboostdepth_amd", "csrc", "bbd_image_math.h"),
        os.path.join(HERE, "..", "include", "bbd_hip.h")]


def build():
    if os.path.isfile(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS):
        return LIB
    cmd = ["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-std=c++17", "-o", LIB, SRC, SRC_IMAGE]
    subprocess.run(cmd, check=True)
    return LIB


class HostPortBackend:
    name = "host-port"

    def __init__(self):
        self.dll = ctypes.CDLL(build())

    def num_tiles(self, H, W):
        return 1

    def num_tiles_fwd(self, H, W):
        return 1

    def num_tiles_bwd(self, H, W):
        return 1

    def smooth_chunks(self):
        return 1

    def fused_work_items(self, plan, S, H, W, backward, device):
        return None       # the work-item table is a launch-order choice of the GPU kernels

    @staticmethod
    def _check(*tensors):
        for t in tensors:
            assert t is None or not t.is_cuda

    def run(self, name, anchor, *args):
        fn = getattr(self.dll, name.replace("bbd_", "hp_"))
        fn.restype = ctypes.c_int
        conv = []
        for a in args:
            if isinstance(a, float):
                conv.append(ctypes.c_double(a))
            elif isinstance(a, int):
                conv.append(ctypes.c_int(a))
            else:
                conv.append(a)
        rc = fn(*conv)
        assert rc == 0, (name, rc)

    def check_div(self, start, count, stride):
        self.dll.hp_check_div.restype = ctypes.c_int
        return self.dll.hp_check_div(ctypes.c_uint32(start), ctypes.c_uint32(count), ctypes.c_uint32(stride))
