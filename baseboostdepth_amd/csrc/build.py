"""In-tree build of the C-ABI library: hipcc for gfx950, nothing else.

    python -m baseboostdepth_amd.csrc.build

The .so is git-ignored (history stays source-only) but travels to the GPU box with the tree.
-ffp-contract=off is REQUIRED: bbd_math.h mirrors the reference CPU path's rounding order and
places every FMA explicitly.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "bbd_kernels.hip"), os.path.join(HERE, "bbd_eval.hip"),
        os.path.join(HERE, "bbd_image.hip"), os.path.join(HERE, "bbd_nn.hip"), os.path.join(HERE, "bbd_vit.hip")]
OUT = os.path.join(HERE, "libbbd_hip.so")
DEPS = SRCS + [os.path.join(HERE, "bbd_math.h"), os.path.join(HERE, "bbd_image_math.h"), os.path.join(HERE, "..", "..", "include", "bbd_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-std=c++17"]


def build(force=False, verbose=False):
    if not force and os.path.isfile(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT] + SRCS
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
