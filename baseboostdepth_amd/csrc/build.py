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
        os.path.join(HERE, "bbd_image.hip"), os.path.join(HERE, "bbd_nn.hip"), os.path.join(HERE, "bbd_vit.hip"), os.path.join(HERE, "bbd_pose.hip"), os.path.join(HERE, "bbd_tokens.hip"), os.path.join(HERE, "bbd_util.hip")]
OUT = os.path.join(HERE, "libbbd_hip.so")
DEPS = SRCS + [os.path.abspath(__file__), os.path.join(HERE, "bbd_math.h"), os.path.join(HERE, "bbd_image_math.h"), os.path.join(HERE, "..", "..", "include", "bbd_hip.h")]
# -fno-slp-vectorize: hipcc otherwise SLP-packs neighbouring scalar fp32 adds / multiplies into v_pk_add/mul_f32 and
# pays for it in v_mov register shuffles (129 moves in the forward's SSIM phase): measured forward 0.222 -> 0.208 ms,
# identity 0.0365 -> 0.0340 ms, backward 0.348 -> 0.343 ms (profiles/r02/slp_variants.txt).  Same operations, same bits.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-std=c++17"]


def source_sha16():
    """First 16 hex digits of sha256 over the fused kernels' sources (bbd_kernels.hip + bbd_math.h): stamped into the
    committed counter / instruction-mix files (tools/pmc_summary.py, tools/isa_mix.py) so that bench.py can say when the
    constants it quotes from them were taken from other code than the one it is timing."""
    import hashlib
    h = hashlib.sha256()
    for name in ("bbd_kernels.hip", "bbd_math.h"):
        with open(os.path.join(HERE, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


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
