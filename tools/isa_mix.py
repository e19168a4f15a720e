#!/usr/bin/env python3
"""Static instruction mix of the hot-path kernels from hipcc's assembly (no GPU needed): how many of a kernel's vector
instructions go to the fp32 pipe (4.1 cycles per wave-instruction per SIMD, profiles/r03/valu_rate.txt) and how many
to the integer / convert pipe that runs beside it (2.1 cycles).  bench.py prices the kernels' issue floor with the
fp32 share printed here.   usage: tools/isa_mix.py [out.json]"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "baseboostdepth_amd", "csrc", "bbd_kernels.hip")
KERNELS = {"bbd_warp_ssim_min_fwd": "warp_ssim_min_fwd_kernel", "bbd_warp_ssim_min_bwd": "warp_ssim_min_bwd2_kernel",
           "bbd_identity_loss_fwd": "identity_loss_grouped_kernel"}
# vector opcodes of the fp32 pipe (measured kinds: fma / mul / add / cmp / cndmask / dpp / max / min / v_mul_lo at ~4.1
# cycles, v_rcp 8.2; packed forms included); everything else (integer add / logic / shifts, 64-bit shift-add,
# conversions, floor) issues at ~2.1 cycles on the second pipe
FP = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32",
      "v_max_f32", "v_min_f32", "v_cndmask_b32", "v_cmp", "v_rcp_f32", "v_pk_", "v_med3_f32", "v_div_", "v_mul_lo_u32",
      "v_mul_hi_u32", "v_mad_u64_u32", "v_permlane", "v_mul_legacy_f32", "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32")


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    asm = "/tmp/bbd_isa_mix.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math",
                    "-fno-slp-vectorize", "-std=c++17", "-S", "--cuda-device-only", "-o", asm, SRC], check=True,
                   stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    res = {}
    for entry, kern in KERNELS.items():
        start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*%s.*:" % kern, l))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        c = collections.Counter()
        for l in lines[start:end]:
            t = l.strip()
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            op = t.split()[0]
            if op.startswith("v_"):
                c["fp" if any(op.startswith(f) for f in FP) else "int"] += 1
                if "dpp" in t and not any(op.startswith(f) for f in FP):
                    c["int"] -= 1
                    c["fp"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "scratch_", "buffer_")):
                c["vmem"] += 1
        valu = c["fp"] + c["int"]
        res[entry] = {"kernel": kern, "valu": valu, "fp32_pipe": c["fp"], "int_pipe": c["int"],
                      "fp32_share": round(c["fp"] / valu, 3), "salu": c["salu"], "lds": c["lds"], "vmem": c["vmem"]}
        print(entry, res[entry])
    res["_note"] = ("static counts over each kernel's whole code (unrolled bodies dominate); fp32 pipe = instructions that "
                    "issue at ~4.1 cycles per wave-instruction per SIMD, int pipe = ~2.1 cycles on a pipe that runs beside it "
                    "(profiles/r03/valu_rate.txt)")
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
