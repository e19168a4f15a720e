#!/usr/bin/env python3
"""Static instruction mix of the hot-path kernels from hipcc's assembly (no GPU needed): the vector instructions by issue
class (see CYCLES below).  bench.py prices the kernels' issue bound with the mean cycles per instruction printed here.
usage: tools/isa_mix.py [out.json]
       tools/isa_mix.py --check committed.json    exit 1 when the committed file was not generated from the shipped source
                                                  (source hash) or its counts differ from a fresh compile"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "baseboostdepth_amd", "csrc", "bbd_kernels.hip")
# (the training path's instantiations: forward fed with disparities = PLANE false, backward with the forward's depth = true)
#  forward: three forms by launch shape (fused_fwd_form)
KERNELS = {"bbd_warp_ssim_min_fwd": "warp_ssim_min_fwd_kernelILb0ELi1", "bbd_warp_ssim_min_fwd_many": "warp_ssim_min_fwd_kernelILb0ELi0",
           "bbd_warp_ssim_min_fwd_held": "warp_ssim_min_fwd_kernelILb0ELi2",
           "bbd_warp_ssim_min_bwd": "warp_ssim_min_bwd9_kernelILb1",
           "bbd_identity_loss_fwd": "identity_loss_grouped_kernel"}
# Issue cost classes measured by tools/microbench/valu_rate.hip (profiles/r03/valu_rate.txt), cycles a wave64 instruction
# occupies a SIMD when at least two waves share it:
#   A  2.2  fp32 / integer / convert instructions whose operands are VGPRs, inline constants or literals
#   B  4.1  the same with an SGPR (or vcc / exec) operand; v_cmp / v_cndmask; every packed v_pk_* (two results);
#           v_max / v_min, v_mul_lo / v_mul_hi / v_mad_u64, DPP forms, lane reads / writes
#   C  8.2  v_rcp / v_rsq / v_sqrt / v_exp / v_log
B_OPS = ("v_pk_", "v_max_f32", "v_min_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_readlane", "v_writelane",
         "v_readfirstlane", "v_cmp", "v_cndmask", "v_permlane", "v_med3")
C_OPS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_")
DORMANT = ("v_div_",)       # the compiler's full IEEE division sequence: only on the guarded fall-back paths, not counted
SCALAR_OPERAND = re.compile(r"(?<![a-z_0-9])s\d+|s\[\d+:\d+\]|vcc|exec")
CYCLES = {"A": 2.2, "B": 4.1, "C": 8.2}


def issue_class(op, operands):
    if any(op.startswith(x) for x in C_OPS):
        return "C"
    if any(op.startswith(x) for x in B_OPS) or "dpp" in operands or SCALAR_OPERAND.search(operands):
        return "B"
    return "A"


def main():
    check = None
    argv = sys.argv[1:]
    if argv and argv[0] == "--check":
        check, argv = argv[1], argv[2:]
    out = argv[0] if argv else None
    sys.path.insert(0, ROOT)
    from baseboostdepth_amd.csrc.build import source_sha16
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
            if op.startswith(DORMANT):
                continue
            if op.startswith("v_"):
                c[issue_class(op, t[len(op):])] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "scratch_", "buffer_")):
                c["vmem"] += 1
        valu = c["A"] + c["B"] + c["C"]
        cyc = sum(c[k] * CYCLES[k] for k in "ABC") / valu
        res[entry] = {"kernel": kern, "valu": valu, "class_A": c["A"], "class_B": c["B"], "class_C": c["C"],
                      "cycles_per_valu_instruction": round(cyc, 3), "salu": c["salu"], "lds": c["lds"], "vmem": c["vmem"]}
        print(entry, res[entry])
    res["_note"] = ("static counts over each kernel's whole code (unrolled bodies dominate), vector instructions by issue class: "
                    "A = 2.2 cycles (register / constant operands), B = 4.1 (an SGPR / vcc operand, packed, compare / select, ...), "
                    "C = 8.2 (reciprocal etc.) per wave-instruction per SIMD with >= 2 waves resident (profiles/r03/valu_rate.txt); "
                    "cycles_per_valu_instruction = their weighted mean, what bench.py prices the launch's counter-measured "
                    "instruction count with")
    res["kernel_source_sha16"] = source_sha16()
    if out:
        json.dump(res, open(out, "w"), indent=1)
    if check:
        old = json.load(open(check))
        bad = []
        if old.get("kernel_source_sha16") != res["kernel_source_sha16"]:
            bad.append("source hash %s != shipped %s" % (old.get("kernel_source_sha16"), res["kernel_source_sha16"]))
        for k in KERNELS:
            for f in ("valu", "class_A", "class_B", "class_C"):
                if old.get(k, {}).get(f) != res[k][f]:
                    bad.append("%s.%s: committed %s, fresh %s" % (k, f, old.get(k, {}).get(f), res[k][f]))
        print("isa_mix --check:", "STALE: " + "; ".join(bad) if bad else "up to date")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
