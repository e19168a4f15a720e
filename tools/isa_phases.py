#!/usr/bin/env python3
"""Static instruction mix of the fused kernels PER PHASE (no GPU): compiles bbd_kernels.hip with -DBBD_MARKS, which turns
every BBD_STAMP site into an assembly comment, and sums the vector instructions by issue class (tools/isa_mix.py) between
consecutive marks in layout order.  usage: tools/isa_phases.py [fwd|fwd_many|bwd] [extra hipcc flags...]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_mix import issue_class, CYCLES, DORMANT  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
kern = {"fwd": "warp_ssim_min_fwd_kernelILb0ELi1", "fwd_many": "warp_ssim_min_fwd_kernelILb0ELi0", "fwd_held": "warp_ssim_min_fwd_kernelILb0ELi2",
        "bwd": "warp_ssim_min_bwd9_kernelILb1"}[which]
src = os.environ.get("BBD_VARIANT_SRC", os.path.join(ROOT, "baseboostdepth_amd", "csrc", "bbd_kernels.hip"))
asm = "/tmp/bbd_isa_phases.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math",
                "-fno-slp-vectorize", "-std=c++17", "-DBBD_MARKS", "-I" + os.path.join(ROOT, "baseboostdepth_amd", "csrc"),
                "-S", "--cuda-device-only", "-o", asm, src] + sys.argv[2:], check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*%s.*:" % kern, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
cur, order, tab = "start", ["start"], collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; bbd_mark (\d+)", t)
    if m:
        cur = "after mark %s" % m.group(1)
        if cur not in order:
            order.append(cur)
        continue
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith(DORMANT):
        continue
    if op.startswith("v_"):
        tab[cur][issue_class(op, t[len(op):])] += 1
    elif op.startswith("s_"):
        tab[cur]["salu"] += 1
    elif op.startswith("ds_"):
        tab[cur]["lds"] += 1
    elif op.startswith(("global_", "scratch_", "buffer_")):
        tab[cur]["vmem"] += 1
tot = collections.Counter()
print("%-16s %6s %6s %6s %8s %6s %5s %5s" % (kern[:16], "A", "B", "C", "cycles", "salu", "lds", "vmem"))
for k in order:
    c = tab[k]
    cyc = sum(c[x] * CYCLES[x] for x in "ABC")
    print("%-16s %6d %6d %6d %8.0f %6d %5d %5d" % (k, c["A"], c["B"], c["C"], cyc, c["salu"], c["lds"], c["vmem"]))
    tot.update(c)
print("%-16s %6d %6d %6d %8.0f %6d %5d %5d" % ("total", tot["A"], tot["B"], tot["C"], sum(tot[x] * CYCLES[x] for x in "ABC"),
                                              tot["salu"], tot["lds"], tot["vmem"]))
