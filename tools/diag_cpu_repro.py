"""Does the oracle (PyTorch CPU ops) on THIS machine reproduce the golden vectors bit for bit?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import torch
from golden_io import Case
from oracle import hotpath_ref as O
print(torch.__config__.show().split("\n")[0:8]); print("threads", torch.get_num_threads())
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Flags' | cut -c1-200")
for nt in (None, 1):
    if nt: torch.set_num_threads(nt)
    for name in ["md2_b2_32x64", "tri_3105_32x64", "md2_b1_192x640"]:
        c = Case(name)
        out = O.hot_path(c.inputs, c.disp, c.poses, c.ms, c.scales, c.trimin, c.decomp, c.noise, c.H, c.W,
                         poses_error=c.poses_error(), keep=True)
        msg = []
        for s in c.scales:
            d = (out["min/%d" % s] - c.expected("out/min/%d" % s)).abs()
            msg.append("s%d exact=%.4f max=%.2e" % (s, float((d == 0).float().mean()), float(d.max())))
            if c.has("out/depth/%d" % s):
                dd = (out[("depth", 0, s)].detach() - c.expected("out/depth/%d" % s)).abs()
                msg.append("depth exact=%.4f" % float((dd == 0).float().mean()))
        for k in c.z.files:
            if k.startswith("out/color/1/0"):
                w = out[("color", 1, 0)].detach(); dd = (w - c.expected(k)).abs()
                msg.append("warp(1,0) exact=%.4f max=%.2e" % (float((dd == 0).float().mean()), float(dd.max())))
        print("threads=%s %s: %s" % (torch.get_num_threads(), name, " | ".join(msg)))
