"""Load tests/golden/*.npz fixtures (made by tools/make_golden.py) back into torch form."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DIRECT_CASES = ["md2_b2_32x64", "md2_mixed_b3_32x64", "md2_b1_192x640", "tri_3105_32x64",
                "tri_7765_32x64", "tri_2102_32x64", "tri_nodecomp_3210_32x64", "tri_4444_16x32",
                "tri_6123_16x32", "tri_0000_16x32", "tri_1357_16x32", "tri_7_b1_192x640", "tri_2_b1_192x640"]
POSE_CASES = ["pose_plain_3105_32x64", "pose_incr_3215_32x64", "pose_incr_partial_4327_32x64",
              "pose_md2_b2_32x64"]


def _frame(tok):
    return "s" if tok == "s" else int(tok)


class Case:
    """One fixture: `inputs` mirrors the reference's post-collate batch dict."""

    def __init__(self, name, device="cpu"):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name, self.z = name, z
        self.ms = [int(v) for v in z["meta/m"]]
        self.scales = [int(v) for v in z["meta/scales"]]
        self.trimin, self.decomp, self.incremental, self.partial = (bool(v) for v in z["meta/flags"])
        self.cutt = float(z["meta/cutt"])
        self.to_use = int(z["meta/to_use"])
        self.frames = ["s" if int(v) == -50 else int(v) for v in z["meta/frames"]]
        self.inputs = {}
        for k in z.files:
            parts = k.split("/")
            if parts[0] == "in" and parts[1] in ("color", "color_aug"):
                img = torch.from_numpy(z[k]).float().div(255)
                self.inputs[(parts[1], _frame(parts[2]), int(parts[3]))] = img.to(device)
        for key in [k for k in self.inputs if k[0] == "color" and k[1] != "s" and k[2] == 0]:
            # lean fixtures drop color_aug (== color for every fixture: tools/make_golden.py builds it as a clone)
            self.inputs.setdefault(("color_aug", key[1], 0), self.inputs[key])
        self.inputs[("K", 0)] = torch.from_numpy(z["in/K"]).to(device)
        self.inputs[("inv_K", 0)] = torch.from_numpy(z["in/inv_K"]).to(device)
        self.inputs["stereo_T"] = torch.from_numpy(z["in/stereo_T"]).to(device)
        self.inputs["frames"] = list(self.frames)
        self.inputs["ordering"] = [[0, "s"] if m == 0 else [0, m, -m] for m in self.ms]
        self.inputs["cutt"] = torch.tensor(self.cutt)
        self.inputs["to_use"] = torch.tensor(self.to_use)
        self.B = len(self.ms)
        self.H, self.W = self.inputs[("color", 0, 0)].shape[-2:]
        self.disp = {s: torch.from_numpy(z["disp/%d" % s]).to(device).requires_grad_(True) for s in self.scales}
        self.poses = {}
        for k in z.files:
            if k.startswith("T/"):
                self.poses[_frame(k[2:])] = torch.from_numpy(z[k]).to(device).requires_grad_(True)
        self.noise = torch.from_numpy(z["noise"]).to(device)

    def poses_error(self, pose_error=5.5):
        out = {}
        for f, T in self.poses.items():
            Te = T.clone().detach().cpu()             # CPU division, as the reference does it
            Te[:, :3, 3:] /= pose_error
            out[f] = Te.to(T.device)
        return out

    def expected(self, key):
        return torch.from_numpy(np.asarray(self.z[key]))

    def has(self, key):
        return key in self.z.files
