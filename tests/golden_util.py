"""Helpers shared by the parity tests: load a golden case written by
oracle/make_golden.py and turn it into the dicts the trainer surface uses."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    cfg = json.loads(str(z["config"]))
    return cfg, z


def fid_key(f):
    return "s" if f == "s" else int(f)


def build_dicts(cfg, z, image_dtype=torch.float64, device="cpu", requires_grad=True):
    """inputs / outputs dicts keyed as Trainer.py does, plus the leaf tensors."""
    fids = [fid_key(f) for f in cfg["frame_ids"]]
    inputs = {"K": torch.from_numpy(z["K"]).to(device), "inv_K": torch.from_numpy(z["inv_K"]).to(device)}
    for f in fids:
        inputs[("image", f)] = torch.from_numpy(z["image_%s" % f]).to(image_dtype).to(device)
    disp = torch.from_numpy(z["disp"]).to(device).requires_grad_(requires_grad)
    outputs = {("disp", 0): disp}
    poses = {}
    for f in fids[1:]:
        T = torch.from_numpy(z["T_%s" % f]).to(device).requires_grad_(requires_grad)
        poses[f] = T
        if f == "s":
            inputs["stereo_T"] = T
        else:
            outputs[("cam_T_cam", f)] = T
    noise = torch.from_numpy(z["noise"]).to(device)
    return inputs, outputs, disp, poses, noise, fids
