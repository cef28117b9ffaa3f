#!/usr/bin/env python3
"""tools/compute_ClassDistribution.py of the reference on MI355X: histogram of the black-box model's pseudo labels over the training
set -> normalised 19-vector -> ClassDist/ClassDist_<name>.npy, the class prior `sig_NTM` loads (model/deeplab_multi.py:255).

    python -m simt_amd.tools.compute_ClassDistribution --pred-dir .../pseudo_bapa --devkit-dir dataset/cityscapes_list --out ClassDist_bapa.npy

Same function name and return value as the reference's `compute_CD(gt_dir, pred_dir, devkit_dir)` (:66-86: counts as a float64 [19]
vector); the reference's hard-coded paths (:89-90) become flags.  Counting runs on the GPU (tools/ntm_stats.py, simt_hist2d_u8)."""
import argparse
import json
from os.path import join

import numpy as np

from .ntm_stats import LabelHistogram, decode_many


def compute_CD(gt_dir, pred_dir, devkit_dir="../dataset/cityscapes_list", device="cuda:0", workers=8):
    """gt_dir is unused, like in the reference.  Reads <devkit_dir>/info.json ('classes') and <devkit_dir>/train.txt."""
    with open(join(devkit_dir, "info.json"), "r") as fp:
        info = json.load(fp)
    num_classes = int(info["classes"])
    print("Num classes", num_classes)
    pred_imgs = [join(pred_dir, x.split("/")[-1]) for x in open(join(devkit_dir, "train.txt"), "r").read().splitlines()]
    h = LabelHistogram(1, 19, device=device)                      # the reference hard-codes 19 here (:81-84)
    for pred in decode_many(pred_imgs, workers):
        h.add(pred)
    return h.result()[0].astype(np.float64)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gt-dir", default="")
    ap.add_argument("--pred-dir", required=True)
    ap.add_argument("--devkit-dir", default="../dataset/cityscapes_list")
    ap.add_argument("--out", default="../ClassDist/ClassDist_bapa.npy")
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    Class_dist = compute_CD(a.gt_dir, a.pred_dir, a.devkit_dir, device=a.device)
    Class_dist_norm = Class_dist / (np.sum(Class_dist) + 10e-10)                       # :91
    np.save(a.out, Class_dist_norm)
    print(Class_dist, Class_dist_norm)


if __name__ == "__main__":
    main()
