#!/usr/bin/env python3
"""tools/compute_ConfusionMatrix.py of the reference on MI355X: 34 x 19 confusion counts between the Cityscapes ground-truth ids
(through `label_mapping` with info.json's 'label2train_1' table) and a black-box model's pseudo labels, over the training set
(:68-98).  Same function name / return value as `compute_CM(gt_dir, pred_dir, devkit_dir)`; the plotting of :100-127 is left to the
caller.  The reference's info.json does not ship 'label2train_1' (the file is an offline analysis script that imports modules absent
from the tree); pass the table with --mapping-key or as `mapping=`; the default falls back to the identity over ids 0..33.

    python -m simt_amd.tools.compute_ConfusionMatrix --gt-dir .../train_label --pred-dir .../pseudo_bapa --devkit-dir dataset/cityscapes_list
"""
import argparse
import json
from os.path import join

import numpy as np

from .ntm_stats import LabelHistogram, decode_many


def compute_CM(gt_dir, pred_dir, devkit_dir="../dataset/cityscapes_list", mapping=None, mapping_key="label2train_1", device="cuda:0",
               workers=8, n_rows=34, n_cols=19):
    with open(join(devkit_dir, "info.json"), "r") as fp:
        info = json.load(fp)
    print("Num classes", int(info["classes"]))
    if mapping is None:
        mapping = np.array(info[mapping_key]) if mapping_key in info else np.array([[i, i] for i in range(n_rows)])
    gt_imgs = [join(gt_dir, x) for x in open(join(devkit_dir, "train_label.txt"), "r").read().splitlines()]
    pred_imgs = [join(pred_dir, x.split("/")[-1]) for x in open(join(devkit_dir, "train.txt"), "r").read().splitlines()]
    h = LabelHistogram(n_rows, n_cols, device=device, mapping=mapping)
    for gi, (label, pred) in enumerate(zip(decode_many(gt_imgs, workers), decode_many(pred_imgs, workers))):
        if not h.add(pred, label):
            print("Skipping: len(gt) = {:d}, len(pred) = {:d}, {:s}, {:s}".format(label.size, pred.size, gt_imgs[gi], pred_imgs[gi]))
    return h.result().astype(np.float64)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gt-dir", required=True)
    ap.add_argument("--pred-dir", required=True)
    ap.add_argument("--devkit-dir", default="../dataset/cityscapes_list")
    ap.add_argument("--mapping-key", default="label2train_1")
    ap.add_argument("--out", default="ConfusionMatrix.npy")
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    CM = compute_CM(a.gt_dir, a.pred_dir, a.devkit_dir, mapping_key=a.mapping_key, device=a.device)
    np.save(a.out, CM)
    print(CM)


if __name__ == "__main__":
    main()
