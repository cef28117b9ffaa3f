"""Offline NTM utilities on the GPU against the reference's own `fast_hist` / `label_mapping` outputs
(tests/golden/g14_hist.npz, oracle/gen_golden_hist.py): integer counts, exact."""
import json
import os

import numpy as np
import pytest

from oracle import simt_oracle as so
from simt_amd.tools.ntm_stats import LabelHistogram, mapping_lut

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_class_distribution_and_confusion_counts_exact(dev):
    d = np.load(os.path.join(G, "g14_hist.npz"))
    h = LabelHistogram(1, 19, device=dev)
    for p in d["cd_preds"]:
        h.add(p)
    counts = h.result()[0].astype(np.float64)
    assert np.array_equal(counts, d["cd_counts"])
    assert np.array_equal(counts / (np.sum(counts) + 10e-10), d["cd_norm"])          # what ClassDist_*.npy hold
    h2 = LabelHistogram(34, 19, device=dev, mapping=d["cm_mapping"])
    for g_, p in zip(d["cm_gts"], d["cm_preds"]):
        assert h2.add(p, g_)
    assert np.array_equal(h2.result().astype(np.float64), d["cm_counts"])
    assert not h2.add(d["cm_preds"][0], d["cm_gts"][0][:5])                           # size mismatch: skipped like the reference


def test_large_image_and_tools_end_to_end(dev, tmp_path):
    """A Cityscapes-sized pair (1024 x 2048) against the numpy restatement, and both tools through their file interface."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    gt = rng.integers(0, 34, (1024, 2048), dtype=np.uint8)
    gt[rng.random(gt.shape) < 0.05] = 255
    pr = rng.integers(0, 19, (1024, 2048), dtype=np.uint8)
    mapping = np.array([[i, i] for i in range(34)])
    h = LabelHistogram(34, 19, device=dev, mapping=mapping)
    h.add(pr, gt)
    assert np.array_equal(h.result(), so.rect_hist(so.label_mapping(gt, mapping).flatten(), pr.flatten(), 34, 19))
    assert np.array_equal(mapping_lut([[3, 9], [9, 4]])[[3, 9, 7]], np.array([9, 4, 7], np.uint8))   # applied on the ORIGINAL values
    # file interface
    from simt_amd.tools.compute_ClassDistribution import compute_CD
    from simt_amd.tools.compute_ConfusionMatrix import compute_CM
    (tmp_path / "pred").mkdir()
    (tmp_path / "gt").mkdir()
    (tmp_path / "kit").mkdir()
    names, preds, gts = [], [], []
    for i in range(3):
        p = rng.integers(0, 19, (32, 48), dtype=np.uint8)
        p[rng.random(p.shape) < 0.1] = 255
        g_ = rng.integers(0, 34, (32, 48), dtype=np.uint8)
        Image.fromarray(p).save(tmp_path / "pred" / f"im{i}.png")
        Image.fromarray(g_).save(tmp_path / "gt" / f"im{i}_gt.png")
        names.append(f"city/im{i}.png")
        preds.append(p)
        gts.append(g_)
    json.dump({"classes": 19, "label2train_1": [[i, (i + 1) % 34] for i in range(34)]}, open(tmp_path / "kit" / "info.json", "w"))
    (tmp_path / "kit" / "train.txt").write_text("\n".join(names) + "\n")
    (tmp_path / "kit" / "train_label.txt").write_text("\n".join(f"im{i}_gt.png" for i in range(3)) + "\n")
    cd = compute_CD("", str(tmp_path / "pred"), str(tmp_path / "kit"), device=dev, workers=2)
    assert np.array_equal(cd, so.class_distribution(preds)[0])
    cm = compute_CM(str(tmp_path / "gt"), str(tmp_path / "pred"), str(tmp_path / "kit"), device=dev, workers=2)
    m = np.array([[i, (i + 1) % 34] for i in range(34)])
    ref = sum(_masked(g_, p, m) for g_, p in zip(gts, preds))
    assert np.array_equal(cm, ref)


def _masked(g_, p, m):
    """The reference's bincount breaks on ignore pixels in the pseudo label (index 19*a + 255 leaves the row); the device kernel skips
    them.  Reference-equivalent count on the valid pixels only."""
    k = p < 19
    return so.rect_hist(so.label_mapping(g_, m)[k], p[k], 34, 19)
