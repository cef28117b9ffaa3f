"""Input contract of the reference's loader (dataset/cityscapes_dataset.py:97-120) -- CPU side: the numpy restatement of Pillow's
8-bit BICUBIC / NEAREST resize (oracle/pil_resize.py) against vectors produced by Pillow itself (tests/golden/g13_pil_resize.npz,
oracle/gen_golden_resize.py), the product's host-side tables against the restatement's, and the dataset classes' list handling."""
import os

import numpy as np
import pytest

from oracle import pil_resize as pr
from simt_amd.data import resample as rs

G = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_matches_pillow_golden():
    d = np.load(os.path.join(G, "g13_pil_resize.npz"))
    for i in range(int(d["n_cases"])):
        cw, ch = (int(v) for v in d[f"crop_{i}"])
        img, lab = pr.cityscapes_pseudo_item(d[f"rgb_{i}"], d[f"lab_{i}"], cw, ch)
        assert np.array_equal(img, d[f"image_{i}"]), i            # float32 BGR - mean, CHW: bit-exact
        assert np.array_equal(lab, d[f"label_{i}"]), i
    for j in range(int(d["n_ramps"])):                            # Cityscapes geometries (2048 / 1024 -> 1024 / 768 / 512 / 1280 / 640)
        src, dst = (int(v) for v in d[f"ramp_{j}"])
        ramp = (np.arange(src) * 37 % 251).astype(np.uint8)
        assert np.array_equal(pr.resize_bicubic_u8(np.repeat(ramp[None, :, None], 3, 2), dst, 1)[0, :, 0], d[f"ramp_bicubic_{j}"])
        assert np.array_equal(ramp[pr.nearest_index_table(src, dst)], d[f"ramp_nearest_{j}"])


@pytest.mark.parametrize("sizes", [(2048, 1024), (2048, 768), (1024, 512), (1024, 768), (50, 21), (31, 90), (96, 36), (1024, 640),
                                   (2048, 1280), (7, 7), (5, 11)])
def test_product_tables_equal_oracle_tables(sizes):
    i, o = sizes
    k1, b1, c1 = pr.precompute_coeffs(i, o)
    k2, b2, c2 = rs.bicubic_tables(i, o)
    assert k1 == k2 and np.array_equal(b1, b2) and np.array_equal(c1, c2)
    assert np.array_equal(pr.nearest_index_table(i, o), rs.nearest_table(i, o))
    assert np.all(b2[:, 0] >= 0) and np.all(b2[:, 0] + b2[:, 1] <= i)          # every tap inside the source
    assert np.all(np.abs(c2.sum(1) - (1 << rs.PRECISION_BITS)) <= k2)          # rows sum to 1.0 in 22-bit fixed point (rounding only)


def test_dataset_lists_and_decode(tmp_path):
    """cityscapesPseudo / cityscapesDataSet: list parsing, max_iters repetition, name derivation (:79-95, :30-41) and decode()."""
    Image = pytest.importorskip("PIL.Image")
    from simt_amd.dataset.cityscapes_dataset import cityscapesDataSet, cityscapesPseudo
    rng = np.random.default_rng(0)
    (tmp_path / "img").mkdir()
    (tmp_path / "lab").mkdir()
    (tmp_path / "val").mkdir()
    lines = []
    for i in range(3):
        rgb = rng.integers(0, 256, (20, 40, 3), dtype=np.uint8)
        lab = rng.integers(0, 19, (20, 40), dtype=np.uint8)
        Image.fromarray(rgb).save(tmp_path / "img" / f"a{i}.png")
        Image.fromarray(lab).save(tmp_path / "lab" / f"a{i}_label.png")
        Image.fromarray(rgb).save(tmp_path / "val" / f"a{i}.png")
        lines.append(f"img/a{i}.png lab/a{i}_label.png")
    (tmp_path / "pseudo.lst").write_text("\n".join(lines) + "\n")
    (tmp_path / "val.txt").write_text("\n".join(f"a{i}.png" for i in range(3)) + "\n")
    ds = cityscapesPseudo(str(tmp_path), str(tmp_path / "pseudo.lst"), max_iters=7, crop_size=(16, 8), mean=pr.IMG_MEAN)
    assert len(ds) == 9 and ds.files[4]["name"] == "a1_label"                 # 3 ids repeated ceil(7/3) = 3 times
    rgb, lab, name = ds.decode(1)
    assert rgb.shape == (20, 40, 3) and rgb.dtype == np.uint8 and lab.shape == (20, 40) and name == "a1_label"
    dv = cityscapesDataSet(str(tmp_path), str(tmp_path / "val.txt"), crop_size=(16, 8), mean=pr.IMG_MEAN, set="val")
    assert len(dv) == 3 and dv.decode(2)[1] is None and dv.decode(2)[2] == "a2.png"


def test_ntm_stats_oracle_matches_reference_golden():
    """oracle.class_distribution / rect_hist / label_mapping (tools/compute_ClassDistribution.py, compute_ConfusionMatrix.py) against
    the outputs of the reference's own functions (tests/golden/g14_hist.npz)."""
    from oracle import simt_oracle as so
    d = np.load(os.path.join(G, "g14_hist.npz"))
    c, nrm = so.class_distribution(list(d["cd_preds"]))
    assert np.array_equal(c, d["cd_counts"]) and np.array_equal(nrm, d["cd_norm"])
    M = np.zeros((34, 19))
    for g_, p in zip(d["cm_gts"], d["cm_preds"]):
        M += so.rect_hist(so.label_mapping(g_, d["cm_mapping"]).flatten(), p.flatten(), 34, 19)
    assert np.array_equal(M, d["cm_counts"])
