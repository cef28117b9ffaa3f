#!/usr/bin/env python3
"""Golden vectors for the input contract (dataset/cityscapes_dataset.py:97-120): small random images pushed through PILLOW itself
(the reference's dependency; run in the build container, Pillow version recorded) exactly like cityscapesPseudo.__getitem__ does
-> tests/golden/g13_pil_resize.npz.  Only inputs + outputs are committed."""
import os
import sys

import numpy as np
import PIL
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
IMG_MEAN = np.array((104.00698793, 116.66876762, 122.67891434), dtype=np.float32)      # tools/trainV2_simt.py:34


def reference_item(rgb, lab, crop):
    image = Image.fromarray(rgb).convert("RGB")
    label = Image.fromarray(lab)
    image = image.resize(crop, Image.BICUBIC)
    label = label.resize(crop, Image.NEAREST)
    image = np.asarray(image, np.float32)
    label = np.asarray(label, np.float32)
    image = image[:, :, ::-1]
    image = image - IMG_MEAN
    return image.transpose((2, 0, 1)).copy(), label.copy()


def main():
    rng = np.random.default_rng(13)
    out = {"pillow_version": np.array(PIL.__version__)}
    cases = [(64, 128, 48, 24), (64, 128, 48, 48), (50, 37, 21, 33), (40, 60, 60, 40), (31, 45, 90, 70), (128, 256, 96, 96),
             (96, 96, 36, 36)]
    for i, (h, w, cw, ch) in enumerate(cases):        # source H, W -> crop (w, h) like --input-size-target "w,h"
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        # piecewise-smooth content too (edges + gradients exercise negative lobes / clipping)
        yy, xx = np.mgrid[0:h, 0:w]
        rgb[:, : w // 2, 0] = (xx[:, : w // 2] * 5) % 256
        rgb[h // 3:, :, 1] = np.where((xx[h // 3:] // 7) % 2 == 0, 255, 0)
        lab = rng.integers(0, 19, (h, w), dtype=np.uint8)
        lab[rng.random((h, w)) < 0.1] = 255
        img_f, lab_f = reference_item(rgb, lab, (cw, ch))
        out[f"rgb_{i}"], out[f"lab_{i}"], out[f"crop_{i}"] = rgb, lab, np.array([cw, ch])
        out[f"image_{i}"], out[f"label_{i}"] = img_f, lab_f
    # Cityscapes geometry: 2048 x 1024 source -> 1024x512 (reference default), 768x768 (bench shape), 1280x640 (eval scale 2):
    # only the index / coefficient tables are pinned (a full image would be megabytes): resize a 1-row and a 1-column ramp
    for j, (src, dst) in enumerate(((2048, 1024), (2048, 768), (1024, 512), (1024, 768), (2048, 1280), (1024, 640))):
        ramp = (np.arange(src) * 37 % 251).astype(np.uint8)
        row = np.repeat(ramp[None, :, None], 3, 2)                                     # [1, src, 3]
        out[f"ramp_bicubic_{j}"] = np.asarray(Image.fromarray(np.repeat(row, 2, 0)).resize((dst, 2), Image.BICUBIC))[0, :, 0]
        out[f"ramp_nearest_{j}"] = np.asarray(Image.fromarray(np.repeat(ramp[None, :], 2, 0)).resize((dst, 2), Image.NEAREST))[0]
        out[f"ramp_{j}"] = np.array([src, dst])
    out["n_cases"], out["n_ramps"] = np.array(len(cases)), np.array(6)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g13_pil_resize.npz"), **out)
    print("wrote g13_pil_resize.npz with Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
