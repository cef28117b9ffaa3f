"""Device-side label statistics shared by the two offline NTM utilities of the reference:
tools/compute_ClassDistribution.py (the 19-vector class prior that sig_NTM multiplies into T, model/deeplab_multi.py:255-261) and
tools/compute_ConfusionMatrix.py (34 x 19 ground-truth-id x pseudo-label confusion counts).  The reference loops over the PNGs with
numpy bincount on one core; here the decoded uint8 images are uploaded through pinned memory and counted by simt_hist2d_u8 (integer
atomics: exact, order independent), with PNG decoding on a few host threads.  PyTorch supplies device memory and streams only."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import _lib as L


def mapping_lut(mapping):
    """label_mapping (evaluate_cityscapes.py:90-94) as a 256-entry table: rows (src, dst) applied in order on the ORIGINAL values."""
    lut = np.arange(256, dtype=np.int64)
    for src, dst in np.asarray(mapping).reshape(-1, 2):
        if 0 <= src < 256:
            lut[int(src)] = int(dst)
    return np.clip(lut, 0, 255).astype(np.uint8)


class LabelHistogram:
    """hist[row(a)][b] accumulated on the device over any number of image pairs; rows = 1 gives the class histogram."""

    def __init__(self, n_rows, n_cols, device="cuda:0", mapping=None):
        self.na, self.nb, self.dev = n_rows, n_cols, torch.device(device)
        if not torch.cuda.is_available():
            raise RuntimeError("LabelHistogram runs on the GPU (libsimt_hip.so); there is no CPU fallback")
        self.hist = torch.zeros(n_rows, n_cols, dtype=torch.int64, device=self.dev)
        self.lut = torch.from_numpy(mapping_lut(mapping)).to(self.dev) if mapping is not None else None
        self.stream = torch.cuda.current_stream(self.dev)

    def add(self, b, a=None):
        """b: uint8 array of column classes (pseudo labels); a: uint8 array of row ids of the same size, or None when n_rows == 1."""
        b = np.ascontiguousarray(b, dtype=np.uint8)
        bd = torch.from_numpy(b.reshape(-1)).to(self.dev, non_blocking=True)
        ad = None
        if a is not None:
            a = np.ascontiguousarray(a, dtype=np.uint8)
            if a.size != b.size:
                return False                   # the reference prints "Skipping" and continues (compute_ConfusionMatrix.py:92-95)
            ad = torch.from_numpy(a.reshape(-1)).to(self.dev, non_blocking=True)
        L.call("simt_hist2d_u8", ad.data_ptr() if ad is not None else None, bd.data_ptr(), bd.numel(), self.na, self.nb,
               self.lut.data_ptr() if self.lut is not None else None, self.hist.data_ptr(), self.stream.cuda_stream)
        return True

    def result(self):
        return self.hist.cpu().numpy()


def decode_many(paths, workers=8):
    """PNG -> uint8 arrays on host threads, in order (np.array(Image.open(p)) like the reference, :83 / :88-89)."""
    from PIL import Image

    def one(p):
        return np.array(Image.open(p))
    with ThreadPoolExecutor(max(1, workers)) as pool:
        yield from pool.map(one, paths)
