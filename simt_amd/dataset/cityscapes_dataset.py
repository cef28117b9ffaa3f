"""`dataset.cityscapes_dataset` of the reference on MI355X: same class names and constructor arguments
(dataset/cityscapes_dataset.py:21-63 `cityscapesDataSet`, :66-120 `cityscapesPseudo`), different division of labour:

  * `decode(index)` -- host side: file IO + PNG/JPEG decoding to uint8 (Pillow; out of the hot path's scope);
  * resize (BICUBIC image / NEAREST label, byte-identical to Pillow), BGR - mean, CHW, int64 labels -- device side
    (simt_amd.data.pipeline.InputPrep -> csrc/input_prep.hip);
  * `simt_amd.data.pipeline.GpuLoader(dataset, ...)` replaces torch's DataLoader and yields batches already resident in HBM.

`__getitem__` keeps the reference's return tuple for spot checks (it runs the device transform on a batch of one).
"""
import os.path as osp

import numpy as np
import torch


class _Base:
    def __init__(self, root, list_path, max_iters, crop_size, mean, scale, mirror, ignore_label):
        self.root, self.list_path, self.crop_size, self.scale = root, list_path, tuple(crop_size), scale
        self.ignore_label, self.mean, self.is_mirror = ignore_label, tuple(float(m) for m in mean), mirror
        self.files = []

    def _repeat(self, ids, max_iters):
        if max_iters is not None:
            ids = ids * int(np.ceil(float(max_iters) / len(ids)))      # the reference's way of making the loader "infinite"
        return ids

    def __len__(self):
        return len(self.files)

    @staticmethod
    def _open_rgb(path):
        from PIL import Image
        return np.asarray(Image.open(path).convert("RGB"), np.uint8)

    def _device_item(self, rgb, lab):
        from ..data.pipeline import InputPrep
        if not torch.cuda.is_available():
            raise RuntimeError("the input transform runs on the GPU (no CPU fallback); use decode() for host-side access")
        dev = torch.device("cuda", torch.cuda.current_device())
        prep = InputPrep(1, rgb.shape[:2], self.crop_size, dev, mean=self.mean, with_label=lab is not None)
        x = torch.empty(1, 3, self.crop_size[1], self.crop_size[0], device=dev)
        lo = torch.empty(1, self.crop_size[1], self.crop_size[0], dtype=torch.int64, device=dev) if lab is not None else None
        flip = bool(np.random.choice(2) == 0) if (self.is_mirror and lab is not None) else False
        prep.run(torch.from_numpy(rgb[None]).to(dev), x, torch.from_numpy(lab[None]).to(dev) if lab is not None else None, lo, mirror=flip)
        return x[0].cpu().numpy(), (lo[0].cpu().numpy().astype(np.float32) if lab is not None else None)


class cityscapesDataSet(_Base):
    """Images only (evaluation / target-domain images): dataset/cityscapes_dataset.py:21-63."""

    def __init__(self, root, list_path, max_iters=None, crop_size=(321, 321), mean=(128, 128, 128), scale=True, mirror=True,
                 ignore_label=255, set="val"):
        super().__init__(root, list_path, max_iters, crop_size, mean, scale, mirror, ignore_label)
        self.set = set
        self.img_ids = self._repeat([i_id.strip() for i_id in open(list_path)], max_iters)
        for name in self.img_ids:
            self.files.append({"img": osp.join(self.root, "%s/%s" % (self.set, name)), "name": name})

    def decode(self, index):
        f = self.files[index]
        return self._open_rgb(f["img"]), None, f["name"]

    def __getitem__(self, index):
        rgb, _, name = self.decode(index)
        image, _ = self._device_item(rgb, None)
        return image, np.array((self.crop_size[1], self.crop_size[0], 3)), name


class cityscapesPseudo(_Base):
    """Image + pseudo-label PNG (the SimT stage's training set): dataset/cityscapes_dataset.py:66-120."""

    def __init__(self, root, list_path, max_iters=None, crop_size=(321, 321), mean=(128, 128, 128), scale=True, mirror=False,
                 ignore_label=255):
        super().__init__(root, list_path, max_iters, crop_size, mean, scale, mirror, ignore_label)
        self.img_ids = self._repeat([i_id.strip().split() for i_id in open(list_path)], max_iters)
        for image_path, label_path in self.img_ids:
            self.files.append({"img": osp.join(self.root, image_path), "label": osp.join(self.root, label_path),
                               "name": osp.splitext(osp.basename(label_path))[0]})

    def decode(self, index):
        from PIL import Image
        f = self.files[index]
        lab = np.asarray(Image.open(f["label"]))
        if lab.ndim != 2 or lab.dtype != np.uint8:
            raise ValueError(f"{f['label']}: expected an 8-bit single-channel label image, got {lab.dtype} {lab.shape}")
        return self._open_rgb(f["img"]), lab, f["name"]

    def __getitem__(self, index):
        rgb, lab, name = self.decode(index)
        image, label = self._device_item(rgb, lab)
        return image, label, np.array((self.crop_size[1], self.crop_size[0], 3)), name
