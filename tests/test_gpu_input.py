"""GPU parity of the input pipeline (csrc/input_prep.hip, simt_amd/data/pipeline.py) through the C ABI:
resize BICUBIC / NEAREST + BGR - mean + CHW + int64 labels are BIT-EXACT against Pillow's own output (tests/golden/g13_pil_resize.npz)
and against the numpy restatement at Cityscapes geometry; the prefetcher hands batches out in order; GpuLoader end to end equals the
reference loader's per-item arithmetic (dataset/cityscapes_dataset.py:97-120)."""
import os

import numpy as np
import pytest
import torch

from oracle import pil_resize as pr
from simt_amd.data.pipeline import DevicePrefetcher, GpuLoader, InputPrep

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _run(dev, rgb, lab, cw, ch, mirror=False):
    B = rgb.shape[0]
    prep = InputPrep(B, rgb.shape[1:3], (cw, ch), dev, mean=pr.IMG_MEAN)
    x = torch.full((B, 3, ch, cw), float("nan"), device=dev)
    lo = torch.full((B, ch, cw), -1, dtype=torch.int64, device=dev)
    prep.run(torch.from_numpy(rgb).to(dev), x, torch.from_numpy(lab).to(dev), lo, mirror=mirror)
    torch.cuda.synchronize()
    return x.cpu().numpy(), lo.cpu().numpy()


def test_resize_and_convert_bit_exact_vs_pillow_golden(dev):
    d = np.load(os.path.join(G, "g13_pil_resize.npz"))
    for i in range(int(d["n_cases"])):
        cw, ch = (int(v) for v in d[f"crop_{i}"])
        x, lo = _run(dev, d[f"rgb_{i}"][None], d[f"lab_{i}"][None], cw, ch)
        assert np.array_equal(x[0], d[f"image_{i}"]), f"case {i}: image differs from Pillow"
        assert np.array_equal(lo[0].astype(np.float32), d[f"label_{i}"]), f"case {i}: label differs from Pillow"


@pytest.mark.parametrize("crop", [(1024, 512), (768, 768)])
def test_cityscapes_geometry_batch_and_mirror_quirk(dev, crop):
    """2 frames of 1024 x 2048 -> the reference's default crop and the benchmark's: equal to the restatement (itself pinned on Pillow at
    these geometries by the ramp vectors); per-item mirror flags reproduce the reference's channel-axis flip (:108-111)."""
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (2, 1024, 2048, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:1024, 0:2048]
    rgb[0, :, :, 1] = ((xx // 9 + yy // 5) % 2 * 255).astype(np.uint8)            # hard edges: negative lobes + clipping
    lab = rng.integers(0, 19, (2, 1024, 2048), dtype=np.uint8)
    lab[rng.random(lab.shape) < 0.1] = 255
    cw, ch = crop
    x, lo = _run(dev, rgb, lab, cw, ch, mirror=[False, True])
    for b, flip in ((0, 1), (1, -1)):
        img, lb = pr.cityscapes_pseudo_item(rgb[b], lab[b], cw, ch, mirror_flip=flip)
        assert np.array_equal(x[b], img) and np.array_equal(lo[b].astype(np.float32), lb), (b, flip)


def test_prefetcher_order_and_content(dev):
    """7 distinct host batches (not pinned: staged through the prefetcher's own pinned buffers) come out in order, transformed."""
    B, H, W = 2, 24, 40
    rng = np.random.default_rng(1)
    batches = [(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8), rng.integers(0, 19, (B, H, W), dtype=np.uint8), k) for k in range(7)]
    prep = InputPrep(B, (H, W), (20, 12), dev, mean=pr.IMG_MEAN)
    got = []
    for x, lab, meta in DevicePrefetcher(iter(batches), prep):
        y = (x * 2).sum()                               # consumer work on the current stream
        got.append((x.clone(), lab.clone(), meta, y))
    torch.cuda.synchronize()
    assert [g[2] for g in got] == list(range(7))
    for (rgb, lab, _k), (x, lo, _m, _y) in zip(batches, got):
        for b in range(B):
            img, lb = pr.cityscapes_pseudo_item(rgb[b], lab[b], 20, 12)
            assert np.array_equal(x[b].cpu().numpy(), img) and np.array_equal(lo[b].cpu().numpy().astype(np.float32), lb)


@pytest.mark.parametrize("hold", [2, 3])
def test_prefetcher_gradient_accumulation_hold(dev, hold):
    """--iter-size > 1 (trainV2_simt.py:341-432): the training loop pulls `iter_size` micro-batches BEFORE it enqueues the step that reads
    them.  With hold=iter_size none of them may be refilled before that step has run: the consumer here delays its reads behind ~20 ms
    of GPU work enqueued after the pulls (round 2's prefetcher overwrote the first micro-batch with batch hold+1 in this pattern)."""
    B, H, W = 2, 24, 40
    rng = np.random.default_rng(11)
    n = 4 * hold
    batches = [(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8), rng.integers(0, 19, (B, H, W), dtype=np.uint8), k) for k in range(n)]
    prep = InputPrep(B, (H, W), (20, 12), dev, mean=pr.IMG_MEAN)
    pf = DevicePrefetcher(iter(batches), prep, hold=hold)
    busy = torch.zeros(64 << 20, device=dev)
    got = []
    for _step in range(n // hold):
        mb = [next(pf) for _ in range(hold)]
        for _ in range(40):
            busy.add_(1.0)                              # the step's kernels are enqueued AFTER all pulls and take a while
        for x, lab, meta in mb:
            got.append((x.clone(), lab.clone(), meta))
    torch.cuda.synchronize()
    assert [g[2] for g in got] == list(range(n))
    for (rgb, lab, _k), (x, lo, _m) in zip(batches, got):
        for b in range(B):
            img, lb = pr.cityscapes_pseudo_item(rgb[b], lab[b], 20, 12)
            assert np.array_equal(x[b].cpu().numpy(), img) and np.array_equal(lo[b].cpu().numpy().astype(np.float32), lb)


def test_gpu_loader_end_to_end(dev, tmp_path):
    Image = pytest.importorskip("PIL.Image")
    from simt_amd.dataset.cityscapes_dataset import cityscapesPseudo
    rng = np.random.default_rng(2)
    (tmp_path / "img").mkdir()
    (tmp_path / "lab").mkdir()
    lines, truth = [], {}
    for i in range(5):
        rgb = rng.integers(0, 256, (64, 128, 3), dtype=np.uint8)
        lab = rng.integers(0, 19, (64, 128), dtype=np.uint8)
        Image.fromarray(rgb).save(tmp_path / "img" / f"f{i}.png")
        Image.fromarray(lab).save(tmp_path / "lab" / f"f{i}.png")
        lines.append(f"img/f{i}.png lab/f{i}.png")
        # the reference's per-item arithmetic, with Pillow itself
        im = np.asarray(Image.fromarray(rgb).resize((48, 24), Image.BICUBIC), np.float32)[:, :, ::-1] - np.asarray(pr.IMG_MEAN, np.float32)
        truth[f"f{i}"] = (im.transpose(2, 0, 1), np.asarray(Image.fromarray(lab).resize((48, 24), Image.NEAREST), np.float32))
    (tmp_path / "list.lst").write_text("\n".join(lines) + "\n")
    ds = cityscapesPseudo(str(tmp_path), str(tmp_path / "list.lst"), crop_size=(48, 24), mean=pr.IMG_MEAN)
    seen = []
    for images, labels, sizes, names in GpuLoader(ds, 2, shuffle=True, num_workers=2, device=dev, seed=3, epochs=2):
        assert images.is_cuda and images.shape == (2, 3, 24, 48) and labels.dtype == torch.int64 and tuple(sizes[0]) == (24, 48, 3)
        torch.cuda.synchronize()
        for b, n in enumerate(names):
            assert np.array_equal(images[b].cpu().numpy(), truth[n][0]) and np.array_equal(labels[b].cpu().numpy().astype(np.float32), truth[n][1])
            seen.append(n)
    assert len(seen) == 8 and len(set(seen[:4])) == 4          # 2 epochs x 2 full batches, no repeats inside an epoch
    item = ds[3]                                               # reference-style tuple through the device transform
    assert np.array_equal(item[0], truth["f3"][0]) and np.array_equal(item[1], truth["f3"][1]) and item[3] == "f3"
