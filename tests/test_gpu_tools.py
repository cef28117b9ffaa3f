"""The outer loop of the reference's SimT stage (tools/trainV2_simt.py:232-464) as `simt_amd.tools.trainV2_simt.main` runs it on the
GPU: real files through the device input pipeline (cityscapesPseudo -> GpuLoader), the periodic evaluate_simt on a validation list
with the best-mIoU snapshot rotation (:452-464), the final `GTA5_<num_steps_stop>.pth` (:447-450) with the reference's 656 state-dict
keys, and the loud failures (missing data directory / checkpoint) that replace the reference's crashes."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_dataset(tmp_path, Image):
    rng = np.random.default_rng(0)
    for d in ("train_img", "train_lab", "val/frankfurt", "gt/frankfurt", "kit", "snap"):
        (tmp_path / d).mkdir(parents=True)
    lines = []
    for i in range(4):
        Image.fromarray(rng.integers(0, 256, (96, 192, 3), dtype=np.uint8)).save(tmp_path / "train_img" / f"t{i}.png")
        lab = rng.integers(0, 19, (96, 192), dtype=np.uint8)
        lab[rng.random(lab.shape) < 0.1] = 255
        Image.fromarray(lab).save(tmp_path / "train_lab" / f"t{i}.png")
        lines.append(f"train_img/t{i}.png train_lab/t{i}.png")
    (tmp_path / "pseudo.lst").write_text("\n".join(lines) + "\n")
    vals = []
    for i in range(2):
        name = f"frankfurt/frankfurt_00000{i}_leftImg8bit.png"
        Image.fromarray(rng.integers(0, 256, (1024, 2048, 3), dtype=np.uint8)).save(tmp_path / "val" / name)
        Image.fromarray(rng.integers(0, 34, (1024, 2048), dtype=np.uint8)).save(
            tmp_path / "gt" / f"frankfurt/frankfurt_00000{i}_gtFine_labelIds.png")
        vals.append(name)
    (tmp_path / "kit" / "val.txt").write_text("\n".join(vals) + "\n")
    l2t = [[i, i if i < 19 else 255] for i in range(34)] + [[-1, 255]]
    json.dump({"classes": 19, "label": [f"c{i}" for i in range(19)], "label2train": l2t}, open(tmp_path / "kit" / "info.json", "w"))


def test_train_tool_real_files_eval_and_snapshots(dev, tmp_path, capsys):
    Image = pytest.importorskip("PIL.Image")
    from simt_amd.tools import trainV2_simt as tool
    _make_dataset(tmp_path, Image)
    snap = str(tmp_path / "snap")
    argv = ["--data-dir-target", str(tmp_path), "--data-list-target", str(tmp_path / "pseudo.lst"), "--input-size-target", "129,65",
            "--batch-size", "2", "--num-steps", "50", "--num-steps-stop", "5", "--save-pred-every", "2", "--print-every", "1",
            "--open-classes", "3", "--learning-rate", "6e-4", "--learning-rate-T", "6e-3", "--from-scratch", "--restore-from", "",
            "--snapshot-dir", snap, "--data-dir-val", str(tmp_path), "--data-list-val", str(tmp_path / "kit" / "val.txt"),
            "--gt-dir-val", str(tmp_path / "gt"), "--devkit-dir", str(tmp_path / "kit"), "--num-workers", "2", "--random-mirror"]
    tool.main(argv)
    out = capsys.readouterr().out
    assert out.count("Begin evaluation") == 1 and out.count("===> mIoU:") == 1            # iteration 2 (iteration 4 is the stop: save + break)
    assert "iter =        4/" in out and "Place_loss" in out and "save model" in out
    final = os.path.join(snap, "GTA5_5.pth")
    assert os.path.exists(final)
    sd = torch.load(final)
    assert len(sd) == 656 and int(sd["bn1.num_batches_tracked"]) == 5 and sd["conv1.weight"].shape == (64, 3, 7, 7)
    best = glob.glob(os.path.join(snap, "GTA5_iter*_mIoU*.pth"))
    assert len(best) == 1 and "GTA5_iter2_mIoU" in best[0]
    assert all(torch.isfinite(v).all() for k, v in sd.items() if v.dtype.is_floating_point)


def test_warmup_tool_real_files_eval_and_snapshots(dev, tmp_path, capsys):
    """The warm-up twin (trainV1_warmup.py:156-256): real files, `evaluate_warmup` every --save-pred-every iterations, the best-mIoU
    rotation with the reference's file names (`GTA5_BAPA_warmup_iter<i>_mIoU<m>.pth`, :249-254), the final `GTA5_<stop>.pth` (:236-239),
    and a reference-style command line that spells out flags the reference parses and never reads."""
    Image = pytest.importorskip("PIL.Image")
    from simt_amd.tools import trainV1_warmup as tool
    _make_dataset(tmp_path, Image)
    snap = str(tmp_path / "snap")
    argv = ["--model", "DeepLab", "--target", "cityscapes", "--data-dir", "/nonexistent", "--data-list", "x.txt", "--ignore-label", "255",
            "--input-size", "1024,512", "--set", "train", "--not-restore-last", "--open-classes", "15", "--learning-rate-T", "2.5e-4",
            "--data-dir-target", str(tmp_path), "--data-list-target", str(tmp_path / "pseudo.lst"), "--input-size-target", "129,65",
            "--batch-size", "2", "--num-steps", "50", "--num-steps-stop", "6", "--save-pred-every", "2", "--print-every", "1",
            "--learning-rate", "2.5e-4", "--from-scratch", "--restore-from", "", "--snapshot-dir", snap,
            "--data-dir-val", str(tmp_path), "--data-list-val", str(tmp_path / "kit" / "val.txt"),
            "--gt-dir-val", str(tmp_path / "gt"), "--devkit-dir", str(tmp_path / "kit"), "--num-workers", "2", "--random-mirror"]
    tool.main(argv)
    out = capsys.readouterr().out
    assert out.count("Begin evaluation") == 2 and out.count("===> mIoU:") == 2             # iterations 2 and 4 (5 is the stop: save + break)
    assert "iter =        5/" in out and "loss_seg1" in out and "save model" in out
    final = os.path.join(snap, "GTA5_6.pth")
    assert os.path.exists(final)
    sd = torch.load(final)
    assert int(sd["bn1.num_batches_tracked"]) == 6 and sd["layer6.conv2d_list.0.weight"].shape[0] == 19
    assert not any(k.startswith(("layer5_1", "layer6_1")) for k in sd)                    # DeeplabMulti(num_classes) without open-set heads
    best = glob.glob(os.path.join(snap, "GTA5_BAPA_warmup_iter*_mIoU*.pth"))
    assert len(best) == 1                                                                  # rotation: only the best one is kept
    assert all(torch.isfinite(v).all() for k, v in sd.items() if v.dtype.is_floating_point)
    # no validation set: a rolling periodic snapshot, one file
    snap2 = str(tmp_path / "snap2")
    i = argv.index("--snapshot-dir")
    argv2 = argv[:i] + ["--snapshot-dir", snap2] + argv[i + 2:]
    j = argv2.index("--data-dir-val")
    argv2 = argv2[:j] + argv2[j + 2:]
    tool.main(argv2)
    assert sorted(os.listdir(snap2)) == ["GTA5_6.pth", "GTA5_BAPA_warmup_iter4.pth"]


def test_train_tool_refuses_missing_data_and_checkpoint(dev, tmp_path):
    from simt_amd.tools import trainV2_simt as tool
    base = ["--input-size-target", "129,65", "--batch-size", "1", "--num-steps-stop", "1", "--snapshot-dir", str(tmp_path / "s")]
    with pytest.raises(FileNotFoundError):
        tool.main(base + ["--synthetic" if False else "--data-dir-target", str(tmp_path / "nope"), "--restore-from", str(tmp_path / "no.pth")])
    with pytest.raises(SystemExit):
        tool.main(base + ["--data-dir-target", str(tmp_path / "nope"), "--from-scratch"])


def test_gpu_loader_iter_size2_equals_resident_batches(dev, tmp_path):
    """--iter-size 2 with REAL files: the training loop pulls two micro-batches from the GpuLoader before it enqueues the step
    that reads them.  A trainer fed that way must follow bit for bit the trainer fed with resident copies of the same batches (round 2's
    prefetcher refilled the first micro-batch's slot before the step had read it: batch 1 lost or torn, batch 3 trained twice)."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import simt_oracle as so
    from simt_amd.data.pipeline import IMG_MEAN, GpuLoader
    from simt_amd.dataset.cityscapes_dataset import cityscapesPseudo
    from simt_amd.step import Hyper, SimTTrainer
    _make_dataset(tmp_path, Image)
    rng = np.random.default_rng(3)
    lines = open(tmp_path / "pseudo.lst").read().split()
    for i in range(4, 8):                                       # 8 training frames: 4 steps x 2 micro-batches of 1
        Image.fromarray(rng.integers(0, 256, (96, 192, 3), dtype=np.uint8)).save(tmp_path / "train_img" / f"t{i}.png")
        lab = rng.integers(0, 19, (96, 192), dtype=np.uint8)
        Image.fromarray(lab).save(tmp_path / "train_lab" / f"t{i}.png")
    (tmp_path / "pseudo.lst").write_text("".join(f"train_img/t{i}.png train_lab/t{i}.png\n" for i in range(8)))
    ds = cityscapesPseudo(str(tmp_path), str(tmp_path / "pseudo.lst"), crop_size=(129, 65), scale=False, mirror=False, mean=IMG_MEAN)
    layers, K, its = (1, 1, 1, 1), 3, 2
    CD = so.load_class_dist()
    st = so.recipe_state(so.state_shapes(19, K, True, layers=layers), seed=11, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=12, head_scale=8.0)
    kw = dict(open_classes=K, lr=6e-4, lr_T=6e-3, iter_size=its)

    def trainer():
        return SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), Hyper(**kw), CD.numpy(), 1, 65, 129,
                           dtype=torch.float32, device=dev, layers=layers)
    # reference run: every batch copied to resident tensors (synchronised) before anything is trained
    batches = []
    for img, lab, _s, _n in GpuLoader(ds, 1, shuffle=True, num_workers=2, device=dev, seed=5, epochs=1):
        torch.cuda.synchronize()
        batches.append((img.clone(), lab.clone()))
    assert len(batches) == 8
    ta = trainer()
    ref = []
    for s_ in range(4):
        ta.step([batches[2 * s_][0], batches[2 * s_ + 1][0]], [batches[2 * s_][1], batches[2 * s_ + 1][1]], s_)
        ref.append(ta.lout.clone())
    # the tool's pattern: pull `iter_size` micro-batches, then step, no synchronisation anywhere
    tb = trainer()
    busy = torch.zeros(32 << 20, device=dev)
    data = iter(GpuLoader(ds, 1, shuffle=True, num_workers=2, device=dev, seed=5, epochs=1, hold=its))
    got = []
    for s_ in range(4):
        mb = [next(data) for _ in range(its)]
        for _ in range(20):
            busy.add_(1.0)                                      # the step's kernels queue up behind other work
        tb.step([m[0] for m in mb], [m[1] for m in mb], s_)
        got.append(tb.lout.clone())
    torch.cuda.synchronize()
    for s_ in range(4):
        assert torch.equal(ref[s_], got[s_]), f"step {s_}: {ref[s_][:9].tolist()} vs {got[s_][:9].tolist()}"
    for n in ("layer3.0.conv1.weight", "layer6.conv2d_list.0.weight"):
        assert torch.equal(ta.params[n], tb.params[n])
