#!/usr/bin/env python3
"""SimT training stage on MI355X: the reference's tools/trainV2_simt.py with the same command-line flags
(tools/trainV2_simt.py:72-157) driving `SimTTrainer` (simt_amd/step.py) -- the whole iteration body of the reference
(:308-436) runs as HIP kernels; this script is only the outer loop: flags, checkpoint restore by key filter (:248-255),
data, LR schedule, printing every 100 iterations (:438-441), snapshots (:447-450).

    python -m simt_amd.tools.trainV2_simt --open-classes 15 --learning-rate 6e-4 --learning-rate-T 6e-3 ...
    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 -m simt_amd.tools.trainV2_simt ...   (data parallel)

Data: `cityscapesPseudo(--data-dir-target, --data-list-target)` through `GpuLoader` (decode on host threads, Pillow-exact resize +
BGR-mean on the GPU, pinned double-buffered uploads; simt_amd/data/pipeline.py), shuffled, --random-mirror as the reference writes
it; `--synthetic` feeds Cityscapes-shaped synthetic batches instead.  Pointing --data-dir-target at a missing directory
without --synthetic is an error (a run that silently trains on noise would still write checkpoints that look like results).
Every --save-pred-every iterations: evaluate_simt on the validation set (--data-dir-val ...) and the best-mIoU snapshot rotation of
trainV2_simt.py:452-464.  Additions over the reference (all optional): --synthetic, --compute-dtype, --eval-dtype (fp32 like the
reference unless bf16 is asked for), --print-every, --data-dir-val,
--data-list-val, --gt-dir-val, --devkit-dir.
"""
import argparse
import os
import os.path as osp
import time

import torch

from simt_amd import model_spec as ms
from simt_amd.step import Hyper, SimTTrainer, lr_poly


def get_arguments(argv=None):
    p = argparse.ArgumentParser(description="SimT (DeepLab-ResNet) on MI355X")
    p.add_argument("--model", type=str, default="DeepLab")
    p.add_argument("--target", type=str, default="cityscapes")
    p.add_argument("--batch-size", type=int, default=1)
    p.add_argument("--iter-size", type=int, default=1)
    p.add_argument("--num-workers", type=int, default=4)
    p.add_argument("--data-dir", type=str, default="")
    p.add_argument("--data-list", type=str, default="../dataset/gta5_list/train.txt")
    p.add_argument("--ignore-label", type=int, default=255)
    p.add_argument("--input-size", type=str, default="1024,512")
    p.add_argument("--data-dir-target", type=str, default="")
    p.add_argument("--data-list-target", type=str, default="../dataset/cityscapes_list/pseudo_bapa.lst")
    p.add_argument("--input-size-target", type=str, default="1024,512")
    p.add_argument("--is-training", action="store_true")
    p.add_argument("--learning-rate", type=float, default=2.5e-4)
    p.add_argument("--learning-rate-T", type=float, default=2.5e-4)
    p.add_argument("--lambda-seg", type=float, default=0.1)
    p.add_argument("--Threshold-high", type=float, default=0.8)
    p.add_argument("--Threshold-low", type=float, default=0.2)
    p.add_argument("--lambda-Place", type=float, default=0.1)
    p.add_argument("--lambda-Convex", type=float, default=0.5)
    p.add_argument("--lambda-Volume", type=float, default=0.1)
    p.add_argument("--lambda-Anchor", type=float, default=0.5)
    p.add_argument("--momentum", type=float, default=0.9)
    p.add_argument("--not-restore-last", action="store_true")
    p.add_argument("--num-classes", type=int, default=19)
    p.add_argument("--open-classes", type=int, default=15)
    p.add_argument("--num-steps", type=int, default=250000)
    p.add_argument("--num-steps-stop", type=int, default=40000)
    p.add_argument("--power", type=float, default=0.9)
    p.add_argument("--random-mirror", action="store_true")
    p.add_argument("--random-scale", action="store_true")
    p.add_argument("--random-seed", type=int, default=1234)
    p.add_argument("--restore-from", type=str, default="../snapshots/resnet_pretrain.pth")
    p.add_argument("--save-pred-every", type=int, default=1000)
    p.add_argument("--snapshot-dir", type=str, default="../snapshots/SimT/")
    p.add_argument("--weight-decay", type=float, default=0.0005)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--set", type=str, default="train")
    p.add_argument("--log-dir", type=str, default="./log/")
    # additions
    p.add_argument("--synthetic", action="store_true", help="synthetic Cityscapes-shaped batches")
    p.add_argument("--compute-dtype", choices=["bf16", "f32"], default="bf16")
    p.add_argument("--eval-dtype", choices=["f32", "bf16"], default="f32",
                   help="arithmetic of the periodic evaluation: fp32 like the reference (evaluate_cityscapes.py:96-162); bf16 is a labelled opt-in")
    p.add_argument("--print-every", type=int, default=100)
    p.add_argument("--data-dir-val", type=str, default="", help="Cityscapes root for the in-loop evaluation (evaluate_cityscapes.py:26)")
    p.add_argument("--data-list-val", type=str, default="../dataset/cityscapes_list/val.txt")
    p.add_argument("--gt-dir-val", type=str, default="", help="directory of *_gtFine_labelIds.png (evaluate_cityscapes.py:140)")
    p.add_argument("--devkit-dir", type=str, default="../dataset/cityscapes_list")
    p.add_argument("--from-scratch", action="store_true", help="allow training from the constructor init (no --restore-from)")
    return p.parse_args(argv)


def restore(state, path, not_restore_last=False, strip_prefix=0, required=False):
    """Filter-by-key load of an AdaptSegNet-style checkpoint into a fresh state (trainV2_simt.py:248-255).  strip_prefix=6: the warm-up
    stage's `k[6:]` (trainV1_warmup.py:177, checkpoints saved from a wrapped module) -- a key is accepted with or without the prefix.
    required: a missing file or zero matching tensors is an error (the reference crashes in torch.load; silently training from the
    constructor init would still produce checkpoints that look like results)."""
    if not path or not osp.exists(path):
        if required:
            raise FileNotFoundError(f"--restore-from {path!r} does not exist (pass --from-scratch to train from the constructor init)")
        return 0
    saved = torch.load(path, map_location="cpu")
    n = 0
    for k, v in saved.items():
        for cand in ((k, k[strip_prefix:]) if strip_prefix else (k,)):
            if cand in state and (not not_restore_last or not cand.startswith(("layer5", "layer6"))) and state[cand].shape == v.shape:
                state[cand] = v.clone()
                n += 1
                break
    if required and n == 0:
        raise RuntimeError(f"--restore-from {path!r}: no tensor matched the model's keys / shapes")
    return n


def save_atomic(obj, path):
    """torch.save to a temporary name, then os.replace: a crash or a full disk during the save leaves the previous file intact."""
    tmp = path + ".tmp"
    torch.save(obj, tmp)
    os.replace(tmp, path)


class SnapshotKeeper:
    """The snapshot rotation of trainV2_simt.py:452-464 / trainV1_warmup.py:243-256: after an evaluation keep ONE file
    `<stem><iter>_mIoU<mIoU>.pth` for the best mIoU so far; without a validation set (the reference hard-codes one) keep ONE rolling
    periodic file `<stem><iter>.pth`.  The new file is complete on disk (save_atomic) BEFORE the old one is removed."""

    def __init__(self, snapshot_dir, stem):
        self.dir, self.stem = snapshot_dir, stem
        self.best_mIoU, self.best_iter, self.rolling_iter = 0, 0, None

    def best(self, state_dict, i_iter, mIoU):
        if not mIoU > self.best_mIoU:
            return False
        old_file = osp.join(self.dir, self.stem + str(self.best_iter) + "_mIoU" + str(self.best_mIoU) + ".pth")
        print("Saving model with mIoU: ", mIoU)
        save_atomic(state_dict, osp.join(self.dir, self.stem + str(i_iter) + "_mIoU" + str(mIoU) + ".pth"))
        if os.path.exists(old_file):
            os.remove(old_file)
        self.best_mIoU, self.best_iter = mIoU, i_iter
        return True

    def rolling(self, state_dict, i_iter):
        save_atomic(state_dict, osp.join(self.dir, self.stem + str(i_iter) + ".pth"))
        if self.rolling_iter is not None and self.rolling_iter != i_iter:
            old_file = osp.join(self.dir, self.stem + str(self.rolling_iter) + ".pth")
            if os.path.exists(old_file):
                os.remove(old_file)
        self.rolling_iter = i_iter


def batches(args, B, H, W, cd, rank, world, dev):
    """-> iterator of (image f32 [B,3,H,W], label i64 [B,H,W]) resident on the device."""
    if args.synthetic:
        def synth():
            it = 0
            while True:                                     # one global sequence of seeds, dealt round-robin to the ranks
                yield ms.synthetic_batch(B, H, W, cd, seed=args.random_seed + it * world + rank, device=dev)
                it += 1
        return synth()
    if not args.data_dir_target or not osp.isdir(args.data_dir_target):
        raise SystemExit(f"--data-dir-target {args.data_dir_target!r} is not a directory; pass --synthetic for synthetic batches")
    from simt_amd.data.pipeline import IMG_MEAN, GpuLoader
    from simt_amd.dataset.cityscapes_dataset import cityscapesPseudo
    ds = cityscapesPseudo(args.data_dir_target, args.data_list_target, crop_size=(W, H), scale=False, mirror=args.random_mirror, mean=IMG_MEAN)
    loader = GpuLoader(ds, B, shuffle=True, num_workers=args.num_workers, device=dev, seed=args.random_seed, rank=rank, world=world,
                       hold=max(1, getattr(args, "iter_size", 1)))     # the loop keeps iter_size micro-batches alive per step
    return ((img, lab) for (img, lab, _sizes, _names) in loader)


def main(argv=None):
    args = get_arguments(argv)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(args.gpu)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("trainV2_simt needs a GPU: the SimT hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        from simt_amd.engine import reserve_streams
        reserve_streams(dev)                                    # the plan's streams take their hardware queues before RCCL makes its own
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")      # the conv tile lists leave 20 CUs to the collective's kernels (engine.TrunkPlan.cu_budget)
        dist.init_process_group("nccl", device_id=dev)
        pg = dist.group.WORLD
    w, h = map(int, args.input_size_target.split(","))
    C, K = args.num_classes, args.open_classes
    state = ms.reference_init(ms.state_shapes(C, K, True), seed=args.random_seed)
    fixed = ms.reference_init(ms.state_shapes(C, 0, False), seed=args.random_seed)
    n1 = restore(state, args.restore_from, args.not_restore_last, required=not (args.synthetic or args.from_scratch))
    n2 = restore(fixed, args.restore_from, required=not (args.synthetic or args.from_scratch))
    cd = ms.load_class_dist("bapa")
    hp = Hyper(num_classes=C, open_classes=K, th_high=args.Threshold_high, th_low=args.Threshold_low,
               lambda_seg=args.lambda_seg, lambda_place=args.lambda_Place, lambda_convex=args.lambda_Convex,
               lambda_volume=args.lambda_Volume, lambda_anchor=args.lambda_Anchor, iter_size=args.iter_size,
               lr=args.learning_rate, lr_T=args.learning_rate_T, momentum=args.momentum,
               weight_decay=args.weight_decay, power=args.power, num_steps=args.num_steps)
    dtype = torch.bfloat16 if args.compute_dtype == "bf16" else torch.float32
    eval_dtype = torch.bfloat16 if args.eval_dtype == "bf16" else torch.float32
    tr = SimTTrainer(state, fixed, ms.ntm_init(C, K, args.random_seed + 1), ms.ntm_init(C, K, args.random_seed + 2), hp, cd,
                     args.batch_size, h, w, dtype=dtype, device=dev, process_group=pg)
    if rank == 0:
        print(f"restored {n1}/{n2} tensors from {args.restore_from}; {world} GPU(s), batch {args.batch_size}/GPU, "
              f"{h}x{w}, {args.compute_dtype}, K={K}")
        os.makedirs(args.snapshot_dir, exist_ok=True)
    data = batches(args, args.batch_size, h, w, cd, rank, world, dev)
    evaluator, keeper = None, SnapshotKeeper(args.snapshot_dir, "GTA5_iter")
    t0 = time.time()
    for i_iter in range(args.num_steps):
        mb = [next(data) for _ in range(args.iter_size)]           # gradient accumulation: iter_size micro-batches per step
        img, lab = ([m[0] for m in mb], [m[1] for m in mb]) if args.iter_size > 1 else mb[0]
        tr.step(img, lab, i_iter)
        if i_iter % args.print_every == 0:
            l = tr.losses()                        # every rank (DP: a bad-label error is raised on all of them together)
            if rank == 0:
                print("iter = {0:8d}/{1:8d}, loss_seg_p = {2:.3f} loss_seg_y = {3:.3f} Convex = {4:.3f} Volume = {5:.3f} "
                      "Anchor = {6:.3f} Place_loss = {7:.3f}  lr = {8:.2e}  ({9:.1f} img/s)".format(
                          i_iter, args.num_steps, l["loss_p1"] + l["loss_p2"], l["loss_y1"] + l["loss_y2"], l["convex"], l["volume"],
                          l["anchor"], l["place"], lr_poly(args.learning_rate, i_iter, args.num_steps, args.power),
                          args.batch_size * world * (i_iter + 1) / max(time.time() - t0, 1e-9)))               # :438-441 (p1+p2, y1+y2)
        if i_iter >= args.num_steps_stop - 1:
            if rank == 0:
                print("save model ...")
                save_atomic(tr.state_dict(), osp.join(args.snapshot_dir, "GTA5_" + str(args.num_steps_stop) + ".pth"))   # :447-450
            break
        if i_iter % args.save_pred_every == 0 and i_iter != 0 and args.data_dir_val:
            # :452-464: evaluate, keep only the best-mIoU snapshot
            from simt_amd.tools.evaluate_cityscapes import Evaluator, evaluate_simt
            if evaluator is None:
                evaluator = Evaluator(tr.params, num_classes=C, open_classes=K, dtype=eval_dtype, device=dev)
            if rank == 0:
                print(time.strftime("%Y-%m-%d %H:%M:%S"), "  Begin evaluation on iter {0:8d}/{1:8d}  ".format(i_iter, args.num_steps))
            mIoU = evaluate_simt(tr.params, args.data_dir_val, args.data_list_val, args.gt_dir_val, args.devkit_dir, num_classes=C,
                                 open_classes=K, device=dev, dtype=eval_dtype, evaluator=evaluator, rank=rank, world=world, process_group=pg)
            if rank == 0:
                print("Finish Evaluation: " + time.asctime(time.localtime(time.time())))
                keeper.best(tr.state_dict(), i_iter, mIoU)
        elif i_iter % args.save_pred_every == 0 and i_iter != 0 and rank == 0:
            # no validation set given (the reference hard-codes one, :452-464): without an evaluation there is no best-mIoU snapshot,
            # so keep a rolling periodic one -- a crash must not lose the run
            keeper.rolling(tr.state_dict(), i_iter)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
