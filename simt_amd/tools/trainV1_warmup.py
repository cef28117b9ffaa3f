#!/usr/bin/env python3
"""Warm-up stage on MI355X: the reference's tools/trainV1_warmup.py (plain CE on both heads, SGD over the whole net)
driven by `WarmupTrainer` (simt_amd/step.py).  EVERY flag of the reference (trainV1_warmup.py:60-150) is accepted, the periodic
`evaluate_warmup` + best-mIoU snapshot rotation of :241-256 keeps the reference's file names; data: `cityscapesPseudo` through the device input pipeline (simt_amd/data/pipeline.py) or, with
--synthetic, Cityscapes-shaped synthetic batches.  --restore-from must exist and match (the reference's `k[6:]` prefix strip of :177
is honoured) unless --from-scratch is given.

    python -m simt_amd.tools.trainV1_warmup --learning-rate 2.5e-4 --input-size-target 1024,512 --num-steps-stop 40000
"""
import argparse
import os
import os.path as osp
import time

import torch

from simt_amd import model_spec as ms
from simt_amd.step import Hyper, WarmupTrainer, lr_poly
from simt_amd.tools.trainV2_simt import SnapshotKeeper, batches, restore, save_atomic


def get_arguments(argv=None):
    """Every flag of the reference (trainV1_warmup.py:60-150), same names / types / defaults; the ones the reference parses and never
    reads (--model, --target, --data-dir, --data-list, --ignore-label, --input-size, --is-training, --learning-rate-T,
    --not-restore-last, --open-classes, --random-scale, --set, --log-dir) are parsed and ignored here too."""
    p = argparse.ArgumentParser(description="DeepLab-ResNet warm-up on MI355X")
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
    p.add_argument("--snapshot-dir", type=str, default="../snapshots/")
    p.add_argument("--weight-decay", type=float, default=0.0005)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--set", type=str, default="train")
    p.add_argument("--log-dir", type=str, default="./log/")
    # additions (all optional)
    p.add_argument("--compute-dtype", choices=["bf16", "f32"], default="bf16")
    p.add_argument("--eval-dtype", choices=["f32", "bf16"], default="f32",
                   help="arithmetic of the periodic evaluation: fp32 like the reference (evaluate_cityscapes.py:96-162); bf16 is a labelled opt-in")
    p.add_argument("--print-every", type=int, default=100)
    p.add_argument("--synthetic", action="store_true", help="synthetic Cityscapes-shaped batches")
    p.add_argument("--from-scratch", action="store_true", help="allow training from the constructor init (no --restore-from)")
    p.add_argument("--data-dir-val", type=str, default="", help="Cityscapes root for the in-loop evaluation (evaluate_cityscapes.py:26)")
    p.add_argument("--data-list-val", type=str, default="../dataset/cityscapes_list/val.txt")
    p.add_argument("--gt-dir-val", type=str, default="", help="directory of *_gtFine_labelIds.png (evaluate_cityscapes.py:140)")
    p.add_argument("--devkit-dir", type=str, default="../dataset/cityscapes_list")
    return p.parse_args(argv)


def main(argv=None):
    args = get_arguments(argv)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(args.gpu)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("trainV1_warmup needs a GPU: no CPU fallback")
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
    if rank == 0:
        print("Start: " + time.asctime(time.localtime(time.time())))                       # :158
    w, h = map(int, args.input_size_target.split(","))
    C = args.num_classes
    state = ms.reference_init(ms.state_shapes(C, 0, False), seed=args.random_seed)
    # trainV1_warmup.py:177: keys of the pretrained checkpoint carry a 6-character prefix (`k[6:]`); shapes are filtered
    n = restore(state, args.restore_from, strip_prefix=6, required=not (args.synthetic or args.from_scratch))
    hp = Hyper(num_classes=C, open_classes=0, lambda_seg=args.lambda_seg, lr=args.learning_rate, iter_size=args.iter_size,
               momentum=args.momentum, weight_decay=args.weight_decay, power=args.power, num_steps=args.num_steps)
    dtype = torch.bfloat16 if args.compute_dtype == "bf16" else torch.float32
    eval_dtype = torch.bfloat16 if args.eval_dtype == "bf16" else torch.float32
    tr = WarmupTrainer(state, hp, args.batch_size, h, w, dtype=dtype, device=dev, process_group=pg)
    cd = ms.load_class_dist("bapa")
    if rank == 0:
        print(f"restored {n} tensors; {world} GPU(s), batch {args.batch_size}/GPU, {h}x{w}, {args.compute_dtype}")
        os.makedirs(args.snapshot_dir, exist_ok=True)                                       # :185-186
    data = batches(args, args.batch_size, h, w, cd, rank, world, dev)
    evaluator, keeper = None, SnapshotKeeper(args.snapshot_dir, "GTA5_BAPA_warmup_iter")
    t0 = time.time()
    for i_iter in range(args.num_steps):
        mb = [next(data) for _ in range(args.iter_size)]             # gradient accumulation: iter_size micro-batches per step
        img, lab = ([m[0] for m in mb], [m[1] for m in mb]) if args.iter_size > 1 else mb[0]
        tr.step(img, lab, i_iter)
        if i_iter % args.print_every == 0:              # :231-234 (every 100 iterations there)
            l = tr.losses()                        # every rank (DP: a bad-label error is raised on all of them together)
            if rank == 0:
                print("iter = {0:8d}/{1:8d}, loss_seg1 = {2:.3f} loss_seg2 = {3:.3f}  lr = {4:.2e}  ({5:.1f} img/s)".format(
                    i_iter, args.num_steps, l["loss_seg1"], l["loss_seg2"], lr_poly(args.learning_rate, i_iter, args.num_steps, args.power),
                    args.batch_size * world * (i_iter + 1) / max(time.time() - t0, 1e-9)))
        if i_iter >= args.num_steps_stop - 1:                         # :236-239
            if rank == 0:
                print("save model ...")
                save_atomic(tr.state_dict(), osp.join(args.snapshot_dir, "GTA5_" + str(args.num_steps_stop) + ".pth"))
            break
        if i_iter % args.save_pred_every == 0 and i_iter != 0 and args.data_dir_val:
            # :241-256: evaluate_warmup on the validation set, keep only the best-mIoU snapshot `GTA5_BAPA_warmup_iter<i>_mIoU<m>.pth`
            from simt_amd.tools.evaluate_cityscapes import Evaluator, evaluate_warmup
            if evaluator is None:
                evaluator = Evaluator(tr.params, num_classes=C, open_classes=0, dtype=eval_dtype, device=dev)
            if rank == 0:
                print(time.strftime("%Y-%m-%d %H:%M:%S"), "  Begin evaluation on iter {0:8d}/{1:8d}  ".format(i_iter, args.num_steps))
            mIoU = evaluate_warmup(tr.params, args.data_dir_val, args.data_list_val, args.gt_dir_val, args.devkit_dir, num_classes=C,
                                   device=dev, dtype=eval_dtype, evaluator=evaluator, rank=rank, world=world, process_group=pg)
            if rank == 0:
                print("Finish Evaluation: " + time.asctime(time.localtime(time.time())))
                keeper.best(tr.state_dict(), i_iter, mIoU)
        elif i_iter % args.save_pred_every == 0 and i_iter != 0 and rank == 0:
            # no validation set given (the reference hard-codes one, evaluate_cityscapes.py:26-28): a rolling periodic snapshot instead
            keeper.rolling(tr.state_dict(), i_iter)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
