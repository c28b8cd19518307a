"""Training driver: config -> loader -> captured training step -> checkpoints (the lineage's `train_end2end`-style
entry point). One process per GPU:

    python tools/train.py --cfg configs/faster_rcnn_r50_fpn.yaml [KEY=VALUE ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py --cfg ...
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default=None)
    ap.add_argument("overrides", nargs="*", help="SECTION.key=value")
    args = ap.parse_args()
    import torch
    from mxdetection_amd.models.builder import build_detector, build_loader
    from mxdetection_amd.utils import WarmupMultiFactorScheduler, epoch_steps, load_config, scaled_lr
    cfg = load_config(args.cfg, args.overrides)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    tr = cfg.TRAIN
    model = build_detector(cfg)
    model.enable_wgrad_stream()
    if hasattr(model, "enable_branch_stream"):
        model.enable_branch_stream()
    model.enable_grouped_wgrad()
    if tr.resume:
        model.load_checkpoint(tr.resume)
    if dist is not None:
        model.enable_data_parallel(world)
        model.broadcast_parameters(0)
    roidb, _, loader = build_loader(cfg, rank=rank, world=world, train=True)
    iters_per_epoch = len(loader)
    base_lr = scaled_lr(tr.lr, tr.batch_images * world) if tr.lr_reference_batch == 16 else tr.lr * tr.batch_images * world / tr.lr_reference_batch
    sched = WarmupMultiFactorScheduler(base_lr, epoch_steps(tr.lr_step, iters_per_epoch, tr.begin_epoch), tr.lr_factor,
                                       tr.warmup_step if (tr.warmup and tr.begin_epoch == 0) else 0,
                                       tr.warmup_lr * base_lr / tr.lr, tr.warmup_mode)
    use_graph = tr.graph and bool(cfg.dataset.fixed_shape)
    if rank == 0:
        print("train: %d images (%d iters/epoch at global batch %d), base lr %.5f, %s launches" %
              (len(roidb), iters_per_epoch, tr.batch_images * world, base_lr, "hipGraph" if use_graph else "eager"), flush=True)
    it, captured = 0, False
    off = rank * tr.batch_images
    t0, seen = time.perf_counter(), 0
    for epoch in range(tr.begin_epoch, tr.end_epoch):
        loader.set_epoch(epoch)
        for batch in loader:
            lr = sched(it)
            img, gt, info, mk = batch["image"], batch["gt_boxes"], batch["im_info"], batch.get("gt_masks")
            if use_graph and not captured:
                model.capture(img, gt, info, lr=lr, image_offset=off, gt_masks=mk, momentum=tr.momentum, wd=tr.wd)
                captured = True
            if use_graph:
                losses = model.replay(img, gt, info, it, gt_masks=mk, lr=lr)
            else:
                losses = model.train_step(img, gt, info, step=it, image_offset=off, lr=lr, gt_masks=mk,
                                          momentum=tr.momentum, wd=tr.wd)
            it += 1
            seen += tr.batch_images * world
            if rank == 0 and tr.log_period and it % tr.log_period == 0:
                vals = [float(v) for v in torch.cat(list(losses)).cpu().numpy()]
                dt = time.perf_counter() - t0
                print("epoch %d iter %d lr %.5f losses %s  %.1f img/s" % (epoch, it, lr, ["%.4f" % v for v in vals], seen / dt),
                      flush=True)
                t0, seen = time.perf_counter(), 0
            if tr.max_iters and it >= tr.max_iters:
                break
        if rank == 0 and tr.checkpoint_prefix and (epoch + 1) % tr.checkpoint_period == 0:
            torch.cuda.synchronize()
            os.makedirs(os.path.dirname(tr.checkpoint_prefix) or ".", exist_ok=True)
            path = "%s-%04d.params" % (tr.checkpoint_prefix, epoch + 1)       # MXNet's prefix-epoch naming
            model.save_checkpoint(path)
            print("saved", path, flush=True)
        if tr.max_iters and it >= tr.max_iters:
            break
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
