"""Evaluation driver: checkpoint -> detections -> COCO box AP (the lineage's `test.py` role).

    python tools/test.py --cfg configs/faster_rcnn_r50_fpn.yaml --params output/faster_rcnn_r50_fpn-0012.params
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default=None)
    ap.add_argument("--params", default="")
    ap.add_argument("--random-init", action="store_true",
                    help="evaluate the randomly initialised model (plumbing checks only); without it --params is required")
    ap.add_argument("--max-images", type=int, default=0)
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args()
    import torch
    from mxdetection_amd.core.evaluation import coco_bbox_eval, coco_segm_eval, detections_to_coco
    from mxdetection_amd.models.builder import build_detector, build_loader
    from mxdetection_amd.utils import load_config
    cfg = load_config(args.cfg, list(args.overrides) + ["TRAIN.flip=false"])
    model = build_detector(cfg)
    if args.params:
        model.load_checkpoint(args.params, strict=True)      # raises on any parameter the file lacks
    elif not args.random_init:
        sys.exit("tools/test.py: no --params given (use --random-init to evaluate untrained weights on purpose)")
    segm = cfg.network.type == "mask_rcnn"
    roidb, _, loader = build_loader(cfg, train=False, with_masks=segm)
    te = cfg.TEST
    order = loader.rank_batches()
    gts, dts, seen = [], [], set()
    sgts, sdts = [], []          # mask evaluation, in the frame of the network input (both sides rasterised there)
    for k, batch in enumerate(loader):
        if segm:
            dets, num, masks = model.predict(batch["image"], batch["im_info"], te.score_thresh, te.nms, te.max_per_image,
                                             with_masks=True)
        else:
            dets, num = model.predict(batch["image"], batch["im_info"], te.score_thresh, te.nms, te.max_per_image)
        ids = [int(i) for i in order[k]]
        fresh = [n for n, i in enumerate(ids) if i not in seen]          # the last batch wraps around
        scales = batch["im_info"][:, 2].cpu().numpy().tolist()
        res = detections_to_coco(dets, num, [roidb[i]["id"] for i in ids], scales)
        keep_ids = {roidb[ids[n]]["id"] for n in fresh}
        dts += [r for r in res if r["image_id"] in keep_ids]
        if segm:
            dn, nn, mk, gm = dets.cpu().numpy(), num.cpu().numpy(), masks.cpu().numpy(), batch["gt_masks"].cpu().numpy()
            for n in fresh:
                e = roidb[ids[n]]
                for j in range(int(nn[n])):
                    sdts.append({"image_id": e["id"], "category_id": int(dn[n, j, 5]), "score": float(dn[n, j, 4]), "mask": mk[n, j] > 0})
                for g in range(min(e["boxes"].shape[0], gm.shape[1])):
                    sgts.append({"image_id": e["id"], "category_id": int(e["gt_classes"][g]), "mask": gm[n, g] > 0})
        for n in fresh:
            e = roidb[ids[n]]
            seen.add(ids[n])
            for b, c in zip(e["boxes"], e["gt_classes"]):
                gts.append({"image_id": e["id"], "category_id": int(c), "bbox": [float(b[0]), float(b[1]), float(b[2] - b[0] + 1),
                                                                               float(b[3] - b[1] + 1)]})
        if args.max_images and len(seen) >= args.max_images:
            break
    torch.cuda.synchronize()
    out = {"images": len(seen), "detections": len(dts), **coco_bbox_eval(gts, dts)}
    if segm:
        out["segm"] = coco_segm_eval(sgts, sdts)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
