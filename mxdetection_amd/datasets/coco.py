"""COCO `instances_*.json` -> roidb (README.md:21; MXNet-lineage role: `datasets/coco.py` `gt_roidb` +
`append_flipped_images` + `filter_roidb`). Pure json/numpy: pycocotools is not needed to READ annotations.

A roidb entry is a dict:
  image      path (or key) handed to the loader's image reader
  height, width
  boxes      [G,4] f32, inclusive pixel corners (x1,y1,x2,y2)
  gt_classes [G] i32, 1..num_classes in sorted-category-id order (0 = background)
  polygons   list (per box) of lists of flat [x0,y0,...] polygons ([] when the annotation carries RLE or nothing)
  flipped    bool
"""
import json
import os

import numpy as np


def load_coco_roidb(ann_file, image_dir="", keep_crowd=False, min_area=0.0):
    with open(ann_file, "r") as f:
        ds = json.load(f)
    cat_ids = sorted(c["id"] for c in ds["categories"])
    cat_to_cls = {cid: i + 1 for i, cid in enumerate(cat_ids)}
    class_names = ["__background__"] + [next(c["name"] for c in ds["categories"] if c["id"] == cid) for cid in cat_ids]
    by_image = {}
    for a in ds.get("annotations", []):
        by_image.setdefault(a["image_id"], []).append(a)
    roidb = []
    for im in sorted(ds["images"], key=lambda r: r["id"]):
        w, h = int(im["width"]), int(im["height"])
        boxes, classes, polys = [], [], []
        for a in sorted(by_image.get(im["id"], []), key=lambda r: r["id"]):
            if a.get("iscrowd", 0) and not keep_crowd:
                continue
            x, y, bw, bh = [float(v) for v in a["bbox"]]
            # lineage sanitising: clip to the frame, inclusive corners, drop degenerate boxes
            x1, y1 = max(0.0, x), max(0.0, y)
            x2 = min(w - 1.0, x1 + max(0.0, bw - 1.0))
            y2 = min(h - 1.0, y1 + max(0.0, bh - 1.0))
            if a.get("area", bw * bh) <= min_area or x2 < x1 or y2 < y1:
                continue
            boxes.append([x1, y1, x2, y2])
            classes.append(cat_to_cls[a["category_id"]])
            seg = a.get("segmentation", [])
            polys.append([list(map(float, p)) for p in seg if len(p) >= 6] if isinstance(seg, list) else [])
        roidb.append({
            "image": os.path.join(image_dir, im["file_name"]) if image_dir else im["file_name"],
            "id": im["id"], "height": h, "width": w,
            "boxes": np.asarray(boxes, np.float32).reshape(-1, 4),
            "gt_classes": np.asarray(classes, np.int32),
            "polygons": polys,
            "flipped": False,
        })
    return roidb, class_names


def append_flipped(roidb):
    """Doubles the roidb with horizontally mirrored entries (the pixels are mirrored by the preprocess kernel; boxes and
    polygons are mirrored when the batch is assembled, see process_data.transform_boxes)."""
    out = list(roidb)
    for r in roidb:
        f = dict(r)
        f["flipped"] = True
        out.append(f)
    return out


def filter_roidb(roidb):
    """Training drops images without any usable box."""
    return [r for r in roidb if r["boxes"].shape[0] > 0]
