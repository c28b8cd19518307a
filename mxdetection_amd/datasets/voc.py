"""PASCAL VOC annotations (one XML per image) -> roidb (README.md:21; MXNet-lineage role: `datasets/pascal_voc.py`).
Same entry layout as datasets/coco.py; VOC pixel coordinates are 1-based and inclusive, the roidb's are 0-based.
Parsed with xml.etree (no external entities are resolved)."""
import os
import xml.etree.ElementTree as ET

import numpy as np

VOC_CLASSES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
               "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")


def load_voc_roidb(ann_dir, image_set=None, image_dir="", classes=VOC_CLASSES, use_difficult=False, image_ext=".jpg"):
    """ann_dir: directory of <name>.xml files; image_set: optional text file listing the names to use (one per line).
    Returns (roidb, class names with "__background__" first). Difficult objects are dropped unless use_difficult."""
    if image_set:
        with open(image_set, "r") as f:
            names = [ln.split()[0] for ln in f if ln.strip()]
    else:
        names = sorted(fn[:-4] for fn in os.listdir(ann_dir) if fn.endswith(".xml"))
    cls_index = {c: i + 1 for i, c in enumerate(classes)}
    roidb = []
    for idx, name in enumerate(names):
        root = ET.parse(os.path.join(ann_dir, name + ".xml")).getroot()
        size = root.find("size")
        w, h = int(size.find("width").text), int(size.find("height").text)
        boxes, labels, difficult = [], [], []
        for obj in root.findall("object"):
            cname = obj.find("name").text.strip().lower()
            if cname not in cls_index:
                continue
            diff = int(obj.find("difficult").text) if obj.find("difficult") is not None else 0
            if diff and not use_difficult:
                continue
            bb = obj.find("bndbox")
            x1, y1, x2, y2 = [float(bb.find(k).text) - 1.0 for k in ("xmin", "ymin", "xmax", "ymax")]
            x1, y1, x2, y2 = max(0.0, x1), max(0.0, y1), min(w - 1.0, x2), min(h - 1.0, y2)
            if x2 < x1 or y2 < y1:
                continue
            boxes.append([x1, y1, x2, y2])
            labels.append(cls_index[cname])
            difficult.append(diff)
        fn = root.find("filename").text.strip() if root.find("filename") is not None else name + image_ext
        roidb.append({"image": os.path.join(image_dir, fn) if image_dir else fn, "id": idx, "name": name, "height": h, "width": w,
                      "boxes": np.asarray(boxes, np.float32).reshape(-1, 4), "gt_classes": np.asarray(labels, np.int32),
                      "difficult": np.asarray(difficult, np.int32), "polygons": [[] for _ in boxes], "flipped": False})
    return roidb, ["__background__"] + list(classes)
