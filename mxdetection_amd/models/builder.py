"""Config -> detector (the README's "Modular Design" -- a model is a choice of backbone,
neck, heads named in an experiment file, /root/reference/README.md:7,13)."""


def build_detector(cfg, device="cuda"):
    from . import FasterRCNN, RetinaNet
    net, tr = cfg.network, cfg.TRAIN
    if net.type in ("faster_rcnn", "mask_rcnn"):
        model = FasterRCNN(device, depth=net.backbone_depth, num_classes=net.num_classes, seed=net.seed,
                           rois_per_image=tr.batch_rois, pre_nms_top_n=tr.rpn_pre_nms_top_n,
                           post_nms_top_n=tr.rpn_post_nms_top_n, with_mask=(net.type == "mask_rcnn"))
    elif net.type == "retinanet":
        model = RetinaNet(device, depth=net.backbone_depth, num_classes=net.num_classes - 1, seed=net.seed)
    else:
        raise ValueError("unknown network.type %r" % (net.type,))
    if net.pretrained:
        from ..utils import load_pretrained_backbone
        load_pretrained_backbone(model, net.pretrained, depth=net.backbone_depth)
    return model


def build_loader(cfg, device="cuda", rank=0, world=1, train=True, with_masks=None):
    """Config -> (roidb, class names, DetectionLoader)."""
    import numpy as np
    from ..datasets import append_flipped, filter_roidb, load_coco_roidb, synthetic_roidb
    from ..datasets.loader import DetectionLoader
    from ..datasets.synthetic import synthetic_reader
    from ..process_data import BatchPreprocessor
    ds = cfg.dataset
    if ds.type == "synthetic":
        roidb, names, reader = synthetic_roidb(ds.num_images, seed=1, num_classes=cfg.network.num_classes - 1), None, synthetic_reader
    elif ds.type == "coco":
        roidb, names = load_coco_roidb(ds.ann_file, ds.image_dir)
        # frames are stored as uint8 [h,w,3] .npy arrays next to / instead of the JPEGs (no decoder in the image)
        reader = lambda e: np.load(e["image"] if e["image"].endswith(".npy") else e["image"].rsplit(".", 1)[0] + ".npy")  # noqa: E731
    elif ds.type == "voc":
        from ..datasets import load_voc_roidb
        roidb, names = load_voc_roidb(ds.ann_file, image_dir=ds.image_dir)       # ann_file = the Annotations directory
        reader = lambda e: np.load(e["image"].rsplit(".", 1)[0] + ".npy")  # noqa: E731
    else:
        raise ValueError("unknown dataset.type %r" % (ds.type,))
    if train:
        roidb = filter_roidb(roidb)
        if cfg.TRAIN.flip:
            roidb = append_flipped(roidb)
    fixed = tuple(ds.fixed_shape) if ds.fixed_shape else None
    pre = BatchPreprocessor(ds.target_size, ds.max_size, ds.pixel_means, ds.pixel_stds, ds.swap_rb,
                            pad_to=fixed, fit_inside=fixed is not None)
    tr = cfg.TRAIN
    loader = DetectionLoader(roidb, tr.batch_images if train else cfg.TEST.batch_images, device=device, rank=rank, world=world,
                             reader=reader, preprocessor=pre, g_max=ds.max_gt,
                             with_masks=(train and cfg.network.type == "mask_rcnn") if with_masks is None else with_masks,
                             shuffle=train and tr.shuffle, aspect_grouping=train and tr.aspect_grouping, seed=tr.seed)
    return roidb, names, loader
