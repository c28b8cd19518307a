"""Host side of the data pipeline (datasets, datasets/loader, process_data; README.md:21-23) and its oracle.

The reference ships no code or fixtures for these (README.md + LICENSE only) and cv2 / pycocotools are not in the
image: the resize is pinned by hand-computed known answers of the published OpenCV 8-bit bilinear arithmetic, parity
with an actual cv2 build is unpinned."""
import os

import numpy as np
import pytest

from mxdetection_amd.datasets import append_flipped, epoch_order, filter_roidb, load_coco_roidb, synthetic_roidb
from mxdetection_amd.datasets.loader import DetectionLoader
from mxdetection_amd.process_data import transform as T

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = np.load(os.path.join(GOLD, "det_small.npz"))


# ---- process_data: scale / shape / coordinates --------------------------------------------------------------------
def test_resize_scale_known_answers():
    assert T.resize_scale(480, 640) == pytest.approx(800 / 480)
    assert T.resized_shape(480, 640, T.resize_scale(480, 640)) == (800, 1067)
    assert T.resize_scale(640, 480) == pytest.approx(800 / 480)
    # the long side would pass 1333: 800/333*1000 = 2402 -> scale by the long side instead
    assert T.resize_scale(333, 1000) == pytest.approx(1.333)
    assert T.resized_shape(333, 1000, 1.333) == (444, 1333)
    assert T.resize_scale(800, 1333) == 1.0
    assert T.pad_shape([(800, 1067), (750, 1333)]) == (800, 1344)


def test_box_and_polygon_transforms():
    b = np.array([[10, 5, 29, 14], [0, 0, 59, 39]], np.float32)
    f = T.flip_boxes(b, 60)
    assert f.tolist() == [[30, 5, 49, 14], [0, 0, 59, 39]]
    assert np.array_equal(T.flip_boxes(f, 60), b)                       # involution
    t = T.transform_boxes(b, 2.0, True, 60)
    assert t.tolist() == [[60, 10, 98, 28], [0, 0, 118, 78]]
    p = T.transform_polygons([[10, 5, 30, 5, 30, 15]], 0.5, True, 60)
    assert p[0].tolist() == [[25, 2.5], [15, 2.5], [15, 7.5]]
    assert T.transform_boxes(np.zeros((0, 4), np.float32), 2.0, True, 60).shape == (0, 4)


# ---- oracle: 8-bit bilinear resize + normalise + pad ----------------------------------------------------------------
def test_resize_identity_and_known_answer(oracle):
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, (9, 12, 3), dtype=np.uint8)
    out, u8 = oracle.image_preprocess([im], [1.0], [False], 16, 16, (0, 0, 0), (1, 1, 1), swap_rb=False, return_u8=True)
    assert np.array_equal(u8[0], im)                                    # scale 1: coefficients (2048, 0)
    val = oracle.bf16_bits_to_f32(out[0])
    assert np.array_equal(val[:, :9, :12], im.transpose(2, 0, 1).astype(np.float32))   # 0..255 are exact in bf16
    assert not val[:, 9:, :].any() and not val[:, :, 12:].any()        # zero padding
    # 2x upscale of the row [0, 255]: source coordinates -0.25, 0.25, 0.75, 1.25 -> 0, 63.75, 191.25, 255
    row = np.zeros((1, 2, 3), np.uint8)
    row[0, 1] = 255
    _, u8 = oracle.image_preprocess([row], [2.0], [False], 8, 8, (0, 0, 0), (1, 1, 1), swap_rb=False, return_u8=True)
    assert u8[0].shape == (2, 4, 3) and u8[0][0, :, 0].tolist() == [0, 64, 191, 255] and np.array_equal(u8[0][0], u8[0][1])


def test_resize_flip_swap_normalise(oracle):
    rng = np.random.default_rng(1)
    im = rng.integers(0, 256, (14, 10, 3), dtype=np.uint8)
    mean, std = (100.0, 110.0, 120.0), (50.0, 60.0, 70.0)
    a, ua = oracle.image_preprocess([im], [1.3], [True], 32, 16, mean, std, swap_rb=True, return_u8=True)
    b, ub = oracle.image_preprocess([im[:, ::-1, ::-1]], [1.3], [False], 32, 16, mean, std, swap_rb=False, return_u8=True)
    assert np.array_equal(a, b) and np.array_equal(ua[0], ub[0])       # flag == pre-flipped / pre-swapped source
    dh, dw = ua[0].shape[:2]
    assert (dh, dw) == (18, 13)
    want = (ua[0].astype(np.float32).transpose(2, 0, 1) - np.float32(mean)[:, None, None]) / np.float32(std)[:, None, None]
    assert np.array_equal(a[0, :, :dh, :dw], oracle.f32_to_bf16_bits(want))
    # a constant frame stays constant at any scale
    c = np.full((7, 5, 3), 77, np.uint8)
    _, uc = oracle.image_preprocess([c], [2.71], [False], 32, 16, mean, std, return_u8=True)
    assert np.all(uc[0] == 77)


def test_preprocess_golden(oracle):
    out, u8 = oracle.image_preprocess([G["dp_im0"], G["dp_im1"]], G["dp_scales"].tolist(), [False, True], 32, 32,
                                      (123.68, 116.779, 103.939), (58.4, 57.1, 57.4), swap_rb=True, return_u8=True)
    assert np.array_equal(out, G["dp_out"]) and np.array_equal(u8[0], G["dp_u8_0"]) and np.array_equal(u8[1], G["dp_u8_1"])


# ---- oracle: polygon rasterisation ------------------------------------------------------------------------------------
def test_polygon_masks_known_answers(oracle):
    sq = [[np.array([[1, 1], [5, 1], [5, 4], [1, 4]], np.float32)]]
    v, s, f = T.pack_polygons([sq], 1, 2)
    assert s.tolist() == [0, 4] and f.tolist() == [0, 1, 1]
    m = oracle.polygon_masks(v, s, f, 2, 6, 8)
    want = np.zeros((6, 8), np.uint8)
    want[1:4, 1:5] = 1                                                  # centres (x+.5, y+.5) inside [1,5) x [1,4)
    assert np.array_equal(m[0], want) and not m[1].any()
    # two polygons of one instance: union; an outer ring with a reversed inner ring given as ONE polygon: hole
    two = [[np.array([[0, 0], [3, 0], [3, 3], [0, 3]], np.float32), np.array([[2, 2], [6, 2], [6, 5], [2, 5]], np.float32)]]
    v, s, f = T.pack_polygons([two], 1, 1)
    m = oracle.polygon_masks(v, s, f, 1, 6, 8)[0]
    assert m.sum() == 9 + 12 - 1 and m[2, 2] == 1
    assert np.array_equal(oracle.polygon_masks(G["pm_verts"], G["pm_start"], G["pm_first"], 3, 16, 24), G["pm_masks"])


# ---- datasets: COCO json -> roidb ---------------------------------------------------------------------------------------
def test_coco_roidb():
    roidb, names = load_coco_roidb(os.path.join(GOLD, "coco_tiny.json"), image_dir="imgs")
    assert names == ["__background__", "person", "dog", "bottle"]       # classes follow sorted category ids
    assert [r["id"] for r in roidb] == [3, 7, 9]
    r7 = roidb[1]
    assert r7["image"] == os.path.join("imgs", "a.npy") and (r7["height"], r7["width"]) == (40, 60)
    # crowd dropped; second box clipped to the frame: x2 = min(59, 50+19), y2 = min(39, 30+19)
    assert r7["boxes"].tolist() == [[10, 5, 29, 14], [50, 30, 59, 39]] and r7["gt_classes"].tolist() == [2, 1]
    assert len(r7["polygons"][1]) == 2
    r3 = roidb[0]
    assert r3["boxes"].tolist() == [[0, 2, 9, 7]] and r3["gt_classes"].tolist() == [3]    # negative x clipped; zero-area dropped
    assert roidb[2]["boxes"].shape == (0, 4)
    assert [r["id"] for r in filter_roidb(roidb)] == [3, 7]
    both = append_flipped(filter_roidb(roidb))
    assert [r["flipped"] for r in both] == [False, False, True, True]
    keep = load_coco_roidb(os.path.join(GOLD, "coco_tiny.json"), keep_crowd=True)[0]
    assert keep[1]["boxes"].shape[0] == 3 and keep[1]["polygons"][2] == []            # RLE segmentation: no polygons


# ---- datasets/loader: epoch order, rank slices, batch assembly --------------------------------------------------------
def test_epoch_order_properties():
    roidb = synthetic_roidb(37, seed=3)
    for gb in (2, 4, 16):
        o = epoch_order(roidb, gb, 0, seed=5)
        assert o.size % gb == 0 and set(o.tolist()) == set(range(37))    # every image, wrapped to whole batches
        assert np.array_equal(o, epoch_order(roidb, gb, 0, seed=5))       # pure function of (seed, epoch)
        assert not np.array_equal(o, epoch_order(roidb, gb, 1, seed=5))
        land = np.array([r["width"] >= r["height"] for r in roidb])[o].reshape(-1, gb)
        assert np.all(land.all(1) | (~land).all(1))                        # no batch mixes orientations
    o = epoch_order(roidb, 4, 0, shuffle=False)
    assert o[:37].tolist() == list(range(37)) and o[37:].tolist() == [0, 1, 2]
    assert epoch_order([], 4, 0).size == 0


def test_rank_slices_partition_the_global_batch():
    roidb = synthetic_roidb(50, seed=1)
    world, b = 4, 2
    loaders = [DetectionLoader(roidb, b, rank=r, world=world, seed=9) for r in range(world)]
    order = epoch_order(roidb, world * b, 0, seed=9).reshape(-1, world * b)
    got = np.concatenate([l.rank_batches() for l in loaders], axis=1)
    assert np.array_equal(got, order) and len(loaders[0]) == order.shape[0]
    for l in loaders:
        l.set_epoch(1)
    assert not np.array_equal(np.concatenate([l.rank_batches() for l in loaders], axis=1), order)


def test_assemble_host_batch():
    roidb = append_flipped(synthetic_roidb(8, seed=2))
    L = DetectionLoader(roidb, 2, with_masks=True, g_max=20, shuffle=False)
    hb = L.assemble([1, 9])                                              # the same image, plain and flipped
    assert hb.flips == [False, True] and np.array_equal(hb.frames[0], hb.frames[1])
    e = roidb[1]
    s = T.resize_scale(e["height"], e["width"])
    dh, dw = T.resized_shape(e["height"], e["width"], s)
    assert hb.im_info[0].tolist() == [dh, dw, np.float32(s)]
    assert hb.pad == ((800, 1344) if e["width"] >= e["height"] else (1344, 800))
    G_ = e["boxes"].shape[0]
    assert np.all(hb.gt[:, G_:] == -1) and np.array_equal(hb.gt[0, :G_, 4], e["gt_classes"])
    # flipped boxes mirror the plain ones in the resized frame: x1' + x2 = s * (w - 1)
    assert np.allclose(hb.gt[1, :G_, 0] + hb.gt[0, :G_, 2], s * (e["width"] - 1), atol=1e-3)
    assert np.array_equal(hb.gt[1, :G_, 1], hb.gt[0, :G_, 1])
    assert np.all(hb.gt[0, :G_, 2] <= dw - 1) and np.all(hb.gt[0, :G_, 3] <= dh - 1)
    v, ps, first = hb.poly
    assert first.shape == (2 * 20 + 1,) and first[-1] == ps.size - 1 == 2 * G_ and v.shape == (ps[-1], 2)


def test_producer_failure_reaches_the_consumer():
    """Host-side half of the iterator only (no device): a reader that raises must surface, not hang."""
    import queue
    import threading
    roidb = synthetic_roidb(4, seed=1)

    def bad_reader(e):
        raise IOError("cannot read %s" % e["image"])
    L = DetectionLoader(roidb, 2, reader=bad_reader, shuffle=False)
    with pytest.raises(IOError):
        L.assemble([0, 1])


def test_voc_roidb():
    from mxdetection_amd.datasets import VOC_CLASSES, load_voc_roidb
    d = os.path.join(GOLD, "voc_tiny")
    roidb, names = load_voc_roidb(d, image_dir="JPEGImages")
    assert names[0] == "__background__" and len(names) == 21 and names[9] == "chair" and names[15] == "person"
    assert [r["name"] for r in roidb] == ["000005", "000007"]
    r = roidb[0]
    assert (r["height"], r["width"]) == (375, 500) and r["image"] == os.path.join("JPEGImages", "000005.jpg")
    # 1-based inclusive -> 0-based; the difficult chair and the unknown class are dropped; names are case-insensitive
    assert r["boxes"].tolist() == [[262, 210, 323, 338], [0, 0, 499, 374]] and r["gt_classes"].tolist() == [9, 15]
    assert roidb[1]["boxes"].shape == (0, 4)
    keep = load_voc_roidb(d, use_difficult=True)[0][0]
    assert keep["boxes"].shape[0] == 3 and keep["difficult"].tolist() == [0, 1, 0]
    lst = os.path.join(d, "set.txt")
    try:
        with open(lst, "w") as f:
            f.write("000007\n")
        assert [x["name"] for x in load_voc_roidb(d, image_set=lst)[0]] == ["000007"]
    finally:
        os.remove(lst)
    assert len(VOC_CLASSES) == 20


def test_epoch_order_small_and_skewed_datasets():
    """A group (or the whole dataset) smaller than the padding it needs: the fill wraps cyclically."""
    roidb = [{"width": 640, "height": 480}] * 40 + [{"width": 480, "height": 640}] * 3
    o = epoch_order(roidb, 16, 0, seed=1)
    assert o.size % 16 == 0 and set(o.tolist()) == set(range(43))
    rows = o.reshape(-1, 16)
    horz = np.array([r["width"] >= r["height"] for r in roidb])
    assert all(len(set(horz[r].tolist())) == 1 for r in rows)              # no batch mixes the two groups
    tiny = [{"width": 640, "height": 480}] * 5
    for grouping in (True, False):
        o = epoch_order(tiny, 16, 0, seed=1, aspect_grouping=grouping)
        assert o.size == 16 and set(o.tolist()) == set(range(5))
    o = epoch_order(tiny, 16, 0, shuffle=False)
    assert o.tolist() == [i % 5 for i in range(16)]
