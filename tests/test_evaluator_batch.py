"""Evaluator-side batching: the batched helpers build the same all_boxes / COCO dict as the
reference's per-image loops (utils/vocapi_evaluator_mask.py:57-82, utils/cocoapi_evaluator.py:66-98) run on the same
drop-in network one image at a time with the rescale on the host."""
import numpy as np
import pytest

from cases import E2E


class _VocSet:
    """dataset.pull_item(i) -> (im tensor [3,H,W], gt, h, w) like data/voc0712_mask.py"""

    def __init__(self, x, sizes):
        self.x, self.sizes = x, sizes

    def __len__(self):
        return len(self.x)

    def pull_item(self, i):
        import torch
        return torch.from_numpy(self.x[i]), None, self.sizes[i][1], self.sizes[i][0]


class _CocoSet:
    class_ids = [11, 22, 33, 44]

    def __init__(self, imgs):
        self.imgs = imgs

    def __len__(self):
        return len(self.imgs)

    def pull_image(self, i):
        return self.imgs[i], 1000 + i


@pytest.mark.gpu
def test_voc_and_coco_batched_equal_the_per_image_loops():
    import torch
    from yolo355 import synth
    from test_dropin import _model           # the q_bf drop-in loaded with quantized synthetic weights
    wkw, anchors, pattern = E2E["c1"]
    net = _model(synth.make_weights(**wkw, num_classes=2), 2, anchors, [416, 416], 0.1, "cuda:0")
    x0 = synth.make_images(1, 1, 416, 416, pattern)
    n = 7
    x = np.concatenate([synth.make_images(900 + i, 1, 416, 416) for i in range(n)])
    sizes = [(640 + 13 * i, 480 - 7 * i) for i in range(n)]          # original (w, h) of every image
    net(torch.from_numpy(x0), quantization=True)                     # first call freezes the trackers
    # reference loop, per image, host rescale
    want = [[[] for _ in range(n)] for _ in range(2)]
    for i in range(n):
        b, s, c = net(torch.from_numpy(x[i:i + 1]), quantization=True)
        b *= np.array([[sizes[i][0], sizes[i][1], sizes[i][0], sizes[i][1]]])
        for j in range(2):
            inds = np.where(c == j)[0]
            want[j][i] = np.hstack((b[inds], s[inds][:, None])).astype(np.float32) if len(inds) else np.empty([0, 5], np.float32)
    from yolo355.utils.evaluator_batch import voc_all_boxes, coco_data_dict
    got = voc_all_boxes(net, _VocSet(x, sizes), 2, batch_size=3, quantization=True)
    for j in range(2):
        for i in range(n):
            assert got[j][i].dtype == np.float32 and got[j][i].shape == want[j][i].shape
            assert np.array_equal(got[j][i], want[j][i]), (j, i)
    # COCO: images are HWC BGR at their own size; a transform brings them to the network size (the caller's cv2 resize)
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 255, (sizes[i][1], sizes[i][0], 3)).astype(np.uint8) for i in range(3)]

    def transform(img):
        chw = x[len(img) % n]                                        # any deterministic network-size image
        return [np.ascontiguousarray(chw.transpose(1, 2, 0)[:, :, (2, 1, 0)])]
    ids, dd = coco_data_dict(net, _CocoSet(imgs), transform, batch_size=2, quantization=True)
    assert ids == [1000, 1001, 1002]
    k = 0
    for i in range(3):
        b, s, c = net(torch.from_numpy(x[len(imgs[i]) % n][None]), quantization=True)
        b *= np.array([[imgs[i].shape[1], imgs[i].shape[0], imgs[i].shape[1], imgs[i].shape[0]]])
        for t in range(len(s)):
            d = dd[k]
            k += 1
            assert d["image_id"] == 1000 + i and d["category_id"] == _CocoSet.class_ids[int(c[t])]
            assert d["score"] == float(s[t])
            assert d["bbox"] == [float(b[t][0]), float(b[t][1]), float(b[t][2]) - float(b[t][0]), float(b[t][3]) - float(b[t][1])]
    assert k == len(dd)


EVAL = dict(weights=dict(seed=2, pred_gain=400.0, obj_bias=-4.0), size=[240, 320], classes=20, seeds=[51, 52, 53, 54, 59],
            sizes=[(640, 480), (500, 375), (333, 500), (1280, 720), (320, 240)], conf=0.1)      # mirrors gen_golden_r2.py


@pytest.mark.gpu
def test_voc_batched_equals_the_reference_evaluator_loop():
    """all_boxes of the REFERENCE's evaluator loop (utils/vocapi_evaluator_mask.py:49-82 driven on a stub dataset with the
    reference's own q_bf model, golden r2.npz `eval/*`) against voc_all_boxes on the drop-in: a fresh, un-calibrated model
    (the loop's first image calibrates the trackers: the batched helper must do the same, not calibrate on the batch
    maximum), batches of 2 over 5 images, rescale on the GPU."""
    import os
    import torch
    from yolo355 import synth
    from yolo355.utils.evaluator_batch import voc_all_boxes
    from test_dropin import _model
    from helpers import dets_match
    r2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "r2.npz"))
    cfg = EVAL
    size, classes = cfg["size"], cfg["classes"]
    x = np.concatenate([synth.make_images(s, 1, size[0], size[1], "blocks") for s in cfg["seeds"]])
    net = _model(synth.make_weights(**cfg["weights"], num_classes=classes), classes, synth.ANCHOR_SIZE, size, cfg["conf"], "cuda:0")
    assert int(net.a_tracker_in.first_a.item()) == 0
    got = voc_all_boxes(net, _VocSet(x, cfg["sizes"]), classes, batch_size=2, quantization=True)
    assert int(net.a_tracker_in.first_a.item()) == 1
    # per-anchor scores (for the tie-tolerant comparison: a tie partner may be a suppressed candidate) from the checker
    from oracle import yolo_oracle as O
    ql = O.quantize_layers(synth.make_weights(**cfg["weights"], num_classes=classes))
    otr = [O.RangeTracker() for _ in range(11)]
    O.detect(x[:1], ql, otr, size, synth.ANCHOR_SIZE, classes, cfg["conf"], 0.5)
    osc = O.detect(x, ql, otr, size, synth.ANCHOR_SIZE, classes, cfg["conf"], 0.5, saturate=True)["cls_scores"]
    exact = 0
    for i in range(len(x)):
        ref = np.concatenate([np.c_[r2["eval/boxes/%d/%d" % (j, i)], np.full(len(r2["eval/boxes/%d/%d" % (j, i)]), j)] for j in range(classes)])
        mine = np.concatenate([np.c_[got[j][i].reshape(-1, 5), np.full(len(got[j][i]), j)] for j in range(classes)])
        w, h = cfg["sizes"][i]
        sc = np.array([w, h, w, h], np.float32)
        ok, msg = dets_match((ref[:, :4] / sc, ref[:, 4], ref[:, 5].astype(np.int64)),
                             (mine[:, :4] / sc, mine[:, 4], mine[:, 5].astype(np.int64)), 2e-5, 2e-6,
                             all_scores=osc[i].max(axis=1))
        assert ok, (i, msg)
        exact += msg in ("exact", "empty")
        if msg == "exact":                                   # same detections: the rescaled pixel boxes agree to fp32 rounding
            for j in range(classes):
                assert np.allclose(got[j][i], r2["eval/boxes/%d/%d" % (j, i)], rtol=1e-6, atol=1e-3)
    assert exact >= 3
