"""The plain-C oracle (CPU baseline) against the numpy oracle and the golden vectors."""
import subprocess
import os

import numpy as np
import pytest

from oracle import yolo_oracle as O
from yolo355 import synth
from helpers import dets_match
from cases import E2E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def coracle():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    from oracle import c_oracle
    return c_oracle


@pytest.mark.parametrize("tag", ["batch", "gap"])
def test_c_oracle_matches_golden_and_numpy_oracle(golden, coracle, tag):
    wkw, anchors, pattern = E2E[tag]
    meta = [int(v) for v in golden[tag + "/meta"]]
    H, W, C, calib_seed = meta[:4]
    conf = float(golden[tag + "/confs"][0])
    ql = O.quantize_layers(synth.make_weights(**wkw, num_classes=C))
    sa = [int(v) for v in golden[tag + "/sa"]]
    xc = synth.make_images(calib_seed, 1, H, W, pattern)
    pred, nsat, dets = coracle.detect(xc, ql, sa, [H, W], anchors, C, conf, 0.5)
    assert np.array_equal(pred, golden[tag + "/calib/pred_q"])          # bit-exact integer pipeline
    assert nsat.sum() == 0
    tr = [O.RangeTracker() for _ in range(11)]
    r = O.detect(xc, ql, tr, [H, W], anchors, C, conf, 0.5)
    ok, msg = dets_match(r["dets"][0][:3], dets[0], 2e-5, 2e-6, all_scores=r["cls_scores"][0].max(1))
    assert ok, msg
    ref = (golden[tag + "/calib/det0/boxes"], golden[tag + "/calib/det0/scores"], golden[tag + "/calib/det0/cls"])
    ok, msg = dets_match(ref, dets[0], 2e-5, 2e-6, all_scores=r["cls_scores"][0].max(1))
    assert ok, msg
