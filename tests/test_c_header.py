"""include/yolo355.h is the product boundary: it must be valid plain C and a C program must link against libyolo355.so.
Builds tests/c_client/client.c with gcc (C99, warnings as errors) and runs it -- argument checks only, no GPU needed."""
import os
import subprocess

import pytest

from yolo355 import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    exe = str(tmp_path / "client")
    libdir = os.path.dirname(_ffi.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_client", "client.c"), "-o", exe, "-L", libdir, "-l:libyolo355.so",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.startswith("ok version")


def test_header_compiles_as_cxx_too(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "yolo355.h"\nint main() { return y355_version() < 0; }\n')
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _gcc(src, exe):
    libdir = os.path.dirname(_ffi.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_client", src), "-o", exe, "-L", libdir, "-l:libyolo355.so",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_forward_client_builds(tmp_path):
    _gcc("forward.c", str(tmp_path / "forward"))


@pytest.mark.gpu
def test_hot_path_driven_from_a_c_program_equals_the_oracle(tmp_path):
    """tests/c_client/forward.c: create / load_layer x 10 / set_act_exponents / forward_host from plain C; detections against
    the oracle (boxes 2e-5, scores 2e-6: the fp32 head's tolerance of tests/test_gpu_parity.py; identical lists), counters 0"""
    import numpy as np
    from oracle import yolo_oracle as O      # checker only
    from yolo355 import synth
    from helpers import dets_match
    H = W = 416
    C, B, conf, nms = 2, 3, 0.1, 0.5
    anchors = synth.ANCHOR_SIZE_MASK
    ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=C))
    xc = synth.make_images(1, 1, H, W)
    otr = [O.RangeTracker() for _ in range(11)]
    O.detect(xc, ql, otr, [H, W], anchors, C, conf, nms)                       # first call: calibrates the trackers
    sa = [int(t.exponent()) for t in otr]
    xs = np.concatenate([synth.make_images(s, 1, H, W) for s in (5, 6, 7)])
    ref = O.detect(xs, ql, otr, [H, W], anchors, C, conf, nms, saturate=True, keep=True)
    blob = tmp_path / "model.bin"
    with open(blob, "wb") as f:
        np.array([H, W, C, len(anchors), B], np.int32).tofile(f)
        np.array([conf, nms], np.float32).tofile(f)
        np.array(anchors, np.float32).ravel().tofile(f)
        np.array(sa, np.int32).tofile(f)
        for L in ql:
            qw = np.ascontiguousarray(L["q_w"], np.int8)
            np.array([qw.shape[0], qw.shape[1], L["e_w"], L["e_b"]], np.int32).tofile(f)
            qw.tofile(f)
            np.ascontiguousarray(L["q_b"], np.int32).tofile(f)
        np.ascontiguousarray(xs, np.float32).tofile(f)
    exe, out = str(tmp_path / "forward"), str(tmp_path / "out.bin")
    _gcc("forward.c", exe)
    r = subprocess.run([exe, str(blob), out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    raw = open(out, "rb").read()
    md = int(np.frombuffer(raw, np.int32, 1)[0])
    off = 4
    count = np.frombuffer(raw, np.int32, B, off); off += 4 * B
    boxes = np.frombuffer(raw, np.float32, B * md * 4, off).reshape(B, md, 4); off += 16 * B * md
    scores = np.frombuffer(raw, np.float32, B * md, off).reshape(B, md); off += 4 * B * md
    cls = np.frombuffer(raw, np.int32, B * md, off).reshape(B, md); off += 4 * B * md
    sat, guard = np.frombuffer(raw, np.int64, 2, off)
    assert guard == 0 and int(sat) == sum(ref["sat_out"]) + ref["sat"][0], (sat, guard, ref["sat_out"], ref["sat"][0])
    for i in range(B):
        ob, os_, oc, _ = O.postprocess(ref["box"][i], ref["cls_scores"][i], conf, nms, C)
        n = int(count[i])
        ok, msg = dets_match((ob, os_, oc), (boxes[i, :n], scores[i, :n], cls[i, :n].astype(np.int64)), 2e-5, 2e-6,
                             all_scores=ref["cls_scores"][i].max(1))
        assert ok and (msg in ("exact", "empty") or msg.startswith("same boxes")), (i, msg)
