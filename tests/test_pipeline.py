"""y355_pipeline (the throughput regime behind the API) and y355_calibrate (the tracker state machine in the C ABI).

CPU part: the tracker arithmetic of the library (y355_tracker_step, no GPU) against torch's -- prep.RangeTracker, the restatement of
AveragedRangeTracker.quantize_activation (models/slim_yolo_v2.py:16-38) that tests/golden pins to the reference.
GPU part: every ticket of a pipeline equals a stand-alone engine's forward bit for bit ("256 images through the
pipeline equal four stand-alone forwards"), ticket life time, caller-owned and pipeline-owned outputs, the uint8 route, the GPU
rescale, and y355_calibrate against the layer-by-layer stepping it replaces.
"""
import ctypes as C

import numpy as np
import pytest

from yolo355 import _ffi, prep, synth


# ----------------------------------------------------------------------------------------------- CPU: tracker arithmetic
def _step(scale, first, m, freeze, momentum=0.1):
    s, f, e = C.c_float(scale), C.c_int32(first), C.c_int32()
    _ffi.check(_ffi.lib().y355_tracker_step(C.byref(s), C.byref(f), float(np.float32(m)), 1 if freeze else 0, momentum, C.byref(e)))
    return s.value, f.value, e.value


def test_tracker_step_matches_torch_first_call_and_frozen():
    rng = np.random.default_rng(0)
    # maxima over ten decades, plus values whose 127 / max sits next to a power of two (where floor(log2) flips)
    vals = list(np.exp(rng.uniform(np.log(1e-4), np.log(1e5), 3000)).astype(np.float32))
    for k in range(-10, 12):
        for d in (-3, -2, -1, 0, 1, 2, 3):
            vals.append(np.nextafter(np.float32(127.0 / 2.0 ** k), np.float32(np.inf if d > 0 else -np.inf)) if d else np.float32(127.0 / 2.0 ** k))
            vals.append(np.float32(127.0 / 2.0 ** k) * np.float32(1 + d * 2.0 ** -22))
    for m in vals:
        t = prep.RangeTracker()
        want = t.update(m, True)                          # first call ever: calibrates even when frozen (:25-27)
        s, f, e = _step(0.0, 0, m, True)
        assert (e, f) == (want, 1), (m, e, want)
        assert np.float32(s).tobytes() == t.scale.numpy().astype(np.float32).tobytes(), (m, s, t.scale)
        want2 = t.update(np.float32(m) * np.float32(3.0), True)      # frozen afterwards (:28-29): nothing moves
        s2, f2, e2 = _step(s, f, np.float32(m) * np.float32(3.0), True)
        assert (s2, f2, e2) == (s, 1, want2) and want2 == want


def test_tracker_step_matches_torch_ema_sequences():
    rng = np.random.default_rng(1)
    for seq in range(200):
        t = prep.RangeTracker()
        s, f = 0.0, 0
        for step in range(12):
            m = np.float32(np.exp(rng.uniform(np.log(0.05), np.log(300.0))))
            want = t.update(m, False)                     # training mode: EMA after the first call (:30-31)
            s, f, e = _step(s, f, m, False)
            assert e == want, (seq, step, m, e, want)
            assert np.float32(s).tobytes() == t.scale.numpy().astype(np.float32).tobytes(), (seq, step)


def test_tracker_step_rejects_an_all_zero_activation():
    s, f, e = C.c_float(0.0), C.c_int32(0), C.c_int32()
    rc = _ffi.lib().y355_tracker_step(C.byref(s), C.byref(f), 0.0, 1, 0.1, C.byref(e))
    assert rc == -4 and b"not a positive finite" in _ffi.lib().y355_last_error()       # Y355_ERANGE; torch's int(floor(log2(inf))) raises too


# ----------------------------------------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


def _weights():
    from oracle import yolo_oracle as O          # checker-side weight recipe, as the other parity tests
    return O.quantize_layers(synth.make_weights(seed=2, num_classes=2))


@gpu
def test_pipeline_256_images_equal_four_stand_alone_forwards():
    """256 images through Pipeline.forward (four chunks of 64 in flight on four handles) == four Engine.forward calls."""
    import torch
    from yolo355.engine import Engine, Pipeline
    H = W = 416
    ql = _weights()
    dev = torch.device("cuda", 0)
    eng = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=64, device=dev)
    eng.load_quantized(ql)
    xc = synth.make_images(1, 1, H, W)
    sa = eng.calibrate(xc, [prep.RangeTracker() for _ in range(11)])
    pipe = Pipeline([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=64, device=dev)
    assert (pipe.handles, pipe.depth) == (_ffi.PIPE_DEFAULT_HANDLES, 2 * _ffi.PIPE_DEFAULT_HANDLES) == (4, 8)
    pipe.load_quantized(ql)
    assert pipe.calibrate(xc, [prep.RangeTracker() for _ in range(11)]) == sa
    x = np.concatenate([synth.make_images(1000 + i, 64, H, W) for i in range(4)])
    got = pipe.forward(x)
    assert len(got) == 256
    for c in range(4):
        want = eng.forward(x[64 * c:64 * (c + 1)])
        for i in range(64):
            for a, b in zip(want[i], got[64 * c + i]):
                assert np.array_equal(a, b), (c, i)
    # a ragged tail and more chunks than tickets in flight: 7 chunks of <= 40 through a depth of 4
    pipe2 = Pipeline([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=40, device=dev, handles=2)
    pipe2.load_quantized(ql)
    pipe2.set_act_exponents(sa)
    got2 = pipe2.forward(x[:250])
    for i in range(250):
        for a, b in zip(got[i], got2[i]):
            assert np.array_equal(a, b), i
    for o in (pipe, pipe2, eng):
        o.close()


@gpu
def test_pipeline_tickets_outputs_and_streams():
    """submit / wait / outputs with caller-owned and pipeline-owned buffers, a ticket's life time, the uint8 route and the GPU
    rescale, all against a stand-alone engine."""
    import torch
    from yolo355.engine import Engine, Pipeline
    H, W, B = 224, 320, 5
    ql = _weights()
    dev = torch.device("cuda", 0)
    eng = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.3, max_batch=B, device=dev)
    eng.load_quantized(ql)
    frames = synth.make_frames_u8(5, 8 * B, H, W, "blocks")
    xs = synth.normalize_frames(frames)
    sa = eng.calibrate(xs[:1], [prep.RangeTracker() for _ in range(11)])
    pipe = Pipeline([H, W], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.3, max_batch=B, device=dev, handles=2)
    pipe.load_quantized(ql)
    pipe.set_act_exponents(sa)
    assert (pipe.handles, pipe.depth) == (2, 4)
    xd = torch.from_numpy(xs).to(dev)
    want = [eng.forward(xs[B * k:B * (k + 1)]) for k in range(8)]

    def check(k, out):
        n = out[3][:B].cpu().numpy()
        for i in range(B):
            wb, ws, wc = want[k][i]
            assert n[i] == len(ws), (k, i)
            assert np.array_equal(out[0][i, :n[i]].cpu().numpy(), wb) and np.array_equal(out[1][i, :n[i]].cpu().numpy(), ws)
            assert np.array_equal(out[2][i, :n[i]].cpu().numpy().astype(np.int64), wc)
    # the input is produced on a side stream right before the submit: ordered=True puts the forward behind it
    side = torch.cuda.Stream(device=dev)
    tickets = []
    with torch.cuda.stream(side):
        for k in range(4):
            xk = xd[B * k:B * (k + 1)] * 1.0              # produced on `side`
            tickets.append(pipe.submit(xk))
    assert tickets == [0, 1, 2, 3]
    for k, t in enumerate(tickets):
        pipe.wait(t)                                      # the current (default) stream waits; no host block
        check(k, pipe.outputs(t))
    # caller-owned buffers, host wait
    mine = tuple(torch.zeros_like(t) for t in pipe.outputs(3))
    t4 = pipe.submit(xd[4 * B:5 * B], out=mine)
    pipe.wait(t4, host=True)
    check(4, mine)
    # ticket 0's slot has been reused by ticket 4
    with pytest.raises(_ffi.Y355Error) as ei:
        pipe.fetch(0)
    assert ei.value.code == _ffi.ENOTREADY
    rc = pipe._lib.y355_pipeline_wait(pipe._h, 0, 0, None)
    assert rc == _ffi.ENOTREADY and b"is gone" in pipe._lib.y355_last_error()
    assert pipe._lib.y355_pipeline_wait(pipe._h, 99, 0, None) == _ffi.EINVAL
    # pipeline-owned device buffers of the C ABI (all four output pointers NULL)
    t5 = C.c_longlong()
    _ffi.check(pipe._lib.y355_pipeline_submit(pipe._h, xd[5 * B:6 * B].contiguous().data_ptr(), B, 0, None, None, None, None, None,
                                              C.byref(t5)))
    ptrs = [C.c_void_p() for _ in range(4)]
    nb = C.c_int()
    _ffi.check(pipe._lib.y355_pipeline_outputs(pipe._h, t5.value, *[C.byref(p) for p in ptrs], C.byref(nb)))
    assert nb.value == B and all(p.value for p in ptrs)
    md = pipe.max_det
    b = np.empty((B, md, 4), np.float32); s = np.empty((B, md), np.float32); c = np.empty((B, md), np.int32); n = np.empty((B,), np.int32)
    _ffi.check(pipe._lib.y355_pipeline_fetch(pipe._h, t5.value, b.ctypes.data, s.ctypes.data, c.ctypes.data, n.ctypes.data))
    for i in range(B):
        assert n[i] == len(want[5][i][1]) and np.array_equal(b[i, :n[i]], want[5][i][0]) and np.array_equal(s[i, :n[i]], want[5][i][1])
    pipe._next = t5.value + 1                             # the raw call above bypassed the Python ticket counter
    # uint8 frames (BaseTransform fused into the first layer) and the evaluators' rescale on the GPU
    sizes = np.array([[640 + 7 * i, 480 + 3 * i] for i in range(2 * B)], np.float32)
    got = pipe.forward(frames[6 * B:8 * B], sizes_wh=sizes, frames=True)
    ref = eng.forward_scaled(xs[6 * B:7 * B], sizes[:B]) + eng.forward_scaled(xs[7 * B:8 * B], sizes[B:])
    for i in range(2 * B):
        for u, v in zip(ref[i], got[i]):
            assert np.array_equal(u, v), i
    # find=True: the 2^15 head-room guard through the pipeline (clean here)
    assert len(pipe.forward(xs[:2 * B], find=True)) == 2 * B
    pipe.close()
    eng.close()


@gpu
def test_calibrate_entry_point_equals_the_stepping_calls():
    """y355_calibrate == the layer-by-layer stepping (y355_input_absmax / y355_run_layer / y355_layer_stats_get) with the tracker
    update done by torch (prep.RangeTracker), first call, frozen call and three EMA steps; tracker buffers bit for bit."""
    import torch
    from yolo355.engine import Engine
    H, W, B = 96, 160, 3
    ql = _weights()
    dev = torch.device("cuda", 0)
    ea = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=B, device=dev)
    eb = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=B, device=dev)
    for e in (ea, eb):
        e.load_quantized(ql)
    lib = _ffi.lib()

    def stepping(eng, x, trackers, freeze):
        xd = eng._dev_input(x)
        m = C.c_float()
        _ffi.check(lib.y355_input_absmax(eng._h, xd.data_ptr(), B, C.byref(m)))
        sa = [trackers[0].update(m.value, freeze)]
        _ffi.check(lib.y355_set_act_exponent(eng._h, 0, sa[0]))
        st = _ffi.LayerStats()
        for k in range(10):
            xp = xd.data_ptr() if k == 0 else None
            _ffi.check(lib.y355_run_layer(eng._h, k, B, 1, xp))
            _ffi.check(lib.y355_layer_stats_get(eng._h, k, C.byref(st)))
            ymax = np.float32(st.absmax_t) * np.float32(2.0 ** (-st.frac_bits))
            sa.append(trackers[k + 1].update(ymax, freeze))
            _ffi.check(lib.y355_set_act_exponent(eng._h, k + 1, sa[-1]))
            _ffi.check(lib.y355_run_layer(eng._h, k, B, 0, xp))
        return sa
    ta = [prep.RangeTracker() for _ in range(11)]
    tb = [prep.RangeTracker() for _ in range(11)]
    for step, freeze in enumerate([True, True, False, False, False]):
        x = synth.make_images(40 + step, B, H, W, "blocks") * np.float32(1.0 + 0.4 * step)
        sa_a = ea.calibrate(x, ta, freeze=freeze)         # one call of the C ABI
        sa_b = stepping(eb, x, tb, freeze)
        assert sa_a == sa_b, (step, sa_a, sa_b)
        for i in range(11):
            assert ta[i].first_a == tb[i].first_a == 1
            assert ta[i].scale.numpy().tobytes() == tb[i].scale.numpy().astype(np.float32).tobytes(), (step, i)
        assert np.array_equal(ea.get_feature(9, B), eb.get_feature(9, B))
        assert ea.get_act_exponents() == sa_a
    ea.close()
    eb.close()
