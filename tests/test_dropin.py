"""The reference's Python surface reproduced by the drop-in classes."""
import numpy as np
import pytest
import torch

from yolo355 import prep, synth
from yolo355.models.slim_yolo_v2 import SlimYOLOv2_quantize_bnfuse
from yolo355.utils import Conv2d, Conv2d_fuse, Conv2d_fuse_nobias, fuse_conv_and_bn
from helpers import dets_match
from cases import E2E

# state_dict keys of the reference's SlimYOLOv2_quantize_bnfuse (probed with the reference
# imported in the build container)
TRACKERS = ["a_tracker_in", "a_tracker1", "a_tracker2", "a_tracker3_1", "a_tracker3_2", "a_tracker4_1",
            "a_tracker4_2", "a_tracker5", "a_tracker6", "a_tracker7", "a_tracker_pred"]
CONVS = ["conv1", "conv2", "conv3_1", "conv3_2", "conv4_1", "conv4_2", "conv5", "conv6", "conv7"]
REF_KEYS = sorted([t + s for t in TRACKERS for s in (".scale", ".first_a")] +
                  [c + ".convs.0" + s for c in CONVS for s in (".weight", ".bias")] + ["pred.weight", "pred.bias"])


def _model(weights, C, anchors, size, conf, device="cpu", retune=False):
    net = SlimYOLOv2_quantize_bnfuse(device, input_size=size, num_classes=C, trainable=False,
                                     conf_thresh=conf, nms_thresh=0.5, anchor_size=anchors)
    sd = net.state_dict()
    for name, w, b in weights:
        k = "pred" if name == "pred" else name + ".convs.0"
        sd[k + ".weight"] = torch.from_numpy(w.copy())
        sd[k + ".bias"] = torch.from_numpy(b.copy())
    net.load_state_dict(sd, strict=False)          # retune_bias_quantize.py:305
    prep.init_quantize_net(net, 8)
    prep.quantize_layers(8, retune=retune)
    return net.eval()


def test_signatures_and_state_dict_layout():
    net = SlimYOLOv2_quantize_bnfuse("cpu", input_size=[416, 416], num_classes=2, anchor_size=synth.ANCHOR_SIZE_MASK)
    assert sorted(net.state_dict().keys()) == REF_KEYS
    assert net.stride == 16 and net.conf_thresh == 0.01 and net.nms_thresh == 0.5 and net.trainable is False
    assert net.pred.out_channels == 35 and net.conv1.convs[0].padding == (1, 1)     # 4th positional = padding
    net.set_grid([240, 320])
    assert net.input_size == [240, 320] and net.scale.tolist() == [[[320, 240, 320, 240]]]
    for cls in (Conv2d, Conv2d_fuse, Conv2d_fuse_nobias):
        m = cls(3, 16, 3, 1, leakyReLU=True)
        assert isinstance(m.convs, torch.nn.Sequential) and m.convs[0].padding == (1, 1)
        assert isinstance(m.convs[-1], torch.nn.LeakyReLU) and m.convs[-1].negative_slope == 0.125
    assert Conv2d_fuse_nobias(3, 8, 3, 1).convs[0].bias is None
    assert callable(fuse_conv_and_bn)


def test_product_never_falls_back():
    net = SlimYOLOv2_quantize_bnfuse("cpu", input_size=[96, 96], num_classes=2, anchor_size=synth.ANCHOR_SIZE_MASK)
    x = torch.zeros(1, 3, 96, 96)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            net(x)                               # quantization=False runs on the bf16 engine: no GPU, no fallback
    net.trainable = True
    with pytest.raises(NotImplementedError):
        net(x, quantization=True)
    net.trainable = False
    if not torch.cuda.is_available():
        with pytest.raises((RuntimeError, ValueError)):
            net(x, quantization=True)            # raw fp32 weights / no GPU: loud, never a CPU path


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["c1", "find", "diverse"])
def test_model_dropin_matches_reference(golden, tag):
    wkw, anchors, pattern = E2E[tag]
    H, W, C, calib_seed = [int(v) for v in golden[tag + "/meta"][:4]]
    confs = [float(v) for v in golden[tag + "/confs"]]
    net = _model(synth.make_weights(**wkw, num_classes=C), C, anchors, [H, W], confs[0], "cuda:0", retune=(tag == "find"))
    x = torch.from_numpy(synth.make_images(calib_seed, 1, H, W, pattern))
    out = net(x, quantization=True, find=(tag == "find"))      # first call: self-calibration (:25-27)
    assert [isinstance(o, np.ndarray) for o in out] == [True] * 3
    assert out[0].dtype == np.float32 and out[1].dtype == np.float32 and out[2].dtype == np.int64
    assert out[0].flags.writeable                                   # callers scale boxes in place (test.py:88-90)
    sa = [int(torch.floor(torch.log2(getattr(net, t).scale)).item()) for t in TRACKERS]
    assert sa == [int(v) for v in golden[tag + "/sa"]]
    assert all(int(getattr(net, t).first_a.item()) == 1 for t in TRACKERS)
    ref = (golden[tag + "/calib/det0/boxes"], golden[tag + "/calib/det0/scores"], golden[tag + "/calib/det0/cls"])
    ok, msg = dets_match(ref, out, 2e-5, 2e-6)
    assert ok and (tag != "diverse" or msg == "exact"), msg
    # second call: trackers frozen, same answer; thresholds are read at call time like the reference
    out2 = net(x, quantization=True, find=(tag == "find"))
    assert all(np.array_equal(a, b) for a, b in zip(out, out2))
    net.conf_thresh = confs[1]
    out3 = net(x, quantization=True, find=(tag == "find"))
    ref = (golden[tag + "/calib/det1/boxes"], golden[tag + "/calib/det1/scores"], golden[tag + "/calib/det1/cls"])
    ok, msg = dets_match(ref, out3, 2e-5, 2e-6)
    assert ok, msg
    # a saved and re-loaded checkpoint carries the calibration
    net2 = SlimYOLOv2_quantize_bnfuse("cuda:0", input_size=[H, W], num_classes=C, conf_thresh=confs[1],
                                      anchor_size=anchors).eval()
    net2.load_state_dict(net.state_dict())
    out4 = net2(x, quantization=True, find=(tag == "find"))
    assert all(np.array_equal(a, b) for a, b in zip(out3, out4))


@pytest.mark.gpu
def test_model_dropin_guard_and_batch(golden):
    H, W, C, seed, gain = [int(v) for v in golden["guard/meta"]]
    net = _model(synth.make_weights(seed=2, weight_gain=float(gain), num_classes=C), C, synth.ANCHOR_SIZE_MASK,
                 [H, W], 0.01, "cuda:0", retune=True)
    x = torch.from_numpy(synth.make_images(seed, 3, H, W))
    with pytest.raises(AssertionError):
        net(x[:1], quantization=True, find=True)              # "too high!!!" (:222-227)
    net = _model(synth.make_weights(seed=2, num_classes=C), C, synth.ANCHOR_SIZE_MASK, [H, W], 0.01, "cuda:0")
    net(x[:1], quantization=True)
    batch = net.forward_batch(x)
    for i in range(3):
        one = net(x[i:i + 1], quantization=True)
        assert all(np.array_equal(a, b) for a, b in zip(one, batch[i]))


@pytest.mark.gpu
def test_conv2d_fuse_operator_matches_reference_layer(golden):
    """utils.modules.Conv2d_fuse.forward on fake-quantized operands == the reference module's fp32
    output (golden G1 stores its max and its requantised value)."""
    from oracle import yolo_oracle as O
    for n in range(4):
        cin, cout, h, w, sa_in, e_w, e_b, sa_out, leaky, s0, s1, s2 = [int(v) for v in golden["layer/%d/meta" % n]]
        rnd = lambda s, shp: (synth.uniform_u8(s, shp).astype(np.int32) - 128).clip(-127, 127)
        q_in, q_w, q_b = rnd(s0, (2, cin, h, w)), rnd(s1, (cout, cin, 3, 3)), rnd(s2, (cout,))
        m = Conv2d_fuse(cin, cout, 3, 1, leakyReLU=True)
        with torch.no_grad():
            m.convs[0].weight.copy_(torch.from_numpy(q_w.astype(np.float32) / np.float32(2.0 ** e_w)))
            m.convs[0].bias.copy_(torch.from_numpy(q_b.astype(np.float32) / np.float32(2.0 ** e_b)))
        x = torch.from_numpy(q_in.astype(np.float32) / np.float32(2.0 ** sa_in)).cuda()
        y = m(x)
        assert y.is_cuda and y.dtype == torch.float32
        t, Fx, _ = O.conv_layer_int(q_in, q_w, q_b, sa_in, e_w, e_b, True)
        assert np.array_equal(y.cpu().numpy(), t.astype(np.float32) * np.float32(2.0 ** -Fx))
        assert float(y.abs().max()) == float(golden["layer/%d/ymax" % n][0])
        q = torch.round(y.cpu() * (2.0 ** sa_out)).numpy().astype(np.int32)
        assert np.array_equal(q, golden["layer/%d/q_out" % n])
    # operands that are not fake-quantized (utils/modules.py:28-29 accepts any fp32 tensor): the same layer on the bf16 MFMA
    # (y355_conv2d_bf16), within the bf16 tolerance of the fp32 reference -- still on the GPU, no CPU / PyTorch fallback
    xr = (torch.rand(1, cin, 6, 7) * 0.123 - 0.06).cuda()
    yr = m(xr)
    with torch.no_grad():
        want = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xr.cpu(), m.convs[0].weight.cpu(), m.convs[0].bias.cpu(), padding=1), 0.125)
    assert yr.is_cuda and yr.shape == want.shape
    assert float((yr.cpu() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max()) + 0.03
