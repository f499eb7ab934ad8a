"""Weight preparation as one command: fp32 SlimYOLOv2 checkpoint -> BN fold ->
power-of-two int8 weights -> tracker calibration -> head-room report -> quantized checkpoint + engine
package.  It chains what the reference spreads over three scripts:

    conv+bn2conv.py:314-326                  fuse_conv_and_bn over the model's Conv2d blocks
    retune_bias_quantize.py:99-119, 357-369  init_quantize_net / quantize_layers, tracker calibration
    retune_bias_quantize_findbest.py:122-141, 356-364   the 2^15 head-room check per layer (scale_retune)

    python -m yolo355.tools.prepare --weights slim_yolo_v2.pth --out out/slim_q --num-classes 2 \\
           --anchors mask --size 416 416 [--calib frames.npy] [--corrected-fold]

writes  out/slim_q.pth   state_dict of SlimYOLOv2_quantize_bnfuse (reference key layout: loads into the
                         reference's class and into the drop-in)
        out/slim_q.npz   engine package: q_w/q_b/e_w/e_b per layer, the 11 activation exponents, meta
                         (the successor of the missing c_embedding/weight.h)
Calibration and the head-room report run on the GPU engine (there is no CPU path)."""
import argparse
import json

import numpy as np
import torch

from .. import prep, synth
from ..models.slim_yolo_v2 import SlimYOLOv2, SlimYOLOv2_quantize_bnfuse, _CONVS

ANCHORS = {"mask": synth.ANCHOR_SIZE_MASK, "voc": synth.ANCHOR_SIZE, "coco": synth.ANCHOR_SIZE_COCO}


def fold_model(fp32_model, corrected=False):
    """conv+bn2conv.py:314-326: [(W, b)] of the ten convs after fuse_conv_and_bn (pred has no BN)."""
    out = []
    for name in _CONVS:
        blk = getattr(fp32_model, name)
        fused = prep.fuse_conv_and_bn(blk.convs[0], blk.convs[1], corrected=corrected)
        out.append((fused.weight.detach().clone(), fused.bias.detach().clone()))
    out.append((fp32_model.pred.weight.detach().clone(), fp32_model.pred.bias.detach().clone()))
    return out


def calib_batches(n_images, calib_batch, calib_images=1000):
    """Number of batches the reference's calibration loop runs over a set of n_images (retune_bias_quantize.py:324,365-367:
    `for iter_i, ... : forward; if batch_size * iter_i > 1000: break`, iter_i 0-based, checked after the batch)."""
    total = -(-n_images // calib_batch)
    first_break = calib_images // calib_batch + 1          # smallest iter_i with calib_batch * iter_i > calib_images
    return min(total, first_break + 1)


def prepare(state_dict, num_classes, anchor_size, input_size, calib, device="cuda:0", corrected_fold=False,
            conf_thresh=0.01, nms_thresh=0.5, calib_batch=0, calib_images=1000):
    """Returns (q_model, package dict, report list).  calib: fp32 NCHW tensor/array (already normalised) or
    uint8 [B,H,W,3] BGR frames (normalised like BaseTransform).  calib_batch = 0: its FIRST image calibrates the
    trackers (first-call rule of an eval-mode model, models/slim_yolo_v2.py:25-27).  calib_batch = N > 0: the reference's
    calibration loop (retune_bias_quantize.py:357-369): batches of N images, first batch sets every scale, every
    further batch moves it by the EMA of :30-31, stop after the batch in front of which more than `calib_images` images
    had been seen (:365-367; `calib_batches`)."""
    fp = SlimYOLOv2(device, input_size=input_size, num_classes=num_classes, anchor_size=anchor_size)
    fp.load_state_dict(state_dict, strict=False)
    fp.eval()
    folded = fold_model(fp, corrected_fold)
    qm = SlimYOLOv2_quantize_bnfuse(device, input_size=input_size, num_classes=num_classes, conf_thresh=conf_thresh,
                                    nms_thresh=nms_thresh, anchor_size=anchor_size)
    with torch.no_grad():
        for name, (w, b) in zip(_CONVS, folded[:-1]):
            conv = getattr(qm, name).convs[0]
            conv.weight.copy_(w)
            conv.bias.copy_(b)
        qm.pred.weight.copy_(folded[-1][0])
        qm.pred.bias.copy_(folded[-1][1])
    prep.init_quantize_net(qm, 8)
    prep.quantize_layers(8)
    qm.eval()
    x = np.asarray(calib)
    if x.dtype == np.uint8:
        x = synth.normalize_frames(x)
    if calib_batch > 0:
        for iter_i, i0 in enumerate(range(0, x.shape[0], calib_batch)):
            xb = torch.from_numpy(np.ascontiguousarray(x[i0:i0 + calib_batch], dtype=np.float32))
            qm.calibrate(xb, freeze=False)
            # retune_bias_quantize.py:324,365-367: `if args.batch_size * iter_i > 1000: break` AFTER the batch, iter_i 0-based:
            # the images seen BEFORE this batch are compared (batch 32 -> 33 batches = 1056 images), see calib_batches()
            if calib_batch * iter_i > calib_images:
                break
    else:
        x = torch.from_numpy(np.ascontiguousarray(x[:1], dtype=np.float32))
        qm(x, quantization=True)                           # first call: calibrates every tracker on the GPU
    eng = qm._engine
    sa = eng.get_act_exponents()
    # head-room report: max |conv output| * 2^retune against 2^15 (retune_bias_quantize_findbest.py:122-141)
    report = []
    mods = [getattr(qm, n).convs[0] for n in _CONVS] + [qm.pred]
    for k, m in enumerate(mods):
        ymax = eng.calib_max[k + 1]                        # max |conv output| the tracker saw (slim_yolo_v2.py:22)
        q_w, e_w = prep.as_dyadic_int8(m.weight)
        q_b, e_b = prep.as_dyadic_int8(m.bias)
        report.append(dict(layer=(_CONVS + ["pred"])[k], e_w=int(e_w), e_b=int(e_b), sa_in=int(sa[k]), sa_out=int(sa[k + 1]),
                           max_abs_output=ymax, retune=int(prep.RETUNE[k]),
                           headroom_bits=float(15 - prep.RETUNE[k] - np.log2(max(ymax, 1e-30))),
                           fits_16bit=bool(ymax * 2.0 ** prep.RETUNE[k] < 2.0 ** 15)))
    package = dict(meta=json.dumps(dict(arch="slim_yolo_v2_q_bf", input_size=list(input_size), num_classes=num_classes,
                                        anchors=[list(map(float, a)) for a in anchor_size], corrected_fold=bool(corrected_fold))),
                   sa=np.asarray(sa, np.int32))
    for k, m in enumerate(mods):
        q_w, e_w = prep.as_dyadic_int8(m.weight)
        q_b, e_b = prep.as_dyadic_int8(m.bias)
        package["q_w%d" % k] = q_w.astype(np.int8)
        package["q_b%d" % k] = q_b.astype(np.int32)
        package["e%d" % k] = np.asarray([e_w, e_b], np.int32)
    return qm, package, report


def load_package(path_or_dict, device="cuda:0", max_batch=1, conf_thresh=0.01, nms_thresh=0.5):
    """Engine ready to run from a package written by this tool."""
    from ..engine import Engine
    pk = np.load(path_or_dict) if isinstance(path_or_dict, str) else path_or_dict
    meta = json.loads(str(pk["meta"]))
    eng = Engine(meta["input_size"], meta["num_classes"], meta["anchors"], conf_thresh, nms_thresh, max_batch=max_batch,
                 device=device)
    for k in range(10):
        e = pk["e%d" % k]
        eng.load_layer(k, pk["q_w%d" % k], pk["q_b%d" % k], int(e[0]), int(e[1]))
    eng.set_act_exponents([int(v) for v in pk["sa"]])
    return eng


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--weights", required=True, help="fp32 SlimYOLOv2 state_dict (.pth)")
    ap.add_argument("--out", required=True, help="output prefix")
    ap.add_argument("--num-classes", type=int, default=2)
    ap.add_argument("--anchors", default="mask", choices=sorted(ANCHORS))
    ap.add_argument("--size", type=int, nargs=2, default=[416, 416], metavar=("H", "W"))
    ap.add_argument("--calib", help=".npy with uint8 [B,H,W,3] BGR frames or fp32 [B,3,H,W]; default: a synthetic frame")
    ap.add_argument("--calib-batch", type=int, default=0,
                    help="N > 0: the reference's calibration loop (retune_bias_quantize.py:357-369) over the --calib file in "
                         "batches of N (EMA trackers); 0: first image only (the eval-mode first-call rule)")
    ap.add_argument("--calib-images", type=int, default=1000, help="stop the loop once more images than this were seen (:365)")
    ap.add_argument("--corrected-fold", action="store_true", help="exact BN fold instead of utils/bn_fuse.py's formula")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    sd = torch.load(args.weights, map_location="cpu")
    calib = np.load(args.calib) if args.calib else synth.make_frames_u8(1, 1, args.size[0], args.size[1], "blocks")
    qm, package, report = prepare(sd, args.num_classes, ANCHORS[args.anchors], args.size, calib, args.device, args.corrected_fold,
                                  calib_batch=args.calib_batch, calib_images=args.calib_images)
    torch.save(qm.state_dict(), args.out + ".pth")
    np.savez_compressed(args.out + ".npz", **package)
    for r in report:
        print("%-8s e_w %3d e_b %3d sa %3d -> %3d  max|y| %10.4f  retune %2d  head-room %6.2f bits  %s" % (
            r["layer"], r["e_w"], r["e_b"], r["sa_in"], r["sa_out"], r["max_abs_output"], r["retune"], r["headroom_bits"],
            "ok" if r["fits_16bit"] else "TOO HIGH"))
    print("wrote", args.out + ".pth", args.out + ".npz")


if __name__ == "__main__":
    main()
