import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'yolo-compression-and-deployment-in-fpga_amd'); sys.path.insert(0,'tests')
import numpy as np
from cases import FP32_CASES, fp32_setup
from helpers import dets_close, dets_match
from oracle import fp32_oracle as F, yolo_oracle as O
import test_fp32_models as T
for case in FP32_CASES:
    tag, arch, size, classes = case[:4]
    net, layers, anchors, x = T._load_net(case, len(case[5]))
    r = F.detect(arch, layers, x, size, anchors, classes, 0.01, 0.5)
    out = net.forward(x, tap=True)
    B = x.shape[0]
    for k, p in enumerate(r["preds"]):
        got = net.get_tensor(net.num_tensors - len(r["preds"]) + k, B)
        err = got.astype(np.float64) - p
        print(tag, "pred", k, "relL2 %.4f maxerr/max %.4f" % (np.sqrt((err**2).sum()/(p.astype(np.float64)**2).sum()), np.abs(err).max()/np.abs(p).max()))
    nt = len(r["taps"])
    for k in range(nt):
        got = net.get_tensor(k, B); p = r["taps"][k]
        err = got.astype(np.float64) - p
        print("   tap", k, p.shape, "relL2 %.4f" % np.sqrt((err**2).sum()/(p.astype(np.float64)**2).sum()))
    cb, cs, cc = net.candidates(B)
    best = r["cls_scores"].max(axis=2)
    ds = np.abs(cs-best); db = np.abs(cb-r["box"]).max(axis=2)
    print("  score err q99 %.4f max %.4f ; box err q98 %.4f q99.9 %.4f max %.4f" % (np.quantile(ds,.99), ds.max(), np.quantile(db,.98), np.quantile(db,.999), db.max()))
    for bi in range(B):
        # NMS of the engine == reference postprocess on the engine's own per-anchor decode
        prob = np.zeros((cs.shape[1], classes), np.float32); prob[np.arange(cs.shape[1]), cc[bi]] = cs[bi]
        ref2 = O.postprocess(cb[bi], prob, 0.01, 0.5, classes)
        ok, msg = dets_match(ref2[:3], out[bi], box_tol=0, score_tol=0, all_scores=cs[bi])
        print("  img", bi, "nms-on-own-candidates:", ok, msg, "| vs oracle dets:", len(r["dets"][bi][1]), len(out[bi][1]),
              [tuple(round(v,3) for v in dets_close(r["dets"][bi], out[bi], iou, st)) for iou, st in ((0.8,0.05),(0.7,0.1),(0.5,0.2))])
    net.profile(True); net.forward(x); ms = net.profile_ms(); print("  ms", [round(m,3) for m in ms])
    net.close()
