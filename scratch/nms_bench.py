import sys, os, numpy as np, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"yolo-compression-and-deployment-in-fpga_amd"), os.path.join(ROOT,"tests")]
import torch
from yolo355 import synth
from yolo355.engine import Engine
g = dict(np.load(os.path.join(ROOT,"tests","golden","e2e.npz")))
pq = g["c1/calib/pred_q"]; sa = int(g["c1/sa"][10])
B=64
pqb = np.concatenate([pq]*B)
for conf, thr in [(0.01,0.5),(0.01,0.9),(0.2530,0.5),(0.9,0.5)]:
    eng = Engine([416,416], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=conf, nms_thresh=thr, max_batch=B)
    for it in range(3):
        t0=time.perf_counter(); d = eng.head_nms(pqb, sa); dt=time.perf_counter()-t0
    print("conf",conf,"thr",thr,"ndet",len(d[0][1]), "host ms", dt*1e3)
    eng.close()
