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
eng = Engine([416,416], 2, synth.ANCHOR_SIZE_MASK, conf_thresh=0.01, nms_thresh=0.5, max_batch=B)
for it in range(3):
    d = eng.head_nms(pqb, sa)
print("ndet",len(d[0][1]))
