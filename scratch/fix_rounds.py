import sys
sys.path.insert(0,'.'); sys.path.insert(0,'yolo-compression-and-deployment-in-fpga_amd'); sys.path.insert(0,'tests')
import numpy as np
from oracle import yolo_oracle as O
from yolo355 import synth
import bench
ql = O.quantize_layers(synth.make_weights(2, num_classes=2))
tr = [O.RangeTracker() for _ in range(11)]
O.detect(synth.make_images(1,1,416,416), ql, tr, [416,416], synth.ANCHOR_SIZE_MASK, 2)
x = synth.make_images(1000, 3, 416, 416)
r = O.detect(x, ql, tr, [416,416], synth.ANCHOR_SIZE_MASK, 2, 0.01, 0.5)
for bi in range(3):
    box, sc = r["box"][bi], r["cls_scores"][bi]
    cls = sc.argmax(1); s = sc.max(1); keep = s >= 0.01
    idx = np.where(keep)[0]; b = box[idx]; s = s[idx]; c = cls[idx]
    n = len(idx)
    x1,y1,x2,y2 = b.T
    area = (x2-x1)*(y2-y1)
    iw = np.maximum(1e-28, np.minimum(x2[:,None],x2[None])-np.maximum(x1[:,None],x1[None]))
    ih = np.maximum(1e-28, np.minimum(y2[:,None],y2[None])-np.maximum(y1[:,None],y1[None]))
    inter = (iw*ih).astype(np.float32)
    iou = inter/(area[:,None]+area[None]-inter)
    adj = ~(iou <= 0.5) & (c[:,None]==c[None]); np.fill_diagonal(adj, False)
    order = np.lexsort((idx, -s)); rank = np.empty(n, int); rank[order] = np.arange(n)
    E = np.argwhere(adj & (rank[:,None] < rank[None]))   # a earlier than b
    print("img",bi,"cands",n,"edges",len(E), "per class", [int((c[E[:,0]]==k).sum()) for k in range(2)], "conflicted", int(adj.any(1).sum()))
    state = np.zeros(n, int)  # 0 und,1 kept,2 dead
    state[~adj.any(1)] = 1
    live = np.ones(len(E), bool); rounds=0
    while (state==0).any():
        rounds+=1
        blocked = np.zeros(n,bool)
        sa = state[E[:,0]]
        kill = live & (sa==1); state[E[kill,1]] = 2
        blk = live & (sa==0); blocked[E[blk,1]] = True
        live = blk & True
        und = (state==0) & ~blocked
        state[und] = 1
        print("   round",rounds,"live edges",int(live.sum()),"undecided",int((state==0).sum()))
    # check against greedy
    ref = O.postprocess(box, sc, 0.01, 0.5, 2)
    print("  rounds", rounds, "kept", int((state==1).sum()), "ref", len(ref[1]))
