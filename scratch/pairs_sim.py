import sys
sys.path.insert(0,'.'); sys.path.insert(0,'yolo-compression-and-deployment-in-fpga_amd'); sys.path.insert(0,'tests')
import numpy as np
from oracle import yolo_oracle as O
from yolo355 import synth
ql = O.quantize_layers(synth.make_weights(2, num_classes=2))
tr = [O.RangeTracker() for _ in range(11)]
O.detect(synth.make_images(1,1,416,416), ql, tr, [416,416], synth.ANCHOR_SIZE_MASK, 2)
x = synth.make_images(1000, 1, 416, 416)
r = O.detect(x, ql, tr, [416,416], synth.ANCHOR_SIZE_MASK, 2, 0.01, 0.5)
box = r["box"][0]; sc = r["cls_scores"][0].max(1); keep = sc >= 0.01
idx = np.where(keep)[0]; b = box[idx]
w = b[:,2]-b[:,0]; h = b[:,3]-b[:,1]; ar = w*h; cx=(b[:,0]+b[:,2])/2; cy=(b[:,1]+b[:,3])/2
anchor = idx % 5
print("cands", len(idx), "w range", w.min(), w.max(), "per-anchor wmax", [float(w[anchor==a].max()) for a in range(5)], "median w", [float(np.median(w[anchor==a])) for a in range(5)])
thr=0.5; kr=(1-thr)*0.5*1.001; thr_lo=thr*0.999
def visits(group, G, Hb, Wb):
    bx = np.clip((cx*Wb).astype(int),0,Wb-1); by = np.clip((cy*Hb).astype(int),0,Hb-1)
    key = group*(Hb*Wb)+by*Wb+bx
    order = np.argsort(key, kind="stable"); 
    pos = np.empty(len(idx),int); pos[order]=np.arange(len(idx))
    # stats
    st = {}
    for g in range(G):
        m = group==g
        if m.any(): st[g]=(w[m].max(),h[m].max(),ar[m].min(),ar[m].max())
    tot=0; mx=0
    # histogram of counts per (group,by,bx)
    cnt = np.zeros((G,Hb,Wb),int); np.add.at(cnt,(group,by,bx),1)
    csum = cnt.cumsum(2)
    for i in range(len(idx)):
        v=0
        for g in range(group[i],G):
            if g not in st: continue
            wm,hm,amin,amax = st[g]
            if ar[i] <= thr_lo*amin or amax <= thr_lo*ar[i]: continue
            rx = kr*(w[i]+wm)+1e-6; ry = kr*(h[i]+hm)+1e-6
            x0=max(0,int(np.floor((cx[i]-rx)*Wb))); x1=min(Wb-1,int(np.floor((cx[i]+rx)*Wb)))
            y0=max(0,int(np.floor((cy[i]-ry)*Hb))); y1=min(Hb-1,int(np.floor((cy[i]+ry)*Hb)))
            if g==group[i]: y0=max(y0,by[i])
            v += cnt[g,y0:y1+1,x0:x1+1].sum()
        tot+=v; mx=max(mx,v)
    return tot/len(idx), mx
print("anchor 26x26", visits(anchor,5,26,26))
ex = np.floor(np.log2(np.maximum(ar,1e-38))).astype(int); oct_ = np.clip(-ex-1,0,15)
print("octave 16x16", visits(oct_,16,16,16))
base = np.array([np.median(ar[anchor==a]) for a in range(5)])
sub = np.clip(np.floor(np.log2(ar/base[anchor])+1.5).astype(int),0,2)
print("anchor x3 16x16", visits(anchor*3+sub,15,16,16))
wo = np.clip(-np.floor(np.log2(np.maximum(w,1e-9))).astype(int),0,3); ho = np.clip(-np.floor(np.log2(np.maximum(h,1e-9))).astype(int),0,3)
print("w-oct x h-oct (4x4) 16x16", visits(wo*4+ho,16,16,16))
