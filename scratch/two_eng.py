import sys
sys.path.insert(0,'.'); sys.path.insert(0,'yolo-compression-and-deployment-in-fpga_amd')
import numpy as np, torch
import bench
from yolo355 import synth, prep
from yolo355.engine import Engine
B=64; dev=torch.device("cuda",0)
streams=[torch.cuda.Stream(device=dev) for _ in range(2)]
engs=[]
for st in streams:
    with torch.cuda.stream(st):
        e=Engine([416,416],2,synth.ANCHOR_SIZE_MASK,0.01,0.5,max_batch=B,device=dev); e.load_quantized(bench.quantized_layers(2))
    engs.append(e)
sa=engs[0].calibrate(synth.make_images(1,1,416,416),[prep.RangeTracker() for _ in range(11)])
for e in engs: e.set_act_exponents(sa)
x=torch.from_numpy(synth.make_images(1000,B,416,416)).to(dev)
bufs=[tuple(torch.empty_like(t) for t in engs[0]._buffers(B)) for _ in range(4)]
torch.cuda.synchronize()
def snap(o): return [t[:B].clone().cpu() for t in o]
o=engs[0].forward_device(x,0,bufs[0]); torch.cuda.synchronize(); ref=snap(o)
o=engs[1].forward_device(x,0,bufs[1]); torch.cuda.synchronize(); r1=snap(o)
n=ref[3]
def same(a,b):
    if not torch.equal(a[3],b[3]): return False
    for i in range(B):
        k=int(a[3][i])
        if not (torch.equal(a[0][i,:k],b[0][i,:k]) and torch.equal(a[1][i,:k],b[1][i,:k]) and torch.equal(a[2][i,:k],b[2][i,:k])): return False
    return True
print("seq", int(ref[3].sum()), int(r1[3].sum()), same(ref,r1))
feats=[engs[1].get_feature(k,B).copy() for k in range(10)]
bad=0
for it in range(4000):
    engs[0].forward_device(x,0,bufs[0]); engs[1].forward_device(x,0,bufs[1]); engs[0].forward_device(x,0,bufs[2])
    torch.cuda.synchronize()
    g=snap(bufs[1])
    if not same(ref,g):
        bad+=1
        msg=[]
        for k in range(10):
            f=engs[1].get_feature(k,B); d=(f!=feats[k])
            if d.any():
                idx=np.argwhere(d)
                msg.append((k,int(d.sum()), idx[0].tolist(), idx[-1].tolist(), sorted(set(idx[:,0].tolist()))[:8]))
        print("it",it,"first diffs", msg[:3])
        f=engs[1].get_feature(9,B); d=(f!=feats[9])
        per_img=d.reshape(B,-1).sum(1); print("  per image", per_img.tolist())
        for b in np.nonzero(per_img)[0][:2]:
            yx=np.argwhere(d[b].any(0)); print("  img",b,"pixels", yx.tolist()[:40])
            y,x_=yx[0]; print("   got",f[b,:8,y,x_].tolist(),"ref",feats[9][b,:8,y,x_].tolist())
            ch=np.nonzero(d[b].any((1,2)))[0]; print("   channels", ch.tolist())
        if bad>=4: break
print("bad",bad)
