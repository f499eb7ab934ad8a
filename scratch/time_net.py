import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'yolo-compression-and-deployment-in-fpga_amd'); sys.path.insert(0,'tests')
import numpy as np, torch
from yolo355 import synth
from yolo355.netengine import Net
from oracle import fp32_oracle as F
arch = sys.argv[1]; B = int(sys.argv[2]); classes = int(sys.argv[3])
anchors = synth.TINY_MULTI_ANCHOR_SIZE if arch=="tiny_yolo_v3" else synth.ANCHOR_SIZE_MASK
A = len(anchors)//2 if arch=="tiny_yolo_v3" else len(anchors)
layers = synth.make_fp32_model(arch, 5, classes, A, pred_gain=1.5, obj_bias=-2.0)
net = Net(arch, [416,416], classes, anchors, 0.01, 0.5, max_batch=B, device="cuda:0")
for i,L in enumerate(layers):
    w,b = L["w"].astype(np.float64), L["b"].astype(np.float64)
    if L["bn"] is not None:
        g,be,mu,var=(a.astype(np.float64) for a in L["bn"]); s=g/np.sqrt(var+1e-5); w,b=w*s[:,None,None,None],(b-mu)*s+be
    net.load_layer(i,w.astype(np.float32),b.astype(np.float32))
x = torch.from_numpy(synth.make_images(1000,B,416,416)).cuda()
for _ in range(3): net.forward_device(x)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): net.forward_device(x)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
net.profile(True); net.forward_device(x); ms=net.profile_ms()
print(arch, "B",B,"ms/step %.3f img/s %.0f"%(dt*1e3,B/dt)); print([round(m,3) for m in ms])
