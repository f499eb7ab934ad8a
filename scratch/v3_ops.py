# per-op device time of the yolo_v3 graph (y355_net profile)
import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"yolo-compression-and-deployment-in-fpga_amd")]
from yolo355 import synth
from yolo355.models.yolo_v3 import myYOLOv3
torch.manual_seed(0)
B=32
m=myYOLOv3("cuda:0",[416,416],20,False,0.1,0.5,synth.MULTI_ANCHOR_SIZE).eval()
net=m._get_net(B)
x=torch.from_numpy(synth.make_images(1000,B,416,416)).cuda()
for _ in range(3): net.forward_device(x)
net.profile(True)
acc=None
for _ in range(5):
    net.forward_device(x); ms=np.array(net.profile_ms()); acc=ms if acc is None else acc+ms
ms=acc/5
print("total ms", ms.sum())
order=np.argsort(-ms)
for i in order[:25]: print(i, round(float(ms[i]),4))
print("first 12 ops:", [round(float(v),3) for v in ms[:12]])
