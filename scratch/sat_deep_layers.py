import sys, os, numpy as np
sys.path[:0]=[os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'yolo-compression-and-deployment-in-fpga_amd')]
from yolo355 import synth
from yolo355.engine import Engine
from yolo355.prep import RangeTracker
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'oracle'))
import yolo_oracle as O
H=W=416; B=2
ql = O.quantize_layers(synth.make_weights(seed=2, num_classes=2, pred_gain=400.0, obj_bias=-4.0))
eng = Engine([H, W], 2, synth.ANCHOR_SIZE_MASK, max_batch=B)
eng.load_quantized(ql)
frames = synth.make_frames_u8(11, B, H, W, "blocks")
xc = synth.normalize_frames(frames)[:1]
eng.calibrate(xc, [RangeTracker() for _ in range(11)])
x = synth.normalize_frames(frames) * np.float32(3.0)
eng.forward(x)
print("sat per layer", [eng.layer_stats(k)["saturated"] for k in range(10)])
