# GPU box host: the C oracle's rate at batch 64 for several OpenMP thread counts (the bench line's cpu_baseline uses all of them)
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")]
from oracle import yolo_oracle as O, c_oracle
from yolo355 import synth
omp = ctypes.CDLL("libgomp.so.1")
ql = O.quantize_layers(synth.make_weights(2, num_classes=2))
tr = [O.RangeTracker() for _ in range(11)]
sa = O.detect(synth.make_images(1, 1, 416, 416), ql, tr, [416, 416], synth.ANCHOR_SIZE_MASK, 2)["sa"]
x = synth.make_images(1000, 64, 416, 416)
c_oracle.detect(x[:1], ql, sa, [416, 416], synth.ANCHOR_SIZE_MASK, 2)
for t in [int(v) for v in (sys.argv[1:] or [1, 16, 64, 128, os.cpu_count()])]:
    omp.omp_set_num_threads(t)
    n = 4 if t == 1 else 64
    t0 = time.perf_counter()
    c_oracle.detect(x[:n], ql, sa, [416, 416], synth.ANCHOR_SIZE_MASK, 2, 0.01, 0.5)
    dt = time.perf_counter() - t0
    print("threads %4d: %d images in %6.2f s = %7.2f images/s" % (t, n, dt, n / dt), flush=True)
