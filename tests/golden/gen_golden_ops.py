#!/usr/bin/env python3
"""Golden vectors of the operator API of the wider model families, produced by running
the REFERENCE's own modules in the build container: utils.modules.reorg_layer / SPP / Conv2d
(utils/modules.py:6-18, 43-72), backbone.darknet.Conv_BN_LeakyReLU / resblock (backbone/darknet.py:12-38),
all in eval mode on CPU fp32.  Operands come from the build-owned generator (tests/cases.py:ops_inputs),
so only the reference's outputs are stored (float16-compressible data is kept as float32: the files are small).

    python tests/golden/gen_golden_ops.py        # rewrites tests/golden/ops.npz
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (import recipe of the reference)

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
from cases import OPS_CASES, ops_inputs  # noqa: E402


def load_cba(m, w, b, bn):
    """m.convs = Sequential(conv, bn, act)"""
    m.convs[0].weight.data = torch.from_numpy(w.copy())
    m.convs[0].bias.data = torch.from_numpy(b.copy())
    m.convs[1].weight.data = torch.from_numpy(bn[0].copy())
    m.convs[1].bias.data = torch.from_numpy(bn[1].copy())
    m.convs[1].running_mean.data = torch.from_numpy(bn[2].copy())
    m.convs[1].running_var.data = torch.from_numpy(bn[3].copy())


def main():
    ref = G.import_reference()
    dk = importlib.import_module("backbone.darknet")
    out = {}
    with torch.no_grad():
        for tag, kind, prm in OPS_CASES:
            d = ops_inputs(tag, kind, prm)
            x = torch.from_numpy(d["x"])
            if kind == "reorg":
                y = ref.modules.reorg_layer(prm[4])(x)
            elif kind == "spp":
                y = ref.modules.SPP()(x)
            elif kind == "conv":
                mod, B, Cin, Cout, H, W, k, s, leaky = prm
                if mod == "Conv2d":
                    m = ref.modules.Conv2d(Cin, Cout, k, padding=k // 2, stride=s, leakyReLU=leaky)
                else:
                    m = dk.Conv_BN_LeakyReLU(Cin, Cout, k, padding=k // 2, stride=s)
                load_cba(m, d["w"], d["b"], [d["bn_w"], d["bn_b"], d["bn_mean"], d["bn_var"]])
                m.eval()
                y = m(x)
            else:
                B, ch, H, W, nb = prm
                m = dk.resblock(ch, nblocks=nb)
                for i in range(nb):
                    for j in range(2):
                        load_cba(m.module_list[i][j], d["w%d_%d" % (i, j)], d["b%d_%d" % (i, j)], d["bn%d_%d" % (i, j)])
                m.eval()
                y = m(x)
            out[tag] = y.numpy().astype(np.float32)
            print(tag, out[tag].shape, float(np.abs(out[tag]).max()))
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)


if __name__ == "__main__":
    main()
