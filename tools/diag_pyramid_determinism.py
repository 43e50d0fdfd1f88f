#!/usr/bin/env python3
"""Diagnostic: per-level differences between repeated serial / multi-stream pyramid runs in several arithmetic modes."""
import os, sys
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import util
from os2d_amd.engine.pyramid import PyramidHeadRunner
from os2d_amd.utils import synthetic
dev = torch.device("cuda:0")
LEVELS = [(30, 40), (38, 50), (48, 64), (60, 80), (72, 96), (84, 112), (96, 128)]
state = synthetic.make_transform_net_state(6, seed=1)
levels = [synthetic.make_feature_map(1024, h, w, seed=100 + i).to(dev) for i, (h, w) in enumerate(LEVELS)]
base = [c.to(dev) for c in synthetic.make_class_feature_maps(8, 1024, seed=7000)]
creator = util.make_head_creator(6, True, state, dev)
modes = sys.argv[1:] or ["fftx3", "fft", "f16x3"]
with torch.no_grad():
    head = creator.create_os2d_head([base[b % 8] for b in range(128)])
    ref = None
    for mode in modes:
        head.precision = mode
        runs = []
        for name, ns in (("serial", 1), ("serial", 1)) + (("par", None),) * 6:
            r = PyramidHeadRunner(head, num_streams=ns, device=dev).run(levels, inputs_are_features=True) if ns else \
                PyramidHeadRunner(head, device=dev).run(levels, inputs_are_features=True)
            torch.cuda.synchronize()
            runs.append([t.clone() for t in r[0]])
        for i, nm in enumerate(("serial2", "par1", "par2", "par3", "par4", "par5", "par6")):
            d = [float((a - b).abs().max()) for a, b in zip(runs[0], runs[i + 1])]
            print(mode, nm, "vs serial1 loc max diff per level:", ["%.2e" % x for x in d])
        if ref is not None:
            print(mode, "serial1 vs", modes[0], ["%.2e" % float((a - b).abs().max()) for a, b in zip(runs[0], ref)])
        else:
            ref = runs[0]
