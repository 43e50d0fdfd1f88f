#!/usr/bin/env python3
"""Diagnostic: where do repeated multi-stream pyramid runs differ from the serial run (class, row, column)?"""
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
with torch.no_grad():
    head = creator.create_os2d_head([base[b % 8] for b in range(128)])
    head.precision = sys.argv[1] if len(sys.argv) > 1 else "fft"
    ser = PyramidHeadRunner(head, num_streams=1, device=dev).run(levels, inputs_are_features=True)
    torch.cuda.synchronize()
    ser = [[t.clone() for t in r] for r in ser[:3]]
    for it in range(8):
        par = PyramidHeadRunner(head, device=dev).run(levels, inputs_are_features=True)
        torch.cuda.synchronize()
        for lvl in range(7):
            d = (par[1][lvl] != ser[1][lvl])          # cls [A,B,1,H,W]
            if d.any():
                idx = d.nonzero()
                bs = sorted(set(idx[:, 1].tolist()))
                Wl = LEVELS[lvl][1]
                hs = sorted(set((idx[:, -1] // Wl).tolist())) if d.dim() == 3 else sorted(set(idx[:, 3].tolist()))
                ws = sorted(set((idx[:, -1] % Wl).tolist())) if d.dim() == 3 else sorted(set(idx[:, 4].tolist()))
                print("run", it, "level", lvl, LEVELS[lvl], "differing cls cells:", int(d.sum()), "classes", bs[:20], "rows", hs[:12], "..", hs[-3:], "cols", ws[:8], "..", ws[-3:],
                      "max", float((par[1][lvl] - ser[1][lvl]).abs().max()))
