#!/usr/bin/env python3
"""Diagnostic: multi-stream pyramid with optional synchronisation after every level (mode sync) or identical levels (mode same)."""
import os, sys
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import util
from os2d_amd.engine.pyramid import level_stream
from os2d_amd.utils import synthetic
dev = torch.device("cuda:0")
what = sys.argv[1]
LEVELS = [(30, 40), (38, 50), (48, 64), (60, 80), (72, 96), (84, 112), (96, 128)]
if what == "same":
    LEVELS = [(48, 64)] * 7
state = synthetic.make_transform_net_state(6, seed=1)
levels = [synthetic.make_feature_map(1024, h, w, seed=100 + i).to(dev) for i, (h, w) in enumerate(LEVELS)]
base = [c.to(dev) for c in synthetic.make_class_feature_maps(8, 1024, seed=7000)]
creator = util.make_head_creator(6, True, state, dev)
with torch.no_grad():
    head = creator.create_os2d_head([base[b % 8] for b in range(128)])
    head.precision = sys.argv[2] if len(sys.argv) > 2 else "fft"
    ser = [head(l)[1].clone() for l in levels]
    torch.cuda.synchronize()
    bad = 0
    for it in range(8):
        outs = []
        for i, l in enumerate(levels):
            st = level_stream(dev, i)
            with torch.cuda.stream(st):
                outs.append(head(l)[1])
            if what == "sync":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        for i in range(7):
            d = outs[i] != ser[i]
            if d.any():
                bad += 1
                print(what, "run", it, "level", i, LEVELS[i], "cells", int(d.sum()), "classes", sorted(set(d.nonzero()[:, 1].tolist()))[:12])
    print(what, "bad level-runs:", bad, "of 56")
