#!/usr/bin/env python3
"""max |f16x3 - f32| of the head outputs on a fixture or synthetic shape (GPU): tools/cmp_modes.py [fixture|C,H,W,B]"""
import os, sys
import torch
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import util
from os2d_amd.utils import synthetic
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["affine_noinv"]:
    if "," in name:
        C, H, W, B = [int(v) for v in name.split(",")]
        fx = dict(P=6, inverse=True, state=synthetic.make_transform_net_state(6, seed=31), fm=synthetic.make_feature_map(C, H, W, seed=41) + 0.05,
                  class_fms=[c + 0.05 for c in synthetic.make_class_feature_maps(B, C, sizes=[(15, 15), (11, 19)], seed=900)])
    else:
        fx = util.load_head_fixture(name)
    creator = util.make_head_creator(fx["P"], fx["inverse"], fx["state"], dev)
    with torch.no_grad():
        head = creator.create_os2d_head([c.to(dev) for c in fx["class_fms"]])
        ref = [t.clone() for t in head(fx["fm"].to(dev), precision="f32")]
        out = head(fx["fm"].to(dev), precision="f16x3")
    d = [float((a - b).abs().max()) for a, b in zip(out, ref)]
    bad = (out[1] - ref[1]).abs() > 1e-5
    print(name, "loc %.3e cls %.3e corners %.3e" % (d[0], d[1], d[3]), "bad cls cells:", int(bad.sum()), "of", bad.numel(),
          "first bad idx:", bad.nonzero()[:6].tolist())
