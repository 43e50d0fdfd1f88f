#!/usr/bin/env python3
"""GPU: compare the split-half heads (f16x3, f16x2) with the fp32 head and the reference golden outputs on every
fixture, then at the bench size (64 classes, 1024 x 60 x 80) against the fp32 head."""
import os
import sys

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import util  # noqa: E402
from os2d_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
MODES = ("f16x3", "f16x2")
for name in util.head_fixture_names():
    fx = util.load_head_fixture(name)
    creator = util.make_head_creator(fx["P"], fx["inverse"], fx["state"], dev)
    with torch.no_grad():
        head = creator.create_os2d_head([c.to(dev) for c in fx["class_fms"]])
        o32 = head(fx["fm"].to(dev), precision="f32")
        outs = {m: head(fx["fm"].to(dev), precision=m) for m in MODES}
    torch.cuda.synchronize()
    print("{:16s} f32 vs ref: cls {:.2e} loc {:.2e} corners {:.2e}".format(
        name, util.maxdiff(o32[1], fx["ref_cls"]), util.maxdiff(o32[0], fx["ref_loc"]), util.maxdiff(o32[3], fx["ref_corners"])))
    for m in MODES:
        o = outs[m]
        print("{:16s} {} vs ref: cls {:.2e} loc {:.2e} corners {:.2e} | vs f32: cls {:.2e} loc {:.2e}".format(
            "", m, util.maxdiff(o[1], fx["ref_cls"]), util.maxdiff(o[0], fx["ref_loc"]), util.maxdiff(o[3], fx["ref_corners"]),
            util.maxdiff(o[1], o32[1]), util.maxdiff(o[0], o32[0])))

for P, inverse, seed in ((6, True, 3), (4, False, 4)):
    state = synthetic.make_transform_net_state(P, seed=seed)
    creator = util.make_head_creator(P, inverse, state, dev)
    fm = synthetic.make_feature_map(1024, 60, 80, seed=11 + seed, A=1).to(dev)
    cf = synthetic.make_class_feature_maps(64, 1024, sizes=[(15, 15), (12, 18), (20, 9), (16, 16)], seed=5 + seed)
    with torch.no_grad():
        head = creator.create_os2d_head([c.to(dev) for c in cf])
        o32 = head(fm, precision="f32")
        for m in MODES:
            o = head(fm, precision=m)
            dl = (o[0] - o32[0]).abs().flatten()
            print("full size P={} inverse={} B=64  {} vs f32: cls {:.2e} loc max {:.2e} p99.9 {:.2e} mean {:.2e} corners {:.2e}".format(
                P, inverse, m, util.maxdiff(o[1], o32[1]), float(dl.max()),
                float(dl.kthvalue(int(0.999 * dl.numel()))[0]), float(dl.mean()), util.maxdiff(o[3], o32[3])))
