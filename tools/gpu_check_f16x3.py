#!/usr/bin/env python3
"""GPU: compare the f16x3 head with the fp32 head and the reference golden outputs on every fixture."""
import os
import sys

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import util  # noqa: E402

dev = torch.device("cuda:0")
for name in util.head_fixture_names():
    fx = util.load_head_fixture(name)
    creator = util.make_head_creator(fx["P"], fx["inverse"], fx["state"], dev)
    with torch.no_grad():
        head = creator.create_os2d_head([c.to(dev) for c in fx["class_fms"]])
        o32 = head(fx["fm"].to(dev), precision="f32")
        o16 = head(fx["fm"].to(dev), precision="f16x3")
    torch.cuda.synchronize()
    print("{:16s} f16x3 vs ref: cls {:.2e} loc {:.2e} corners {:.2e} | f32 vs ref: cls {:.2e} loc {:.2e} | f16x3 vs f32: cls {:.2e} loc {:.2e}".format(
        name, util.maxdiff(o16[1], fx["ref_cls"]), util.maxdiff(o16[0], fx["ref_loc"]), util.maxdiff(o16[3], fx["ref_corners"]),
        util.maxdiff(o32[1], fx["ref_cls"]), util.maxdiff(o32[0], fx["ref_loc"]),
        util.maxdiff(o16[1], o32[1]), util.maxdiff(o16[0], o32[0])))
