"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np
import torch

from os2d_amd.utils import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def head_fixture_names():
    return sorted(os.path.basename(p)[len("head_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "head_*.npz")))


def load_head_fixture(name):
    d = np.load(os.path.join(GOLDEN, "head_{}.npz".format(name)))
    P, inverse = int(d["P"]), bool(d["inverse"])
    state = synthetic.make_transform_net_state(P, seed=int(d["seed_net"]))
    assert abs(synthetic.state_checksum(state) - float(d["net_checksum"])) < 1e-9, "regenerated weights differ"
    fx = dict(P=P, inverse=inverse, state=state, fm=torch.from_numpy(d["fm"]),
              class_fms=[torch.from_numpy(d["class_fm_{}".format(b)]) for b in range(int(d["n_classes"]))])
    for k in d.files:
        if k.startswith("ref_"):
            fx[k] = torch.from_numpy(d[k])
    return fx


def make_head_creator(P, inverse, state, device, stride=16, rec_field=16):
    """Build the product head creator on `device` and load a TransformNet state dict into it."""
    from os2d_amd.modeling.head import build_os2d_head_creator
    from os2d_amd.structures.feature_map import FeatureMapSize
    creator = build_os2d_head_creator(P == 4, False, inverse, FeatureMapSize(w=stride, h=stride),
                                      FeatureMapSize(w=rec_field, h=rec_field))
    creator.aligner.parameter_regressor.load_state_dict(state)
    creator.to(device)
    creator.eval()
    return creator


def maxdiff(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())
