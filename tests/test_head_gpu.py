"""GPU parity of the HIP head (through the C ABI) against the reference-generated golden fixtures and the oracle.

Tolerances (north_star: score maps within 1e-4 fp32):
  cls      1e-5 absolute  (values ~0.3-0.45; measured ~3e-7)
  loc      1e-4 absolute  (values up to ~1;   measured ~3e-6)
  corners  2e-3 absolute  (pixel coordinates up to ~600; fp32 ulp there is 6e-5; measured ~1e-4)
"""
import pytest
import torch

import util

pytestmark = pytest.mark.gpu

TOL_CLS, TOL_LOC, TOL_CORNERS = 1e-5, 1e-4, 2e-3


@pytest.mark.parametrize("name", util.head_fixture_names())
def test_head_matches_reference_golden(name, device):
    fx = util.load_head_fixture(name)
    creator = util.make_head_creator(fx["P"], fx["inverse"], fx["state"], device)
    with torch.no_grad():
        head = creator.create_os2d_head([c.to(device) for c in fx["class_fms"]])
        loc, cls, cls_det, corners = head(fx["fm"].to(device))
    torch.cuda.synchronize()
    assert cls_det is cls
    assert loc.shape == fx["ref_loc"].shape and cls.shape == fx["ref_cls"].shape and corners.shape == fx["ref_corners"].shape
    assert util.maxdiff(head.class_feature_maps, fx["ref_q15"]) < 1e-6
    assert util.maxdiff(cls, fx["ref_cls"]) < TOL_CLS
    assert util.maxdiff(loc, fx["ref_loc"]) < TOL_LOC
    assert util.maxdiff(corners, fx["ref_corners"]) < TOL_CORNERS
