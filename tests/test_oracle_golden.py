"""CPU: the oracle (oracle/*.py) against the golden vectors recorded from the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU tests then compare HIP <-> oracle/golden."""
import os

import numpy as np
import pytest
import torch

import util
from oracle import decode_oracle as D
from oracle import head_oracle as O


@pytest.mark.parametrize("name", util.head_fixture_names())
def test_head_oracle_is_bit_exact_with_reference(name):
    fx = util.load_head_fixture(name)
    q = O.prepare_class_maps(fx["class_fms"])
    if "ref_q15" in fx:            # the extreme-deformation fixtures (x_*) record outputs + TransformNet parameters only
        assert torch.equal(q, fx["ref_q15"])
    with torch.no_grad():
        loc, cls, cls2, corners, mid = O.head_forward(fx["fm"], q, fx["state"], fx["inverse"], return_intermediate=True)
    # same operator sequence as the reference -> identical bits
    if "ref_corr" in fx:
        assert torch.equal(mid["corr"], fx["ref_corr"])
    assert torch.equal(mid["params"], fx["ref_params"])
    assert torch.equal(loc, fx["ref_loc"])
    assert torch.equal(cls, fx["ref_cls"])
    assert torch.equal(corners, fx["ref_corners"])
    assert cls2 is cls


@pytest.mark.parametrize("name", util.head_fixture_names()[:3])
def test_looped_equals_batched(name):
    """evaluate.py drives one class per head call; batching classes must not change any value."""
    fx = util.load_head_fixture(name)
    q = O.prepare_class_maps(fx["class_fms"])
    with torch.no_grad():
        out = O.head_forward_looped(fx["fm"], q, fx["state"], fx["inverse"])
    assert util.maxdiff(out[0], fx["ref_loc"]) < 1e-6
    assert util.maxdiff(out[1], fx["ref_cls"]) < 1e-6
    assert util.maxdiff(out[3], fx["ref_corners"]) < 1e-4


@pytest.mark.parametrize("name", util.head_fixture_names())
def test_closed_form_fp64_agrees(name):
    """The independent closed form in float64: the error budget of the fp32 reference itself."""
    fx = util.load_head_fixture(name)
    q = O.prepare_class_maps(fx["class_fms"])
    with torch.no_grad():
        loc, cls, _, corners = O.head_forward_closed_form(fx["fm"], q, fx["state"], fx["inverse"])
    util.assert_close(cls, fx["ref_cls"], 2e-5 if name == "x_tiny_scale_inv" else 1e-6, 0.0, name + " cls")
    util.assert_close(loc, fx["ref_loc"], 5e-5, 3e-6, name + " loc")       # -27.4 = 5*log(1/240) for the min-size boxes
    util.assert_close(corners, fx["ref_corners"], 5e-4, 1e-6, name + " corners")


def test_extreme_fixtures_leave_the_identity_neighbourhood():
    """What the x_* fixtures are for: large / tiny scales, rotations, reflections, a 1e-6 determinant,
    the min-size clip, grids outside the map and negative scores are all present in the reference outputs."""
    fx = {n: util.load_head_fixture(n) for n in util.head_fixture_names() if n.startswith("x_")}
    assert len(fx) >= 12
    p = fx["x_scale_large"]["ref_params"]
    assert float(p[:, 0].min()) > 3.5 and float(fx["x_scale_small"]["ref_params"][:, 0].max()) < 0.3
    for n in ("x_rot90_inv", "x_rotm90"):
        p = fx[n]["ref_params"]
        assert float(p[:, 0].abs().max()) < 0.2 and float(p[:, 1].abs().min()) > 0.8           # |cos| ~ 0, |sin| ~ 1
    p = fx["x_reflect_inv"]["ref_params"]
    assert float((p[:, 0] * p[:, 4] - p[:, 1] * p[:, 3]).max()) < -0.5                          # det < 0
    for n, lo, hi in (("x_tiny_scale_inv", 0.5e-6, 2e-6), ("x_near_singular_inv", 1e-6, 4e-6)):
        p = fx[n]["ref_params"]
        det = p[:, 0] * p[:, 4] - p[:, 1] * p[:, 3]
        assert lo < float(det.min()) and float(det.max()) < hi, (n, float(det.min()), float(det.max()))
    # clip_to_min_size: box side exactly 1 px -> loc = 5 * log(1 / 240)
    import math
    for n in ("x_min_size", "x_min_size_v1_inv"):
        assert abs(float(fx[n]["ref_loc"][:, :, 2:4].min()) - 5 * math.log(1 / 240.0)) < 1e-4
    # grids outside the map: template translated by 3 (= 22 cells) / -2.5 template units
    assert float(fx["x_outside_v1"]["ref_params"][:, 1].min()) > 2.9
    assert float(fx["x_neg_scores"]["ref_cls"].min()) < -0.01 < 0.01 < float(fx["x_neg_scores"]["ref_cls"].max())


def test_resample_fast_vs_simple_statement():
    """The reference documents two statements of the resampling (head.py:439-520 fast, :523-594 simple); with the
    symmetric pooling mask they agree.  Re-derive the 'simple' one here from its description: per template point a
    separate bilinear grid_sample of channel x*15+y."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    A, B, H, W, T = 1, 2, 7, 9, 15
    corr = torch.randn(A, B, T * T, H, W)
    grid = torch.rand(A, B, H, W, T, T, 2) * 2.4 - 1.2
    mask = O.pool_mask()
    fast = O.resample_and_pool(corr, grid.clamp(-1, 1), mask)
    acc = torch.zeros(A * B, H, W)
    for x in range(T):
        for y in range(T):
            ch = corr.view(A * B, T * T, H, W)[:, x * T + y:x * T + y + 1]
            pts = grid.view(A * B, H, W, T, T, 2)[:, :, :, y, x, :].clamp(-1, 1)
            acc += mask[y, x] * F.grid_sample(ch, pts, mode="bilinear", padding_mode="border", align_corners=True)[:, 0]
    assert util.maxdiff(fast.view(A * B, H, W), acc) < 1e-6


def test_encode_decode_roundtrip():
    """reference box_coder.py:323-325: build_loc_targets(build_boxes_from_loc_scores(loc)) == loc."""
    rs = np.random.RandomState(3)
    H, W = 6, 7
    loc = torch.from_numpy((rs.standard_normal((2, 4, H * W)) * 1.5).astype(np.float32))
    boxes = D.decode_level(loc, H, W, 1e9, 1e9)           # huge image: no clipping
    boxes_noclip = boxes.clone()
    anchors = O.anchor_grid(H, W, 240.0, 16.0)
    for b in range(2):
        bx = boxes_noclip[b]
        if (bx[:, :2] <= 0).any():
            continue
        back = O.encode_boxes(bx, anchors)
        assert util.maxdiff(back.t(), loc[b]) < 1e-4


def test_decode_pyramid_matches_reference():
    d = np.load(util.GOLDEN + "/decode_pyramid.npz")
    L = int(d["n_levels"])
    img = [tuple(int(v) for v in x) for x in d["img_sizes"]]

    def c4(s):
        for _ in range(4):
            s = (s + 1) // 2
        return s
    fm_sizes = [(c4(h), c4(w)) for w, h in img]
    locs = [torch.from_numpy(d["loc_%d" % i]) for i in range(L)]
    clss = [torch.from_numpy(d["cls_%d" % i]) for i in range(L)]
    for i in range(L):
        bx = D.decode_level(locs[i], fm_sizes[i][0], fm_sizes[i][1], img[i][0], img[i][1])
        assert torch.equal(bx, torch.from_numpy(d["ref_boxes_%d" % i]))
    orig = tuple(int(v) for v in d["orig_size"])
    for name, thr in (("t0", 0.0), ("tinf", float("-inf")), ("t06", 0.6)):
        b, s, l = D.decode_pyramid(locs, clss, fm_sizes, img, orig, thr, 0.3)
        assert torch.equal(b, torch.from_numpy(d["ref_%s_boxes" % name]))
        assert torch.equal(s, torch.from_numpy(d["ref_%s_scores" % name]))
        assert torch.equal(l, torch.from_numpy(d["ref_%s_labels" % name]))
    b, s, l = D.decode_pyramid(locs, clss, fm_sizes, img, orig, 0.0, 0.3, nms_across_classes=True)
    assert torch.equal(b, torch.from_numpy(d["ref_across_boxes"])) and torch.equal(s, torch.from_numpy(d["ref_across_scores"]))
    assert torch.equal(l, torch.from_numpy(d["ref_across_labels"]))


def test_chunked_nms_oracle_matches_reference():
    """oracle.nms_chunked against the reference's own ``nms`` run with small batch sizes (several passes, several chunks)."""
    from oracle import decode_oracle as D
    d = np.load(os.path.join(util.GOLDEN, "nms_chunked.npz"))
    for name in d["cases"]:
        boxes, scores = torch.from_numpy(d["boxes_" + name]), torch.from_numpy(d["scores_" + name])
        for thr_name, thr in (("tinf", float("-inf")), ("t0", 0.0)):
            got = D.nms_chunked(boxes, scores, 0.3, int(d["max_batch_" + name]), thr)
            assert torch.equal(got, torch.from_numpy(d["ref_{}_{}".format(name, thr_name)])), (name, thr_name)
