"""GPU parity of the decode path (os2d_decode_boxes, os2d_nms, Os2dBoxCoder.decode_pyramid) against the
reference-generated fixture and the decode oracle.  Box decode is fp32 elementwise: tolerance 1e-3 px on values up to
~400 (measured ~3e-5); NMS decisions and orderings must be identical."""
import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu


def _coder():
    from os2d_amd.modeling.box_coder import BoxGridGenerator, Os2dBoxCoder
    from os2d_amd.structures.feature_map import FeatureMapSize
    gen = BoxGridGenerator(box_size=FeatureMapSize(w=240, h=240), box_stride=FeatureMapSize(w=16, h=16))
    return Os2dBoxCoder(output_box_grid_generator=gen)


def test_decode_level_and_pyramid_match_reference(device):
    from os2d_amd.structures.feature_map import FeatureMapSize
    d = np.load(util.GOLDEN + "/decode_pyramid.npz")
    L = int(d["n_levels"])
    sizes = [FeatureMapSize(w=int(w), h=int(h)) for w, h in d["img_sizes"]]
    locs = [torch.from_numpy(d["loc_%d" % i]).to(device) for i in range(L)]
    clss = [torch.from_numpy(d["cls_%d" % i]).to(device) for i in range(L)]
    coder = _coder()
    for i in range(L):
        boxes = coder.decode_level(locs[i], sizes[i])
        assert util.maxdiff(boxes, torch.from_numpy(d["ref_boxes_%d" % i])) < 1e-3
    orig = FeatureMapSize(w=int(d["orig_size"][0]), h=int(d["orig_size"][1]))
    inverse = [(lambda b: b.resize(orig)) for _ in range(L)]
    for name, thr in (("t0", 0.0), ("tinf", float("-inf")), ("t06", 0.6)):
        res = coder.decode_pyramid(locs, clss, sizes, class_ids=list(range(int(d["n_classes"]))),
                                   nms_score_threshold=thr, nms_iou_threshold=0.3, inverse_box_transforms=inverse)
        assert len(res) == len(d["ref_%s_scores" % name]), name
        assert torch.equal(res.get_field("labels").cpu(), torch.from_numpy(d["ref_%s_labels" % name]))
        assert torch.equal(res.get_field("scores").cpu(), torch.from_numpy(d["ref_%s_scores" % name]))
        assert util.maxdiff(res.bbox_xyxy, torch.from_numpy(d["ref_%s_boxes" % name])) < 1e-3
        assert res.image_size == orig


@pytest.mark.parametrize("n,spread", [(1, 50.0), (63, 40.0), (64, 40.0), (65, 30.0), (700, 120.0), (4800, 400.0)])
def test_nms_matches_greedy_oracle(n, spread, device):
    """Random overlapping boxes; several class lists of different valid lengths in one launch."""
    from oracle import decode_oracle as D
    from os2d_amd.modeling.box_coder import Os2dBoxCoder
    rs = np.random.RandomState(n)
    NC = 3
    ctr = rs.uniform(0, spread, size=(NC, n, 2))
    wh = rs.uniform(5, 40, size=(NC, n, 2))
    boxes = torch.from_numpy(np.concatenate([ctr - wh / 2, ctr + wh / 2], -1).astype(np.float32))
    scores = torch.from_numpy(rs.uniform(-1, 1, size=(NC, n)).astype(np.float32))
    counts = torch.tensor([n, max(n // 2, 1), max(n - 1, 1)])
    order = torch.argsort(scores, dim=1, descending=True, stable=True)
    b_sorted = torch.gather(boxes, 1, order.unsqueeze(-1).expand(-1, -1, 4))
    keep = Os2dBoxCoder.nms_sorted(b_sorted.to(device), counts, 0.3).cpu()
    for c in range(NC):
        m = int(counts[c])
        ref = D.greedy_nms(b_sorted[c, :m], torch.arange(m, 0, -1).float(), 0.3)
        got = keep[c, :m].nonzero().squeeze(1)
        assert torch.equal(got, torch.sort(ref)[0]), "class {} differs".format(c)
        assert not keep[c, m:].any()


def test_duplicate_class_ids_are_merged(device):
    """Two heads of the same real label (class-image augmentation, evaluate.py:241-269) are NMS-ed together."""
    from oracle import decode_oracle as D
    from os2d_amd.structures.feature_map import FeatureMapSize
    rs = np.random.RandomState(5)
    size = FeatureMapSize(w=208, h=176)
    H, W = 11, 13
    loc = torch.from_numpy((rs.standard_normal((3, 4, H * W)) * 1.5).astype(np.float32))
    cls = torch.from_numpy(rs.uniform(-1, 1, size=(3, H * W)).astype(np.float32))
    coder = _coder()
    res = coder.decode_pyramid([loc.to(device)], [cls.to(device)], [size], class_ids=[7, 3, 7], nms_score_threshold=0.2)
    # oracle: label 3 = row 1; label 7 = rows 0 and 2 as two "levels" of one class
    b3, s3, _ = D.decode_pyramid([loc[1:2]], [cls[1:2]], [(H, W)], [(208, 176)], None, 0.2, 0.3)
    b7, s7, _ = D.decode_pyramid([loc[0:1], loc[2:3]], [cls[0:1], cls[2:3]], [(H, W), (H, W)], [(208, 176)] * 2, None, 0.2, 0.3)
    lab = res.get_field("labels").cpu()
    assert lab.tolist() == [3] * len(s3) + [7] * len(s7)
    assert torch.equal(res.get_field("scores").cpu(), torch.cat([s3, s7]))
    assert util.maxdiff(res.bbox_xyxy, torch.cat([b3, b7])) < 1e-3


def test_no_detections_and_single_box(device):
    """Edge cases of the decode path: a threshold above every score (empty result), and lists with one survivor."""
    from os2d_amd.structures.feature_map import FeatureMapSize
    rs = np.random.RandomState(8)
    size = FeatureMapSize(w=208, h=176)
    H, W = 11, 13
    loc = torch.from_numpy(rs.standard_normal((2, 4, H * W)).astype(np.float32)).to(device)
    cls = torch.from_numpy(rs.uniform(-1, 1, size=(2, H * W)).astype(np.float32)).to(device)
    coder = _coder()
    res = coder.decode_pyramid([loc], [cls], [size], class_ids=[0, 1], nms_score_threshold=2.0)
    assert len(res) == 0 and res.bbox_xyxy.shape == (0, 4) and res.get_field("scores").numel() == 0
    cls2 = cls.clone()
    cls2[1] = -1.0
    cls2[1, 17] = 0.9        # class 1: exactly one candidate above the threshold
    res = coder.decode_pyramid([loc], [cls2], [size], class_ids=[0, 1], nms_score_threshold=0.8)
    lab = res.get_field("labels").cpu()
    assert int((lab == 1).sum()) == 1 and abs(float(res.get_field("scores")[lab.tolist().index(1)]) - 0.9) < 1e-6
