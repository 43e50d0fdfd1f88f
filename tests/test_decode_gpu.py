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
    corners = [torch.from_numpy(d["corners_%d" % i]).to(device) for i in range(L)]
    for name, thr in (("t0", 0.0), ("tinf", float("-inf")), ("t06", 0.6)):
        res = coder.decode_pyramid(locs, clss, sizes, class_ids=list(range(int(d["n_classes"]))),
                                   nms_score_threshold=thr, nms_iou_threshold=0.3, inverse_box_transforms=inverse,
                                   transform_corners_pyramid=corners)
        # the anchors and the transform corners of the surviving boxes, mapped to the original image like the boxes
        assert util.maxdiff(res.get_field("default_boxes").bbox_xyxy, torch.from_numpy(d["ref_%s_default_boxes" % name])) < 1e-3
        assert util.maxdiff(res.get_field("transform_corners"), torch.from_numpy(d["ref_%s_corners" % name])) < 1e-3
        assert len(res) == len(d["ref_%s_scores" % name]), name
        assert torch.equal(res.get_field("labels").cpu(), torch.from_numpy(d["ref_%s_labels" % name]))
        assert torch.equal(res.get_field("scores").cpu(), torch.from_numpy(d["ref_%s_scores" % name]))
        assert util.maxdiff(res.bbox_xyxy, torch.from_numpy(d["ref_%s_boxes" % name])) < 1e-3
        assert res.image_size == orig
    # eval.nms_across_classes (reference box_coder.py:530-532): second NMS over the union of the labels, by score
    from os2d_amd.modeling.box_coder import BoxGridGenerator, Os2dBoxCoder
    coder_x = Os2dBoxCoder(output_box_grid_generator=coder.output_box_grid_generator, do_nms_across_classes=True)
    res = coder_x.decode_pyramid(locs, clss, sizes, class_ids=list(range(int(d["n_classes"]))), nms_score_threshold=0.0,
                                 nms_iou_threshold=0.3, inverse_box_transforms=inverse)
    assert torch.equal(res.get_field("labels").cpu(), torch.from_numpy(d["ref_across_labels"]))
    assert torch.equal(res.get_field("scores").cpu(), torch.from_numpy(d["ref_across_scores"]))
    assert util.maxdiff(res.bbox_xyxy, torch.from_numpy(d["ref_across_boxes"])) < 1e-3


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


def _assert_same_detections(a, b):
    assert len(a) == len(b) and a.image_size == b.image_size
    assert torch.equal(a.get_field("labels"), b.get_field("labels"))
    assert torch.equal(a.get_field("scores"), b.get_field("scores"))
    assert torch.equal(a.bbox_xyxy, b.bbox_xyxy)
    assert torch.equal(a.get_field("default_boxes").bbox_xyxy, b.get_field("default_boxes").bbox_xyxy)
    if a.has_field("transform_corners"):
        assert torch.equal(a.get_field("transform_corners"), b.get_field("transform_corners"))


@pytest.mark.parametrize("H,W,B", [(11, 13, 5), (60, 80, 8), (1, 1, 2), (64, 80, 3)])
@pytest.mark.parametrize("thr", [float("-inf"), 0.0, 0.4])
def test_fused_level_kernel_equals_generic_path(H, W, B, thr, device):
    """os2d_detect_level (decode + filter + sort + NMS + compaction in one launch) against the generic chain
    os2d_decode_boxes -> stable sort -> os2d_nms, which the other tests pin to the reference fixture and the oracle:
    identical boxes, scores, labels, anchors and corners, bit for bit, incl. an anisotropic resize to the original
    image, unsorted class ids and tied scores."""
    from os2d_amd.modeling.box_coder import ResizeBoxes
    from os2d_amd.structures.feature_map import FeatureMapSize
    rs = np.random.RandomState(H * 100 + W + B)
    size = FeatureMapSize(w=16 * W, h=16 * H)
    loc = torch.from_numpy((rs.standard_normal((B, 4, H * W)) * 1.2).astype(np.float32)).to(device)
    cls_np = rs.uniform(-1, 1, size=(B, H * W)).astype(np.float32)
    if H * W > 40:
        cls_np[0, 5:25] = cls_np[0, 5]          # ties: location order must decide
        cls_np[1, :] = 0.5                      # a whole class of equal scores
        loc[1] *= 0.1
    cls = torch.from_numpy(cls_np).to(device)
    corners = torch.from_numpy(rs.uniform(0, 300, size=(B, 8, H * W)).astype(np.float32)).to(device)
    ids = list(rs.permutation(B) * 3 + 1)
    coder = _coder()
    assert coder._decode_single_level_fused([loc], [cls], [size], ids, thr, 0.3, None, None) is not None, "fused path not taken"
    for inverse in (None, [ResizeBoxes(FeatureMapSize(w=int(size.w * 1.7), h=int(size.h * 1.3) + 1))]):
        coder.use_fused_level_kernel = True
        fused = coder.decode_pyramid([loc], [cls], [size], ids, nms_score_threshold=thr, inverse_box_transforms=inverse,
                                     transform_corners_pyramid=[corners])
        coder.use_fused_level_kernel = False
        generic = coder.decode_pyramid([loc], [cls], [size], ids, nms_score_threshold=thr, inverse_box_transforms=inverse,
                                       transform_corners_pyramid=[corners])
        _assert_same_detections(fused, generic)
        assert len(fused) > 0 or thr > 0.3 or H * W == 1


def test_fused_level_kernel_fallbacks(device):
    """Levels too large for LDS, merged labels and unknown box transforms use the generic path."""
    from os2d_amd import _lib
    from os2d_amd.structures.feature_map import FeatureMapSize
    lib = _lib.load()
    assert lib.os2d_detect_level_supported(60, 80) == 1 and lib.os2d_detect_level_supported(96, 128) == 0
    coder = _coder()
    size = FeatureMapSize(w=208, h=176)
    loc = torch.zeros(2, 4, 11 * 13, device=device)
    cls = torch.rand(2, 11 * 13, device=device)
    assert coder._decode_single_level_fused([loc], [cls], [size], [4, 4], 0.0, 0.3, None, None) is None
    from os2d_amd.structures.bounding_box import BoxList
    assert coder._decode_single_level_fused([loc], [cls], [size], [1, 2], 0.0, 0.3, [lambda b: BoxList(b.bbox_xyxy * 2.0, b.image_size)], None) is None
    assert coder._decode_single_level_fused([loc], [cls], [size], [1, 2], 0.0, 0.3, [lambda b: b], None) is not None   # an identity traces to the empty chain
    assert coder._decode_single_level_fused([loc, loc], [cls, cls], [size, size], [1, 2], 0.0, 0.3, None, None) is None
    assert coder._decode_single_level_fused([loc], [cls], [size], [1, 2], 0.0, 0.3, None, None) is not None


def test_all_boxes_survive_long_kept_lists(device):
    """4800 disjoint 8x8 boxes per class: every box is kept, so the kept list outgrows the LDS lists of nms_kernel (2048)
    and detect_level_kernel (pow2/4) and continues in global memory - both paths must still agree and keep everything."""
    from os2d_amd.structures.feature_map import FeatureMapSize
    H, W, B = 60, 80, 3
    rs = np.random.RandomState(11)
    size = FeatureMapSize(w=16 * W, h=16 * H)
    loc = torch.zeros(B, 4, H * W)
    loc[:, 2:] = 5.0 * float(np.log(8.0 / 240.0))
    loc[2, :2] = torch.from_numpy(rs.uniform(-0.1, 0.1, size=(2, H * W)).astype(np.float32))   # jitter < 2.4 px: still disjoint
    cls = torch.from_numpy(rs.uniform(0.1, 1, size=(B, H * W)).astype(np.float32))
    coder = _coder()
    out = []
    for fused in (True, False):
        coder.use_fused_level_kernel = fused
        out.append(coder.decode_pyramid([loc.to(device)], [cls.to(device)], [size], [0, 1, 2], nms_score_threshold=0.0))
    _assert_same_detections(out[0], out[1])
    assert len(out[0]) == B * H * W
    s = out[0].get_field("scores").view(B, H * W).cpu()
    assert torch.equal(s, torch.sort(cls, dim=1, descending=True)[0])


def test_nms_decisions_at_the_iou_threshold(device):
    """Pairs of boxes whose IoU sits within a few ulps of the threshold (0.3): the kernels only divide when
    inter is within 1e-5 of thr * union (os2d_iou_gt) - the decision must still equal the fp32 quotient test
    inter / union > thr of torchvision / the oracle for every pair.  One pair per class list."""
    from oracle import decode_oracle as D
    from os2d_amd.modeling.box_coder import Os2dBoxCoder
    rs = np.random.RandomState(3)
    pairs = []
    for k in range(1500):
        s = float(rs.uniform(0.5, 40.0))                      # scale of the pair
        up = np.float32(np.inf if k % 2 else -np.inf)
        y = np.float32(3.0 * s)
        for _ in range(int(rs.randint(0, 4)) if k % 7 else 0):   # 0..3 ulps off the exact ratio
            y = np.nextafter(y, up, dtype=np.float32)
        pairs.append([[0.0, 0.0, 10.0 * s, 10.0 * s],          # kept (listed first = higher score)
                      [0.0, 0.0, 10.0 * s, float(y)]])         # IoU = y / (10 s) ~ 0.3
    b = torch.tensor(pairs, dtype=torch.float32)              # [1500, 2, 4]
    ref = torch.zeros(b.size(0), 2, dtype=torch.bool)
    iou = torch.zeros(b.size(0))
    for c in range(b.size(0)):
        ref[c, D.greedy_nms(b[c], torch.tensor([2.0, 1.0]), 0.3)] = True
        iou[c] = D.box_iou_matrix(b[c])[0, 1]
    assert int(((iou - 0.3).abs() < 1e-6).sum()) > 1400
    assert bool(ref[:, 0].all()) and 100 < int(ref[:, 1].sum()) < 1400       # both outcomes occur
    keep = Os2dBoxCoder.nms_sorted(b.to(device), torch.full((b.size(0),), 2), 0.3).cpu()
    assert torch.equal(keep, ref)


def test_chunked_nms_matches_reference_fixture(device):
    """Lists longer than ``nms_max_batch``: the reference's chunk-and-repeat NMS (bounding_box.py:343-374), recorded from
    the reference itself with small batch sizes (golden nms_chunked.npz: several passes over several chunks), one list
    at a time and several lists of different lengths in one batched call."""
    import os
    from oracle import decode_oracle as D
    d = np.load(os.path.join(util.GOLDEN, "nms_chunked.npz"))
    coder = _coder()
    for name in d["cases"]:
        boxes, scores = torch.from_numpy(d["boxes_" + name]), torch.from_numpy(d["scores_" + name])
        for thr_name, thr in (("tinf", float("-inf")), ("t0", 0.0)):
            keep = coder._nms_lists(boxes.unsqueeze(0).to(device), scores.unsqueeze(0).to(device),
                                    (scores > thr).unsqueeze(0).to(device), 0.3, int(d["max_batch_" + name]))[0].cpu()
            ref = torch.zeros_like(keep)
            ref[torch.from_numpy(d["ref_{}_{}".format(name, thr_name)])] = True
            assert torch.equal(keep, ref), (name, thr_name)
    # batched: four lists of different lengths, one batch size
    names = list(d["cases"])
    N = max(int(d["boxes_" + n].shape[0]) for n in names)
    B = torch.zeros(len(names), N, 4)
    S = torch.full((len(names), N), -2.0)
    V = torch.zeros(len(names), N, dtype=torch.bool)
    for i, n in enumerate(names):
        k = d["boxes_" + n].shape[0]
        B[i, :k], S[i, :k], V[i, :k] = torch.from_numpy(d["boxes_" + n]), torch.from_numpy(d["scores_" + n]), True
    V &= S > -0.5
    keep = coder._nms_lists(B.to(device), S.to(device), V.to(device), 0.3, 40).cpu()
    for i in range(len(names)):
        sel = V[i].nonzero().squeeze(1)
        ref = torch.zeros(N, dtype=torch.bool)
        ref[sel[D.nms_chunked(B[i][sel], S[i][sel], 0.3, 40)]] = True
        assert torch.equal(keep[i], ref), names[i]


def test_label_order_follows_set_iteration_like_the_reference(device):
    """reference box_coder.py:483 loops ``for real_label in set(class_ids)``: with ids such as [1000, 3, 70] that is NOT
    ascending; both decode paths reproduce the same order."""
    from os2d_amd.structures.feature_map import FeatureMapSize
    rs = np.random.RandomState(21)
    size = FeatureMapSize(w=208, h=176)
    loc = torch.from_numpy(rs.standard_normal((4, 4, 11 * 13)).astype(np.float32)).to(device)
    cls = torch.from_numpy(rs.uniform(0.1, 1, size=(4, 11 * 13)).astype(np.float32)).to(device)
    coder = _coder()
    for ids in ([1000, 3, 70, 5], [70, 1000, 70, 3]):           # unique ids (fused kernel) / a merged label (generic path)
        expect = list(set(ids))
        res = coder.decode_pyramid([loc], [cls], [size], ids, nms_score_threshold=0.0)
        lab = res.get_field("labels").cpu().tolist()
        seen = [l for i, l in enumerate(lab) if i == 0 or lab[i - 1] != l]
        assert seen == expect, (ids, seen, expect)


# ------------------------------------------------------------------------------------------------ pyramid on the device
def test_fused_pyramid_matches_reference_fixture(device):
    """The reference's decode_pyramid fixture (2 levels, 3 classes, 3 score thresholds) through os2d_detect_pyramid
    (``ResizeBoxes`` mappings are what lets decode_pyramid take the device path)."""
    from os2d_amd.modeling.box_coder import ResizeBoxes
    from os2d_amd.structures.feature_map import FeatureMapSize
    d = np.load(util.GOLDEN + "/decode_pyramid.npz")
    L = int(d["n_levels"])
    sizes = [FeatureMapSize(w=int(w), h=int(h)) for w, h in d["img_sizes"]]
    locs = [torch.from_numpy(d["loc_%d" % i]).to(device) for i in range(L)]
    clss = [torch.from_numpy(d["cls_%d" % i]).to(device) for i in range(L)]
    corners = [torch.from_numpy(d["corners_%d" % i]).to(device) for i in range(L)]
    orig = FeatureMapSize(w=int(d["orig_size"][0]), h=int(d["orig_size"][1]))
    inverse = [ResizeBoxes(orig) for _ in range(L)]
    coder = _coder()
    ids = list(range(int(d["n_classes"])))
    assert coder._decode_pyramid_fused(locs, clss, sizes, ids, 0.0, 0.3, inverse, corners) is not None, "device path not taken"
    for name, thr in (("t0", 0.0), ("tinf", float("-inf")), ("t06", 0.6)):
        res = coder.decode_pyramid(locs, clss, sizes, class_ids=ids, nms_score_threshold=thr, nms_iou_threshold=0.3,
                                   inverse_box_transforms=inverse, transform_corners_pyramid=corners)
        assert len(res) == len(d["ref_%s_scores" % name]), name
        assert torch.equal(res.get_field("labels").cpu(), torch.from_numpy(d["ref_%s_labels" % name]))
        assert torch.equal(res.get_field("scores").cpu(), torch.from_numpy(d["ref_%s_scores" % name]))
        assert util.maxdiff(res.bbox_xyxy, torch.from_numpy(d["ref_%s_boxes" % name])) < 1e-3
        assert util.maxdiff(res.get_field("default_boxes").bbox_xyxy, torch.from_numpy(d["ref_%s_default_boxes" % name])) < 1e-3
        assert util.maxdiff(res.get_field("transform_corners"), torch.from_numpy(d["ref_%s_corners" % name])) < 1e-3
        assert res.image_size == orig


def _pyramid_inputs(levels, B, seed, device, loc_scale=1.2):
    from os2d_amd.structures.feature_map import FeatureMapSize
    rs = np.random.RandomState(seed)
    sizes = [FeatureMapSize(w=16 * w, h=16 * h) for h, w in levels]
    locs = [torch.from_numpy((rs.standard_normal((B, 4, h * w)) * loc_scale).astype(np.float32)).to(device) for h, w in levels]
    clss = [torch.from_numpy(rs.uniform(-1, 1, size=(B, h * w)).astype(np.float32)).to(device) for h, w in levels]
    corners = [torch.from_numpy(rs.uniform(0, 300, size=(B, 8, h * w)).astype(np.float32)).to(device) for h, w in levels]
    return sizes, locs, clss, corners


@pytest.mark.parametrize("max_batch", [10000, 300, 97])
@pytest.mark.parametrize("thr", [float("-inf"), 0.0, 0.5])
def test_fused_pyramid_equals_generic_path(max_batch, thr, device):
    """os2d_detect_pyramid against the generic chain (os2d_decode_boxes -> sorts -> os2d_nms per pass, pinned to the
    reference's chunked-NMS fixture by test_chunked_nms_matches_reference_fixture): identical detections bit for bit on a
    4-level pyramid (1,863 candidates per class) with small ``nms_max_batch`` values that force several passes over
    several chunks, tied scores across levels, unsorted class ids and an anisotropic resize to the original image."""
    from os2d_amd.modeling.box_coder import ResizeBoxes
    from os2d_amd.structures.feature_map import FeatureMapSize
    levels = [(9, 12), (15, 20), (23, 30), (27, 31)]
    B = 5
    sizes, locs, clss, corners = _pyramid_inputs(levels, B, 77, device)
    clss[0][0, 3:30] = clss[1][0, 7]                 # ties inside a level and across levels: list order decides
    clss[2][0, 100:140] = clss[1][0, 7]
    clss[3][1, :] = 0.25                             # a whole level of equal scores
    ids = [11, 3, 8, 40, 5]
    orig = FeatureMapSize(w=700, h=433)
    for inverse in ([ResizeBoxes(orig) for _ in levels], None):
        if inverse is None:
            # without transforms back to a common frame only levels of ONE image size can be merged (the reference's cat_boxlist
            # asserts it, bounding_box.py:390-437; so does decode_pyramid here): four "levels" of the same size
            sizes, locs, clss, corners = _pyramid_inputs([levels[2]] * 4, B, 78, device)
            clss[2][0, 100:140] = clss[1][0, 7]
        coder = _coder()
        coder.nms_max_batch = max_batch
        coder.fused_pyramid_passes = 6
        assert coder._decode_pyramid_fused(locs, clss, sizes, ids, thr, 0.3, inverse, corners) is not None
        fused = coder.decode_pyramid(locs, clss, sizes, ids, nms_score_threshold=thr, inverse_box_transforms=inverse,
                                     transform_corners_pyramid=corners)
        coder.use_fused_level_kernel = False
        generic = coder.decode_pyramid(locs, clss, sizes, ids, nms_score_threshold=thr, inverse_box_transforms=inverse,
                                       transform_corners_pyramid=corners)
        _assert_same_detections(fused, generic)
        assert len(fused) > 0


def test_fused_pyramid_falls_back_when_more_passes_are_needed(device):
    """Sparse boxes + a tiny ``nms_max_batch``: the chunk-and-repeat scheme needs more passes than the device path was
    told to launch -> `unfinished` is reported and decode_pyramid transparently uses the generic path; with enough passes
    the device path gives the same result."""
    levels = [(20, 24), (20, 24)]          # one image size: no transforms are needed to merge the levels
    sizes, locs, clss, corners = _pyramid_inputs(levels, 2, 5, device, loc_scale=0.3)
    coder = _coder()
    coder.nms_max_batch = 64
    coder.use_fused_level_kernel = False
    generic = coder.decode_pyramid(locs, clss, sizes, [0, 1], nms_score_threshold=-1.0)
    coder.use_fused_level_kernel = True
    coder.fused_pyramid_passes = 1
    assert coder._decode_pyramid_fused(locs, clss, sizes, [0, 1], -1.0, 0.3, None, None) is None
    _assert_same_detections(coder.decode_pyramid(locs, clss, sizes, [0, 1], nms_score_threshold=-1.0), generic)
    coder.fused_pyramid_passes = 12
    fused = coder._decode_pyramid_fused(locs, clss, sizes, [0, 1], -1.0, 0.3, None, None)
    assert fused is not None
    _assert_same_detections(fused, generic)


def test_fused_pyramid_full_size_seven_levels(device):
    """BASELINE.json configs[4] shape: the 7 levels of a 1280x960 image (39,580 candidates per class), 6 classes, the
    reference's defaults (score threshold -inf, nms_max_batch 10000: four chunks in the first pass)."""
    from os2d_amd.modeling.box_coder import ResizeBoxes
    from os2d_amd.structures.feature_map import FeatureMapSize
    levels = [(30, 40), (38, 50), (48, 64), (60, 80), (72, 96), (84, 112), (96, 128)]
    sizes, locs, clss, corners = _pyramid_inputs(levels, 6, 9, device, loc_scale=1.0)
    inverse = [ResizeBoxes(FeatureMapSize(w=1280, h=960)) for _ in levels]
    ids = [4, 1, 9, 2, 7, 3]
    coder = _coder()
    fused = coder._decode_pyramid_fused(locs, clss, sizes, ids, float("-inf"), 0.3, inverse, corners)
    assert fused is not None
    coder.use_fused_level_kernel = False
    generic = coder.decode_pyramid(locs, clss, sizes, ids, nms_score_threshold=float("-inf"), inverse_box_transforms=inverse,
                                   transform_corners_pyramid=corners)
    _assert_same_detections(fused, generic)
    assert len(fused) > 100


@pytest.mark.parametrize("max_batch,single_level", [(10000, False), (300, False), (10000, True), (97, True)])
def test_fused_pyramid_with_merged_labels_matches_generic_chain(max_batch, single_level, device):
    """Merged labels (class-image views: several head rows carry one class id, reference evaluate.py:241-269,294 +
    box_coder.py:483-487) through os2d_detect_pyramid_merged: a label's rows are NMS-ed together, its candidate list being
    its rows in row order, each row level by level.  Ragged view counts (3 / 1 / 2 rows), interleaved and unsorted ids,
    ties across the rows of a label, several chunks and passes; identical to the generic chain bit for bit."""
    from os2d_amd.modeling.box_coder import ResizeBoxes
    from os2d_amd.structures.feature_map import FeatureMapSize
    levels = [(15, 20)] if single_level else [(9, 12), (15, 20), (23, 30)]
    B = 6
    sizes, locs, clss, corners = _pyramid_inputs(levels, B, 31, device)
    ids = [7, 2, 7, 40, 2, 7]                          # label 7: rows 0, 2, 5; label 2: rows 1, 4; label 40: row 3
    clss[0][2, 5:60] = clss[0][0, 17]                  # ties between two rows of label 7: list (row) order decides
    clss[-1][4, :] = 0.3                               # a whole level of equal scores in label 2's second row
    orig = FeatureMapSize(w=640, h=411)
    for inverse in ([ResizeBoxes(orig) for _ in levels], None):
        if inverse is None and not single_level:      # levels of one image size (see test_fused_pyramid_equals_generic_path)
            sizes, locs, clss, corners = _pyramid_inputs([levels[1]] * 3, B, 32, device)
            clss[0][2, 5:60] = clss[0][0, 17]
        coder = _coder()
        coder.nms_max_batch = max_batch
        coder.fused_pyramid_passes = 8
        thr = 0.1
        if single_level:
            assert coder._decode_single_level_fused(locs, clss, sizes, ids, thr, 0.3, inverse, corners) is None    # one row per label only
        fused = coder._decode_pyramid_fused(locs, clss, sizes, ids, thr, 0.3, inverse, corners)
        assert fused is not None, "merged labels must take the device path"
        coder.use_fused_level_kernel = False
        generic = coder.decode_pyramid(locs, clss, sizes, ids, nms_score_threshold=thr, inverse_box_transforms=inverse,
                                       transform_corners_pyramid=corners)
        _assert_same_detections(fused, generic)
        assert set(fused.get_field("labels").tolist()) == {2, 7, 40}


def test_fused_pyramid_merged_labels_full_size(device):
    """The evaluation's shape with class-image augmentation: 7 levels of a 1280x960 image x 4 labels x 8 views = 32 head
    rows, 316,640 candidates per label (32 chunks of 10000 in the first pass), reference defaults."""
    from os2d_amd.modeling.box_coder import ResizeBoxes
    from os2d_amd.structures.feature_map import FeatureMapSize
    levels = [(30, 40), (38, 50), (48, 64), (60, 80), (72, 96), (84, 112), (96, 128)]
    sizes, locs, clss, corners = _pyramid_inputs(levels, 32, 13, device, loc_scale=1.0)
    inverse = [ResizeBoxes(FeatureMapSize(w=1280, h=960)) for _ in levels]
    ids = [c for c in (3, 0, 9, 5) for _ in range(8)]
    coder = _coder()
    coder.fused_pyramid_passes = 4
    fused = coder._decode_pyramid_fused(locs, clss, sizes, ids, float("-inf"), 0.3, inverse, corners)
    assert fused is not None
    coder.use_fused_level_kernel = False
    generic = coder.decode_pyramid(locs, clss, sizes, ids, nms_score_threshold=float("-inf"), inverse_box_transforms=inverse,
                                   transform_corners_pyramid=corners)
    _assert_same_detections(fused, generic)
    assert len(fused) > 100


def test_levels_of_different_image_sizes_need_transforms(device):
    """The reference merges the levels of a label with cat_boxlist, which asserts that all of them live on one image size
    (bounding_box.py:390-437): without inverse_box_transforms a multi-size pyramid is an error there and here - on the device
    path and on the generic chain alike (the fused path used to label the result with the first level's size)."""
    levels = [(9, 12), (15, 20)]
    sizes, locs, clss, corners = _pyramid_inputs(levels, 2, 3, device)
    for fused in (True, False):
        coder = _coder()
        coder.use_fused_level_kernel = fused
        with pytest.raises(ValueError, match="image size"):
            coder.decode_pyramid(locs, clss, sizes, [0, 1], nms_score_threshold=0.0)


def test_single_level_beyond_the_level_kernel_takes_the_pyramid_path(device):
    """os2d_detect_level keeps a whole level in LDS and stops at ~5,900 locations (60 x 80 fits, 72 x 96 does not): such a
    level is decoded as a pyramid of one level on the device (os2d_detect_pyramid), not through the generic chain."""
    levels = [(72, 96)]
    sizes, locs, clss, corners = _pyramid_inputs(levels, 3, 21, device)
    ids = [5, 1, 9]
    coder = _coder()
    assert coder._decode_single_level_fused(locs, clss, sizes, ids, 0.0, 0.3, None, corners) is None
    fused = coder._decode_pyramid_fused(locs, clss, sizes, ids, 0.0, 0.3, None, corners)
    assert fused is not None
    coder.use_fused_level_kernel = False
    generic = coder.decode_pyramid(locs, clss, sizes, ids, nms_score_threshold=0.0, transform_corners_pyramid=corners)
    _assert_same_detections(fused, generic)
    assert len(fused) > 0


# ------------------------------------------------------------------------- the reference's closure lists on the device path
def test_dataloader_style_transforms_take_the_fused_path_and_match_the_reference(device):
    """VERDICT r3 item 3 / SURVEY 8(b): ``inverse_box_transforms`` as the reference's dataloader builds them - one list of
    closures per level (flips, mined crop, two resizes; os2d/data/dataloader.py:286-336) - now reach the fused decode
    (os2d_detect_pyramid_ops): the device path runs, matches what the reference's decode_pyramid returned for the same
    closures (fixture), and equals the generic chain - which calls the closures on BoxLists - bit for bit, incl. anchors and
    transform corners (which the reference maps as pairs of points, so a flip exchanges their x / y)."""
    from os2d_amd.structures.feature_map import FeatureMapSize
    d = np.load(util.GOLDEN + "/decode_transforms.npz")
    L = int(d["n_levels"])
    sizes = [FeatureMapSize(w=int(w), h=int(h)) for w, h in d["img_sizes"]]
    locs = [torch.from_numpy(d["loc_%d" % i]).to(device) for i in range(L)]
    clss = [torch.from_numpy(d["cls_%d" % i]).to(device) for i in range(L)]
    corners = [torch.from_numpy(d["corners_%d" % i]).to(device) for i in range(L)]
    orig = FeatureMapSize(w=int(d["orig_size"][0]), h=int(d["orig_size"][1]))
    inverse = [util.dataloader_style_inverse(d["chain"][i]) for i in range(L)]
    ids = list(range(int(d["n_classes"])))
    coder = _coder()
    assert coder._decode_pyramid_fused(locs, clss, sizes, ids, 0.0, 0.3, inverse, corners) is not None, "device path not taken"
    for name, thr in (("t0", 0.0), ("tinf", float("-inf"))):
        coder.use_fused_level_kernel = True
        res = coder.decode_pyramid(locs, clss, sizes, class_ids=ids, nms_score_threshold=thr, nms_iou_threshold=0.3,
                                   inverse_box_transforms=inverse, transform_corners_pyramid=corners)
        assert res.image_size == orig and len(res) == len(d["ref_%s_scores" % name]), name
        assert torch.equal(res.get_field("labels").cpu(), torch.from_numpy(d["ref_%s_labels" % name]))
        assert torch.equal(res.get_field("scores").cpu(), torch.from_numpy(d["ref_%s_scores" % name]))
        assert util.maxdiff(res.bbox_xyxy, torch.from_numpy(d["ref_%s_boxes" % name])) < 1e-3
        assert util.maxdiff(res.get_field("default_boxes").bbox_xyxy, torch.from_numpy(d["ref_%s_default_boxes" % name])) < 1e-3
        assert util.maxdiff(res.get_field("transform_corners"), torch.from_numpy(d["ref_%s_corners" % name])) < 1e-3
        coder.use_fused_level_kernel = False
        generic = coder.decode_pyramid(locs, clss, sizes, class_ids=ids, nms_score_threshold=thr, nms_iou_threshold=0.3,
                                       inverse_box_transforms=inverse, transform_corners_pyramid=corners)
        _assert_same_detections(res, generic)
    # one level alone: the single-level kernel (os2d_detect_level_ops) with the same chain; merged labels: 2 views per label
    coder.use_fused_level_kernel = True
    assert coder._decode_single_level_fused(locs[:1], clss[:1], sizes[:1], ids, 0.0, 0.3, inverse[:1], corners[:1]) is not None
    fused1 = coder.decode_pyramid(locs[:1], clss[:1], sizes[:1], ids, nms_score_threshold=0.0, inverse_box_transforms=inverse[:1],
                                  transform_corners_pyramid=corners[:1])
    merged_ids = [7, 2, 7]
    assert coder._decode_pyramid_fused(locs, clss, sizes, merged_ids, 0.0, 0.3, inverse, corners) is not None
    fused_m = coder.decode_pyramid(locs, clss, sizes, merged_ids, nms_score_threshold=0.0, inverse_box_transforms=inverse,
                                   transform_corners_pyramid=corners)
    coder.use_fused_level_kernel = False
    _assert_same_detections(fused1, coder.decode_pyramid(locs[:1], clss[:1], sizes[:1], ids, nms_score_threshold=0.0,
                                                         inverse_box_transforms=inverse[:1], transform_corners_pyramid=corners[:1]))
    _assert_same_detections(fused_m, coder.decode_pyramid(locs, clss, sizes, merged_ids, nms_score_threshold=0.0,
                                                          inverse_box_transforms=inverse, transform_corners_pyramid=corners))
    assert len(fused1) > 0 and len(fused_m) > 0


def test_untraceable_transform_falls_back_to_the_generic_chain(device):
    """An entry that does anything else than BoxList.resize / transpose / crop cannot be handed to the kernels: the decode
    then runs the closures on BoxLists (generic chain) and still returns the right thing."""
    from os2d_amd.structures.bounding_box import BoxList
    from os2d_amd.structures.feature_map import FeatureMapSize
    levels = [(9, 12), (9, 12)]
    sizes, locs, clss, corners = _pyramid_inputs(levels, 2, 31, device)
    def shift2(b):        # not one of the three BoxList operations; keeps the fields like they do (the reference reads them back)
        out = BoxList(b.bbox_xyxy + 2.0, b.image_size)
        for k in b.fields():
            out.add_field(k, b.get_field(k))
        return out
    shift = [shift2 for _ in levels]
    coder = _coder()
    assert coder._decode_pyramid_fused(locs, clss, sizes, [0, 1], 0.0, 0.3, shift, None) is None
    res = coder.decode_pyramid(locs, clss, sizes, [0, 1], nms_score_threshold=0.0, inverse_box_transforms=shift)
    plain = coder.decode_pyramid(locs, clss, sizes, [0, 1], nms_score_threshold=0.0)
    assert len(res) == len(plain) and torch.equal(res.bbox_xyxy, plain.bbox_xyxy + 2.0)
