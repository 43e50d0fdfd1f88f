#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation (aosokin/os2d) in the dev container.

Run only where ``/root/reference`` exists:   python tests/golden/make_golden.py
It writes ``tests/golden/head_*.npz`` + ``decode_*.npz`` (inputs, seeds for the weights, expected outputs).
Nothing from the reference travels: the fixtures are data (inputs and expected outputs).

torchvision is not installed in this image and cannot be installed (no network), while the reference
imports it at module level (reference os2d/modeling/box_coder.py:7, os2d/structures/bounding_box.py:4-5,
os2d/modeling/feature_extractor.py:5).  To import the reference we register a minimal stand-in for
those module names.  Only these stand-in functions are numerically exercised by the paths we record:
  * ``encode_boxes``            (box_coder.py:316, inside Os2dHead.forward)
  * ``BoxCoder.decode_single``  (box_coder.py:329, decode_pyramid)
  * ``clip_boxes_to_image``, ``nms``, ``box_area``  (bounding_box.py:262,367, decode_pyramid)
They are restated here from torchvision's published closed forms (torchvision 0.5 / 0.14,
torchvision/models/detection/_utils.py and torchvision/ops/boxes.py).  Parity at the torchvision
boundary is therefore pinned to the published formulas, not to a torchvision binary.
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REFERENCE = "/root/reference"
sys.path.insert(0, REPO)

from os2d_amd.utils import synthetic  # noqa: E402


# ----------------------------------------------------------------------------- torchvision stand-in
def _tv_encode_boxes(reference_boxes, proposals, weights):
    # published closed form: dx = wx*(gcx-ecx)/ew, dy likewise, dw = ww*log(gw/ew), dh likewise
    wx, wy, ww, wh = [float(w) for w in weights]
    px1, py1, px2, py2 = proposals.unbind(1)
    gx1, gy1, gx2, gy2 = reference_boxes.unbind(1)
    ew, eh = px2 - px1, py2 - py1
    ecx, ecy = px1 + 0.5 * ew, py1 + 0.5 * eh
    gw, gh = gx2 - gx1, gy2 - gy1
    gcx, gcy = gx1 + 0.5 * gw, gy1 + 0.5 * gh
    return torch.stack([wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh,
                        ww * torch.log(gw / ew), wh * torch.log(gh / eh)], dim=1)


class _TvBoxCoder(object):
    def __init__(self, weights, bbox_xform_clip=math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def decode_single(self, rel_codes, boxes):
        boxes = boxes.to(rel_codes.dtype)
        w = boxes[:, 2] - boxes[:, 0]
        h = boxes[:, 3] - boxes[:, 1]
        cx = boxes[:, 0] + 0.5 * w
        cy = boxes[:, 1] + 0.5 * h
        wx, wy, ww, wh = [float(v) for v in self.weights]
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pcx = dx * w[:, None] + cx[:, None]
        pcy = dy * h[:, None] + cy[:, None]
        pw = torch.exp(dw) * w[:, None]
        ph = torch.exp(dh) * h[:, None]
        return torch.cat([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], dim=1)


def _tv_box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def _tv_box_iou(a, b):
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (_tv_box_area(a)[:, None] + _tv_box_area(b)[None, :] - inter)


def _tv_clip_boxes_to_image(boxes, size):
    h, w = size
    x = boxes[:, 0::2].clamp(min=0, max=w)
    y = boxes[:, 1::2].clamp(min=0, max=h)
    return torch.stack([x[:, 0], y[:, 0], x[:, 1], y[:, 1]], dim=1)


def _tv_nms(boxes, scores, thr):
    # greedy NMS: visit by decreasing score, suppress IoU > thr; returns kept indices sorted by score
    order = torch.argsort(scores, descending=True, stable=True)
    keep = []
    suppressed = torch.zeros(boxes.shape[0], dtype=torch.bool)
    iou = _tv_box_iou(boxes, boxes) if boxes.shape[0] else None
    for i in order.tolist():
        if suppressed[i]:
            continue
        keep.append(i)
        suppressed |= iou[i] > thr
    return torch.tensor(keep, dtype=torch.long)


class _TvBottleneck(torch.nn.Module):
    # torchvision.models.resnet.Bottleneck as published (ResNet v1.5: the stride sits on the 3x3 convolution)
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, norm_layer):
        super().__init__()
        nn = torch.nn
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        return self.relu(self.bn3(self.conv3(out)) + idt)


class _TvResNet(torch.nn.Module):
    # torchvision.models.resnet.ResNet as published: conv1 7x7/2, bn1, relu, maxpool 3x3/2, layer1..4, avgpool, fc.
    # The reference subclasses it and copies its __dict__ (os2d/modeling/feature_extractor.py:23-52).
    def __init__(self, layers=(3, 4, 6, 3), norm_layer=None):
        super().__init__()
        nn = torch.nn
        norm_layer = norm_layer or nn.BatchNorm2d
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        for i, (planes, blocks, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2))):
            down = None
            if stride != 1 or self.inplanes != planes * 4:
                down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), norm_layer(planes * 4))
            seq = [_TvBottleneck(self.inplanes, planes, stride, down, norm_layer)]
            self.inplanes = planes * 4
            seq += [_TvBottleneck(self.inplanes, planes, 1, None, norm_layer) for _ in range(1, blocks)]
            setattr(self, "layer{}".format(i + 1), nn.Sequential(*seq))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)


def install_torchvision_standin():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    tv = mod("torchvision")
    models = mod("torchvision.models")
    resnet = mod("torchvision.models.resnet")
    det = mod("torchvision.models.detection")
    utils = mod("torchvision.models.detection._utils")
    ops = mod("torchvision.ops")
    boxes = mod("torchvision.ops.boxes")
    transforms = mod("torchvision.transforms")
    tv.models, tv.ops, tv.transforms = models, ops, transforms
    models.resnet, models.detection = resnet, det
    det._utils = utils
    ops.boxes = boxes

    resnet.ResNet = _TvResNet
    resnet.resnet50 = lambda norm_layer=None, **kw: _TvResNet([3, 4, 6, 3], norm_layer)
    resnet.resnet101 = lambda norm_layer=None, **kw: _TvResNet([3, 4, 23, 3], norm_layer)
    utils.encode_boxes = _tv_encode_boxes
    utils.BoxCoder = _TvBoxCoder

    class Matcher(object):  # training only, never reached
        def __init__(self, *a, **k):
            pass

    utils.Matcher = Matcher
    boxes.nms = ops.nms = _tv_nms
    boxes.box_iou = ops.box_iou = _tv_box_iou
    boxes.box_area = ops.box_area = _tv_box_area
    boxes.clip_boxes_to_image = ops.clip_boxes_to_image = _tv_clip_boxes_to_image


# ----------------------------------------------------------------------------- head fixtures
HEAD_CASES = [
    # name,            P, inverse, C,  H,  W,  class map sizes (h, w),                seeds (fm, cls, net)
    ("v2_affine_inv",  6, True,    64, 11, 13, [(15, 15), (12, 18), (17, 13)],        (11, 1100, 21)),
    ("v1_simple",      4, False,   64, 12, 14, [(15, 15), (18, 12), (13, 17)],        (12, 1200, 22)),
    ("affine_noinv",   6, False,   48,  9, 16, [(14, 16), (15, 15)],                  (13, 1300, 23)),
    ("simple_inv",     4, True,    32, 16,  9, [(15, 15), (16, 14), (15, 15), (11, 20)], (14, 1400, 24)),
    ("v2_c256_wide",   6, True,   256, 10, 33, [(15, 15), (13, 17)],                  (15, 1500, 25)),
]


# Cases that LEAVE the identity neighbourhood (the five above have theta within +-0.18 of identity and scores in
# [0.30, 0.45]): the last TransformNet layer's bias is set to the wanted transform (linear.weight ~ N(0, linear_std) adds a
# small location-dependent part), so every location sees a strongly deformed template.
#   name, P, inverse, C, H, W, class map sizes, seeds (fm, cls, net), linear.bias, linear_std, feature kind
EXTREME_CASES = [
    # box scales 0.2x .. 4x (head.py:405-419), with shear
    ("x_scale_small",      6, False, 32,  9, 11, [(15, 15), (13, 17)], (31, 3100, 41), [0.20, 0.02, 0.05, -0.03, 0.25, -0.04], 0.004, "relu"),
    ("x_scale_large",      6, False, 32,  9, 11, [(15, 15), (13, 17)], (32, 3200, 42), [4.0, 0.3, 0.3, -0.2, 3.0, -0.2], 0.004, "relu"),
    ("x_scale_inv",        6, True,  32,  9, 11, [(15, 15), (17, 13)], (33, 3300, 43), [0.30, 0.05, 0.10, -0.04, 0.25, -0.20], 0.002, "relu"),
    # +-90 degree rotations and reflections (det < 0), with and without the inverse (head.py:111-151)
    ("x_rot90_inv",        6, True,  32, 10,  9, [(15, 15), (12, 18)], (34, 3400, 44), [0.0, -1.0, 0.1, 1.0, 0.0, -0.1], 0.004, "relu"),
    ("x_rotm90",           6, False, 32, 10,  9, [(15, 15), (12, 18)], (35, 3500, 45), [0.0, 1.1, 0.0, -0.9, 0.0, 0.05], 0.004, "relu"),
    ("x_reflect_inv",      6, True,  32,  8, 12, [(15, 15), (16, 14)], (36, 3600, 46), [-1.0, 0.1, 0.0, 0.05, 1.0, 0.0], 0.004, "relu"),
    ("x_reflect_v1",       4, False, 32,  8, 12, [(15, 15), (16, 14)], (37, 3700, 47), [-0.8, 0.1, 1.2, -0.1], 0.004, "relu"),
    # det ~ 1e-6 under use_inverse_geom_model: well conditioned (1e-3 * identity -> boxes 1000x the anchor) ...
    ("x_tiny_scale_inv",   6, True,  16,  7,  8, [(15, 15), (14, 16)], (38, 3800, 48), [1e-3, 0.0, 0.0, 0.0, 1e-3, 0.0], 0.0, "relu"),
    # ... and ill conditioned (cond ~ 2e6: the fp32 LU of torch.inverse is only good to ~10 % there)
    ("x_near_singular_inv", 6, True, 16,  7,  8, [(15, 15), (14, 16)], (39, 3900, 49), [1.0, 1.0, 0.1, 1.0, 1.000002, -0.1], 0.0, "relu"),
    # transformed boxes below one pixel: clip_to_min_size (bounding_box.py:267-277)
    ("x_min_size",         6, False, 32,  9, 11, [(15, 15), (13, 17)], (40, 4000, 50), [0.002, 0.0, 0.3, 0.0, 0.003, -0.2], 0.0002, "relu"),
    ("x_min_size_v1_inv",  4, True,  32,  9, 11, [(15, 15), (13, 17)], (41, 4100, 51), [900.0, 0.3, 700.0, -0.2], 0.0, "relu"),
    # sampling grids (almost) entirely outside the feature map: every tap clamps to the border (head.py:371-384)
    ("x_outside_v1",       4, False, 32,  9, 11, [(15, 15), (13, 17)], (42, 4200, 52), [1.0, 3.0, 1.1, -2.5], 0.004, "relu"),
    # sign-mixed, unnormalised features: negative correlations and scores (the ReLU of head.py:650 is live)
    ("x_neg_scores",       6, True,  32, 10, 12, [(15, 15), (13, 17), (16, 15)], (43, 4300, 53), None, 0.02, "randn"),
]


def make_extreme_fixtures():
    for name, P, inverse, C, H, W, sizes, (s_fm, s_cls, s_net), bias, std, kind in EXTREME_CASES:
        if kind == "relu":
            fm = synthetic.make_feature_map(C, H, W, seed=s_fm)
            class_fms = synthetic.make_class_feature_maps(len(sizes), C, sizes=sizes, seed=s_cls)
        else:
            fm = synthetic.randn_tensor((1, C, H, W), seed=s_fm, scale=3.0)
            class_fms = [synthetic.randn_tensor((1, C, h, w), seed=s_cls + b, scale=0.5) for b, (h, w) in enumerate(sizes)]
        state = synthetic.make_transform_net_state(P, seed=s_net, linear_std=std, linear_bias=bias)
        out = run_reference_head(P, inverse, fm, class_fms, state)
        p = out["params"]
        print("{:20s} cls[{:+.4f},{:+.4f}] params[{:+.3g},{:+.3g}] loc[{:+.3f},{:+.3f}] |corners| max {:.4g}".format(
            name, float(out["cls"].min()), float(out["cls"].max()), float(p.min()), float(p.max()),
            float(out["loc"].min()), float(out["loc"].max()), float(out["corners"].abs().max())))
        arrays = dict(P=np.int64(P), inverse=np.int64(inverse), seed_net=np.int64(s_net),
                      net_checksum=np.float64(synthetic.state_checksum(state)), linear_std=np.float64(std),
                      fm=fm.numpy(), n_classes=np.int64(len(class_fms)))
        if bias is not None:
            arrays["linear_bias"] = np.asarray(bias, dtype=np.float32)
        for b, c in enumerate(class_fms):
            arrays["class_fm_{}".format(b)] = c.numpy()
        for k in ("loc", "cls", "corners", "params"):      # corr / q15 are covered by the five base cases
            arrays["ref_" + k] = out[k].numpy()
        np.savez_compressed(os.path.join(HERE, "head_{}.npz".format(name)), **arrays)


def run_reference_head(P, inverse, fm, class_fms, state):
    from os2d.modeling.head import build_os2d_head_creator
    from os2d.structures.feature_map import FeatureMapSize

    creator = build_os2d_head_creator(P == 4, False, inverse,
                                      FeatureMapSize(w=16, h=16), FeatureMapSize(w=16, h=16))
    creator.aligner.parameter_regressor.load_state_dict(state)
    creator.eval()
    grabbed = {}
    hook = creator.aligner.parameter_regressor.register_forward_hook(
        lambda m, i, o: grabbed.update(corr=i[0].detach().clone(), params=o.detach().clone()))
    with torch.no_grad():
        head = creator.create_os2d_head(class_fms)
        loc, cls, cls_det, corners = head(fm)
    hook.remove()
    return dict(loc=loc, cls=cls, cls_detached=cls_det, corners=corners.contiguous(),
                q15=head.class_feature_maps, corr=grabbed["corr"], params=grabbed["params"])


def make_head_fixtures():
    for name, P, inverse, C, H, W, sizes, (s_fm, s_cls, s_net) in HEAD_CASES:
        fm = synthetic.make_feature_map(C, H, W, seed=s_fm)
        class_fms = synthetic.make_class_feature_maps(len(sizes), C, sizes=sizes, seed=s_cls)
        state = synthetic.make_transform_net_state(P, seed=s_net)
        out = run_reference_head(P, inverse, fm, class_fms, state)
        theta = out["params"]
        print("{:16s} cls[{:+.4f},{:+.4f}] |p - id| max {:.3f}  loc absmax {:.3f}".format(
            name, float(out["cls"].min()), float(out["cls"].max()),
            float((theta - theta.mean(dim=(2, 3), keepdim=True)).abs().max()), float(out["loc"].abs().max())))
        arrays = dict(P=np.int64(P), inverse=np.int64(inverse), seed_net=np.int64(s_net),
                      net_checksum=np.float64(synthetic.state_checksum(state)),
                      fm=fm.numpy(), n_classes=np.int64(len(class_fms)))
        for b, c in enumerate(class_fms):
            arrays["class_fm_{}".format(b)] = c.numpy()
        for k, v in out.items():
            arrays["ref_" + k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, "head_{}.npz".format(name)), **arrays)


# ----------------------------------------------------------------------------- decode fixtures
def make_decode_fixture():
    """decode_pyramid (reference box_coder.py:448-536) on a 2-level pyramid, 3 classes."""
    from os2d.modeling.box_coder import Os2dBoxCoder, BoxGridGenerator
    from os2d.structures.feature_map import FeatureMapSize

    gen = BoxGridGenerator(box_size=FeatureMapSize(w=240, h=240), box_stride=FeatureMapSize(w=16, h=16))

    def fm_size(img_size):
        f = lambda s: -(-(-(-(-(-(-(-s // 2)) // 2)) // 2)) // 2)  # ceil-halving four times
        return FeatureMapSize(w=f(img_size.w), h=f(img_size.h))

    coder = Os2dBoxCoder(0.5, 0.1, 0.5, 0.1, gen, fm_size, do_nms_across_classes=False)
    rs = np.random.RandomState(77)
    img_sizes = [FeatureMapSize(w=208, h=176), FeatureMapSize(w=320, h=272)]
    n_cls = 3
    locs, clss = [], []
    for s in img_sizes:
        f = fm_size(s)
        n = f.w * f.h
        locs.append(torch.from_numpy((rs.standard_normal((n_cls, 4, n)) * np.array([2.0, 2.0, 1.5, 1.5])[None, :, None]).astype(np.float32)))
        clss.append(torch.from_numpy(rs.uniform(-1, 1, size=(n_cls, n)).astype(np.float32)))
    arrays = dict(n_levels=np.int64(len(img_sizes)), n_classes=np.int64(n_cls),
                  img_sizes=np.array([[s.w, s.h] for s in img_sizes], dtype=np.int64))
    for i, (l, c) in enumerate(zip(locs, clss)):
        arrays["loc_{}".format(i)] = l.numpy()
        arrays["cls_{}".format(i)] = c.numpy()
        # per-level decode (box_coder.py:319-330) + clip, before masking / NMS
        default_boxes = coder._get_default_boxes(img_sizes[i])
        per_class = []
        for k in range(n_cls):
            bl = coder.build_boxes_from_loc_scores(l[k].transpose(0, 1), default_boxes)
            bl.clip_to_image(remove_empty=False)
            per_class.append(bl.bbox_xyxy.clone())
        arrays["ref_boxes_{}".format(i)] = torch.stack(per_class, 0).numpy()
    # the eval path maps every level back to the original image with a resize (reference
    # os2d/data/dataloader.py:326-336, os2d/structures/transforms.py:12-27,70-96)
    from os2d.structures.transforms import TransformList
    orig_size = FeatureMapSize(w=416, h=352)
    arrays["orig_size"] = np.array([orig_size.w, orig_size.h], dtype=np.int64)
    inverse = []
    for s in img_sizes:
        t = TransformList()
        t.append(lambda boxes: boxes.resize(orig_size))
        inverse.append(t)
    # transform corners (box_coder.py:439-446,493-503): random parallelogram end points per location
    corners = [torch.from_numpy(rs.uniform(0, 300, size=(n_cls, 8, l.shape[2])).astype(np.float32)) for l in locs]
    for i, k in enumerate(corners):
        arrays["corners_{}".format(i)] = k.numpy()
    for thr_name, score_thr in (("t0", 0.0), ("tinf", float("-inf")), ("t06", 0.6)):
        res = coder.decode_pyramid([l.clone() for l in locs], [c.clone() for c in clss], img_sizes,
                                   class_ids=list(range(n_cls)), nms_score_threshold=score_thr,
                                   nms_iou_threshold=0.3, inverse_box_transforms=inverse,
                                   transform_corners_pyramid=[k.clone() for k in corners])
        order = torch.argsort(res.get_field("labels") * 10 - res.get_field("scores"), stable=True)
        arrays["ref_{}_default_boxes".format(thr_name)] = res.get_field("default_boxes").bbox_xyxy[order].numpy()
        arrays["ref_{}_corners".format(thr_name)] = res.get_field("transform_corners")[order].numpy()
        arrays["ref_{}_boxes".format(thr_name)] = res.bbox_xyxy[order].numpy()
        arrays["ref_{}_scores".format(thr_name)] = res.get_field("scores")[order].numpy()
        arrays["ref_{}_labels".format(thr_name)] = res.get_field("labels")[order].numpy()
        print("decode {}: {} detections".format(thr_name, len(order)))
    # eval.nms_across_classes = True (config.py:202 default False): a second NMS over the union of all labels
    # (box_coder.py:530-532), result sorted by score
    coder_x = Os2dBoxCoder(0.5, 0.1, 0.5, 0.1, gen, fm_size, do_nms_across_classes=True)
    res = coder_x.decode_pyramid([l.clone() for l in locs], [c.clone() for c in clss], img_sizes,
                                 class_ids=list(range(n_cls)), nms_score_threshold=0.0,
                                 nms_iou_threshold=0.3, inverse_box_transforms=inverse)
    arrays["ref_across_boxes"] = res.bbox_xyxy.numpy()
    arrays["ref_across_scores"] = res.get_field("scores").numpy()
    arrays["ref_across_labels"] = res.get_field("labels").numpy()
    print("decode across classes: {} detections".format(len(res)))
    np.savez_compressed(os.path.join(HERE, "decode_pyramid.npz"), **arrays)


def make_model_fixture():
    """Os2dModel.forward (reference model.py:235-276) end to end on CPU: images + class images -> backbone ->
    class heads -> outputs, for the two branch layouts; also records the reference's state-dict keys and shapes.
    The ResNet comes from the published-architecture stand-in above (torchvision is absent), weights from
    ``synthetic.fill_model_state`` (a function of key names and shapes)."""
    import logging
    from os2d.modeling.model import Os2dModel
    arrays = {}
    g = torch.Generator().manual_seed(99)
    image = torch.randn(1, 3, 96, 128, generator=g)
    class_images = [torch.randn(3, 64, 64, generator=g), torch.randn(3, 48, 80, generator=g)]
    arrays["image"] = image.numpy()
    for i, c in enumerate(class_images):
        arrays["class_image_{}".format(i)] = c.numpy()
    # v1_r101 = BASELINE.json configs[3]: ResNet101 backbone (reference feature_extractor.py:120-129), V1-style simplified
    # affine head, separate branches
    for name, merge, simplify, inverse, arch in (("v2_merged", True, False, True, "resnet50"),
                                                 ("v1_split", False, True, False, "resnet50"),
                                                 ("v1_r101", False, True, False, "resnet101")):
        net = Os2dModel(logger=logging.getLogger("golden"), is_cuda=False, merge_branch_parameters=merge,
                        backbone_arch=arch, use_inverse_geom_model=inverse, simplify_affine=simplify)
        sd = net.state_dict()
        filled = synthetic.fill_model_state(sd, seed=500, P=4 if simplify else 6)
        net.load_state_dict(filled)
        net.eval()
        with torch.no_grad():
            loc, cls, cls_det, fm_size, corners = net(images=image, class_images=class_images)
        arrays["keys_" + name] = np.array(list(sd.keys()))
        arrays["shapes_" + name] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
        arrays["checksum_" + name] = np.float64(synthetic.state_checksum(filled))
        arrays["ref_loc_" + name] = loc.numpy()
        arrays["ref_cls_" + name] = cls.numpy()
        arrays["ref_corners_" + name] = corners.numpy()
        arrays["fm_size_" + name] = np.array([fm_size.w, fm_size.h], dtype=np.int64)
        print("model {}: {} state-dict entries, outputs {} {}".format(name, len(sd), tuple(loc.shape), tuple(cls.shape)))
    np.savez_compressed(os.path.join(HERE, "model_forward.npz"), **arrays)


def make_decode_transforms_fixture():
    """decode_pyramid with the inverse box transforms the reference's OWN dataloader code builds (os2d/data/dataloader.py:
    286-336): image -> hflip + vflip (transforms.py:32-52) -> mined crop (:84-191) -> resize to the augmentation size (:55-81) ->
    one more resize per pyramid level on a deep copy of the list.  The TransformList of a level then undoes, in order:
    pyramid resize, augmentation resize, crop ("uncrop"), vertical flip, horizontal flip.  The fixture stores that chain as
    data (kind + parameters per step, in the order the inverse applies them) and what the reference's decode_pyramid returns."""
    import copy
    from PIL import Image
    from os2d.modeling.box_coder import Os2dBoxCoder, BoxGridGenerator
    from os2d.structures import transforms as T
    from os2d.structures.bounding_box import BoxList
    from os2d.structures.feature_map import FeatureMapSize

    gen = BoxGridGenerator(box_size=FeatureMapSize(w=240, h=240), box_stride=FeatureMapSize(w=16, h=16))

    def fm_size(img_size):
        f = lambda s: -(-(-(-(-(-(-(-s // 2)) // 2)) // 2)) // 2)
        return FeatureMapSize(w=f(img_size.w), h=f(img_size.h))

    coder = Os2dBoxCoder(0.5, 0.1, 0.5, 0.1, gen, fm_size, do_nms_across_classes=False)
    rs = np.random.RandomState(91)
    orig = Image.new("RGB", (500, 380))
    orig_size = FeatureMapSize(img=orig)
    boxes = BoxList(torch.tensor([[40.0, 30.0, 200.0, 180.0], [250.0, 100.0, 470.0, 350.0]]), orig_size, mode="xyxy")
    tl = T.TransformList()
    img, boxes = T.transpose(orig, hflip=True, vflip=True, boxes=boxes, transform_list=tl)
    crop_xyxy = (37, 21, 421, 341)
    crop_pos = BoxList(torch.tensor([[float(v) for v in crop_xyxy]]), FeatureMapSize(img=img), mode="xyxy")
    img, boxes, _, _ = T.crop(img, crop_position=crop_pos, boxes=boxes, transform_list=tl)
    crop_size = FeatureMapSize(img=img)
    aug_size = FeatureMapSize(w=320, h=272)
    img, boxes = T.resize(img, target_size=aug_size, boxes=boxes, transform_list=tl)
    level_sizes = [FeatureMapSize(w=208, h=176), FeatureMapSize(w=320, h=272), FeatureMapSize(w=400, h=340)]
    inverse, spec = [], []
    uncrop = (-crop_xyxy[0], -crop_xyxy[1], -crop_xyxy[0] + orig_size.w, -crop_xyxy[1] + orig_size.h)
    for p_size in level_sizes:
        tl_l = copy.deepcopy(tl)
        T.resize(img, target_size=p_size, boxes=boxes, transform_list=tl_l)
        inverse.append(tl_l)
        # kind 1 = resize to (w, h); 4 = crop (left, top, right, bottom); 3 = FLIP_TOP_BOTTOM; 2 = FLIP_LEFT_RIGHT
        spec.append([[1, aug_size.w, aug_size.h, 0, 0], [1, crop_size.w, crop_size.h, 0, 0], [4] + list(uncrop),
                     [3, 0, 0, 0, 0], [2, 0, 0, 0, 0]])
    n_cls = 3
    arrays = dict(n_levels=np.int64(len(level_sizes)), n_classes=np.int64(n_cls),
                  img_sizes=np.array([[s.w, s.h] for s in level_sizes], dtype=np.int64),
                  orig_size=np.array([orig_size.w, orig_size.h], dtype=np.int64), chain=np.array(spec, dtype=np.float64))
    locs, clss, corners = [], [], []
    for i, s_ in enumerate(level_sizes):
        f = fm_size(s_)
        n = f.w * f.h
        locs.append(torch.from_numpy((rs.standard_normal((n_cls, 4, n)) * np.array([2.0, 2.0, 1.5, 1.5])[None, :, None]).astype(np.float32)))
        clss.append(torch.from_numpy(rs.uniform(-1, 1, size=(n_cls, n)).astype(np.float32)))
        corners.append(torch.from_numpy(rs.uniform(0, 300, size=(n_cls, 8, n)).astype(np.float32)))
        arrays["loc_{}".format(i)], arrays["cls_{}".format(i)], arrays["corners_{}".format(i)] = \
            locs[-1].numpy(), clss[-1].numpy(), corners[-1].numpy()
    # the chain on plain boxes (what every step does to coordinates and to the image size)
    probe = torch.tensor([[0.0, 0.0, 1.0, 1.0], [13.25, 7.5, 211.0, 95.75], [3.0, 150.5, 200.125, 170.0]])
    for i, s_ in enumerate(level_sizes):
        bl = BoxList(probe.clone(), s_, mode="xyxy")
        # the anchors as decode_pyramid carries them: a BoxList FIELD of the boxes (transposed / cropped with them, not resized:
        # bounding_box.py:162,196-199,222-225), then the transform once more on the field (box_coder.py:515-516)
        bl.add_field("default_boxes", BoxList(probe.clone() + 0.5, s_, mode="xyxy"))
        out = inverse[i](bl)
        assert out.image_size == orig_size
        arrays["probe_out_{}".format(i)] = out.bbox_xyxy.numpy()
        arrays["probe_default_out_{}".format(i)] = inverse[i](out.get_field("default_boxes")).bbox_xyxy.numpy()
    arrays["probe"] = probe.numpy()
    for thr_name, score_thr in (("t0", 0.0), ("tinf", float("-inf"))):
        res = coder.decode_pyramid([l.clone() for l in locs], [c.clone() for c in clss], level_sizes,
                                   class_ids=list(range(n_cls)), nms_score_threshold=score_thr, nms_iou_threshold=0.3,
                                   inverse_box_transforms=inverse, transform_corners_pyramid=[k.clone() for k in corners])
        order = torch.argsort(res.get_field("labels") * 10 - res.get_field("scores"), stable=True)
        arrays["ref_{}_default_boxes".format(thr_name)] = res.get_field("default_boxes").bbox_xyxy[order].numpy()
        arrays["ref_{}_corners".format(thr_name)] = res.get_field("transform_corners")[order].numpy()
        arrays["ref_{}_boxes".format(thr_name)] = res.bbox_xyxy[order].numpy()
        arrays["ref_{}_scores".format(thr_name)] = res.get_field("scores")[order].numpy()
        arrays["ref_{}_labels".format(thr_name)] = res.get_field("labels")[order].numpy()
        assert res.image_size == orig_size
        print("decode with dataloader transforms {}: {} detections".format(thr_name, len(order)))
    np.savez_compressed(os.path.join(HERE, "decode_transforms.npz"), **arrays)


def make_chunked_nms_fixture():
    """The reference's memory-bounded NMS (bounding_box.py:343-374: lists longer than nms_max_batch are NMS-ed in
    chunks of that size, survivors concatenated chunk by chunk in score order, repeated until one chunk is left or
    nothing changes) on random boxes with batch sizes small enough that 2-3 passes with several chunks happen."""
    from os2d.structures.bounding_box import BoxList, nms
    from os2d.structures.feature_map import FeatureMapSize
    rs = np.random.RandomState(123)
    arrays = {}
    cases = [("a", 600, 900.0, 64), ("b", 600, 250.0, 50), ("c", 257, 2000.0, 16), ("d", 90, 60.0, 100)]
    for name, n, spread, max_batch in cases:
        ctr = rs.uniform(0, spread, size=(n, 2))
        wh = rs.uniform(10, 60, size=(n, 2))
        boxes = torch.from_numpy(np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32))
        scores = torch.from_numpy(rs.uniform(-1, 1, size=n).astype(np.float32))
        bl = BoxList(boxes, FeatureMapSize(w=4000, h=4000), mode="xyxy")
        bl.add_field("scores", scores)
        for thr_name, score_thr in (("tinf", float("-inf")), ("t0", 0.0)):
            keep = nms(bl, 0.3, nms_max_batch=max_batch, nms_score_threshold=score_thr)
            arrays["ref_{}_{}".format(name, thr_name)] = keep.numpy().astype(np.int64)
            print("chunked nms {} {}: {} of {} kept (batch {})".format(name, thr_name, keep.numel(), n, max_batch))
        arrays["boxes_" + name] = boxes.numpy()
        arrays["scores_" + name] = scores.numpy()
        arrays["max_batch_" + name] = np.int64(max_batch)
    arrays["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "nms_chunked.npz"), **arrays)


def main():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("the reference checkout {} is not present; fixtures can only be regenerated "
                         "in the development container".format(REFERENCE))
    install_torchvision_standin()
    sys.path.insert(0, REFERENCE)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    make_head_fixtures()
    make_extreme_fixtures()
    make_decode_fixture()
    make_chunked_nms_fixture()
    make_decode_transforms_fixture()
    make_model_fixture()


if __name__ == "__main__":
    main()
