"""Anchor grids and the decode side of the reference's box coder (os2d/modeling/box_coder.py) on the HIP library.

    create_strided_boxes_columnfirst / BoxGridGenerator   reference box_coder.py:16-76
    Os2dBoxCoder.build_boxes_from_loc_scores              reference box_coder.py:319-330   (os2d_decode_boxes)
    Os2dBoxCoder.decode_pyramid                           reference box_coder.py:448-536   (os2d_detect_level for one level,
                                                                                            os2d_decode_boxes + os2d_nms otherwise)

Only the inference half of the reference class is mirrored: target encoding / anchor matching / hard-negative
remapping are training-only and out of scope.
"""
import collections
import weakref
import ctypes
from functools import lru_cache

import torch

from .. import _lib
from ..structures.feature_map import FeatureMapSize
from ..structures.bounding_box import BoxList, FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM

BOX_ENCODING_WEIGHTS = (10.0, 10.0, 5.0, 5.0)   # reference box_coder.py:13
OP_SCALE, OP_HFLIP, OP_VFLIP, OP_SHIFT = 1, 2, 3, 4   # OS2D_BOX_OP_* of include/os2d_hip.h
MAX_BOX_OPS = 6                                       # OS2D_BOX_MAX_OPS
MAX_DEFAULT_BOX_OPS = 12                              # OS2D_BOX_MAX_DEFAULT_OPS


@lru_cache()
def create_strided_boxes_columnfirst(grid_size, box_size, box_stride):
    """Row-major (index h*W + w, despite the historical name) grid of xyxy anchor boxes with centres
    ((w+.5)*stride_w, (h+.5)*stride_h); CPU float tensor [H*W, 4] (reference box_coder.py:16-59)."""
    cy = (torch.arange(grid_size.h, dtype=torch.float32) + 0.5) * box_stride.h
    cx = (torch.arange(grid_size.w, dtype=torch.float32) + 0.5) * box_stride.w
    cx = cx.view(1, -1).expand(grid_size.h, grid_size.w).reshape(-1)
    cy = cy.view(-1, 1).expand(grid_size.h, grid_size.w).reshape(-1)
    hw, hh = box_size.w / 2.0, box_size.h / 2.0
    return torch.stack([cx - hw, cy - hh, cx + hw, cy + hh], dim=1)


class ResizeBoxes(object):
    """``inverse_box_transforms`` entry that maps a level's boxes to ``target_size`` (what the reference's
    ``TransformList`` inverse amounts to for a resized image, box_coder.py:499-503): ``boxlist.resize(target_size)``.
    (Any entry made of BoxList.resize / transpose / crop calls - this class, the reference's ``TransformList`` of lambdas -
    reaches the fused decode kernels: ``trace_box_transform`` records the chain.)"""

    def __init__(self, target_size):
        self.target_size = target_size

    def __call__(self, boxlist):
        return boxlist.resize(self.target_size)

    def ratios(self, img_size):
        return float(self.target_size.w) / img_size.w, float(self.target_size.h) / img_size.h


class _BoxTrace(object):
    """Stand-in for a ``BoxList`` that RECORDS what an ``inverse_box_transforms`` entry does to it.  The reference hands
    ``decode_pyramid`` one ``TransformList`` of closures per level (os2d/structures/transforms.py:12-27; the closures are
    ``lambda boxes: boxes.resize(size)`` / ``boxes.transpose(FLIP_*)`` / ``boxes.crop(uncrop_xyxy)``, appended by
    transforms.py:32-52, 78-79, 188-191 from os2d/data/dataloader.py:286-336) and calls it on BoxLists
    (os2d/modeling/box_coder.py:499-503).  Called on this object instead, the same closures leave the exact sequence of
    operations with their parameters - which the fused decode kernels then apply op by op with the reference's roundings
    (a numerical probe could only recover the composite map, not where it rounds).  Anything else a closure might touch
    (``bbox_xyxy``, fields, ...) is not defined here: the AttributeError sends the caller to the generic chain."""

    def __init__(self, image_size, ops=(), fields=None):
        self.image_size = image_size
        self.ops = tuple(ops)
        self.extra_fields = dict(fields or {})     # name -> _BoxTrace (BoxList-valued fields: "default_boxes")

    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def resize(self, target_size):                      # BoxList.resize, reference bounding_box.py:138-163
        op = (OP_SCALE, float(target_size.w) / self.image_size.w, float(target_size.h) / self.image_size.h)
        return _BoxTrace(target_size, self.ops + (op,), self.extra_fields)          # fields are copied AS THEY ARE (:162)

    def transpose(self, method):                        # reference bounding_box.py:165-200
        if method == FLIP_LEFT_RIGHT:
            op = (OP_HFLIP, float(self.image_size.w), 0.0)
        elif method == FLIP_TOP_BOTTOM:
            op = (OP_VFLIP, 0.0, float(self.image_size.h))
        else:
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        # BoxList-valued fields are flipped too, each inside ITS OWN image (:196-199)
        return _BoxTrace(self.image_size, self.ops + (op,), {k: v.transpose(method) for k, v in self.extra_fields.items()})

    def crop(self, box):                                # reference bounding_box.py:202-226
        size = FeatureMapSize(w=box[2] - box[0], h=box[3] - box[1])
        return _BoxTrace(size, self.ops + ((OP_SHIFT, float(box[0]), float(box[1])),),
                         {k: v.crop(box) for k, v in self.extra_fields.items()})


def apply_box_ops(boxes, ops):
    """The recorded chain on a [n,4] xyxy tensor with the arithmetic of the BoxList methods (one float32 rounding per
    product / difference) - what os2d_apply_box_ops does on the device."""
    x1, y1, x2, y2 = boxes.unbind(1)
    for kind, ax, ay in ops:
        if kind == OP_SCALE:
            x1, y1, x2, y2 = x1 * ax, y1 * ay, x2 * ax, y2 * ay
        elif kind == OP_HFLIP:
            x1, x2 = ax - x2, ax - x1
        elif kind == OP_VFLIP:
            y1, y2 = ay - y2, ay - y1
        else:
            x1, y1, x2, y2 = x1 - ax, y1 - ay, x2 - ax, y2 - ay
    return torch.stack([x1, y1, x2, y2], dim=1)


_PROBE_BOXES = ((0.0, 0.0, 1.0, 1.0), (13.25, 7.5, 211.0, 95.75), (3.0, 250.5, 640.125, 479.0))


def transform_level_boxes(transform, boxes, default_boxes, corners, img_size):
    """What reference box_coder.py:509-521 does with a level's ``inverse_box_transforms`` entry, on plain tensors:
    boxes [n,4] -> transform(boxes); the anchors ride along as the BoxList FIELD "default_boxes" (BoxList.transpose / crop
    transform BoxList-valued fields as well, each inside its own image; resize copies them as they are) and the transform is
    applied to that field once more afterwards; the corners [m,4] go through as boxes of their own.  For a resize-only chain
    (the evaluation's) all three see the same map; with flips / crops the anchors see them twice - reproduced as is, so that
    a caller of the reference gets the reference's numbers.  Returns (boxes, default_boxes, corners or None, output size)."""
    bl = BoxList(boxes, img_size)
    bl.add_field("default_boxes", BoxList(default_boxes, img_size))
    out = transform(bl)
    dflt = transform(out.get_field("default_boxes"))
    cor = transform(BoxList(corners, img_size)).bbox_xyxy if corners is not None else None
    return out.bbox_xyxy, dflt.bbox_xyxy, cor, out.image_size


def trace_box_transform(transform, img_size):
    """-> (box ops, anchor ops, output image size) of one ``inverse_box_transforms`` entry applied to a level on an image of
    ``img_size`` (see ``transform_level_boxes`` for why the anchors have a chain of their own), or None when the entry is not a
    chain of BoxList.resize / transpose / crop - then only the generic decode can run it.  ``None`` entries trace to empty
    chains.  The recorded chains are checked against the entry itself on three probe boxes (unit, generic, near the border):
    bit-equal coordinates and the same output size, or they are not used."""
    if transform is None:
        return (), (), img_size
    # Two steps: the TRACE (the closures run on _BoxTrace objects: no tensors, a few Python calls) is made on every call, the
    # PROBE CHECK (three closure runs on CPU BoxLists) only once per (callable, image size, traced chains).  Keyed on the chains
    # themselves, a hit cannot be stale: a TransformList that was appended to (whatever the attribute its list lives in - the
    # reference keeps it in ``_transforms``, structures/transforms.py:18-22), or a closure whose captured state changed, traces to
    # other chains and is probed again (round 5 keyed on object identity + an attribute named ``transforms`` only).
    # The cache holds WEAK references: it keeps no transform (nor what it captured) alive, a dead entry's id may be reused and
    # is then not a hit; callables that cannot be weakly referenced are not cached.
    traced = _trace_chains(transform, img_size)
    if traced is None:
        return None
    try:
        ref = weakref.ref(transform)
    except TypeError:
        return traced if _probe_check(transform, img_size, traced) else None
    ckey = (id(transform), int(img_size.w), int(img_size.h), traced[0], traced[1], int(traced[2].w), int(traced[2].h))
    hit = _TRACE_CACHE.get(ckey)
    if hit is not None and hit[0]() is transform:
        _TRACE_CACHE.move_to_end(ckey)
        return traced if hit[1] else None
    ok = _probe_check(transform, img_size, traced)
    _TRACE_CACHE[ckey] = (ref, ok)
    while len(_TRACE_CACHE) > 64:
        _TRACE_CACHE.popitem(last=False)
    return traced if ok else None


_TRACE_CACHE = collections.OrderedDict()


def _trace_chains(transform, img_size):
    """(box ops, anchor ops, output size) recorded by running the entry on _BoxTrace objects, or None."""
    try:
        root = _BoxTrace(img_size)
        root.add_field("default_boxes", _BoxTrace(img_size))
        traced = transform(root)
        if not isinstance(traced, _BoxTrace) or len(traced.ops) > MAX_BOX_OPS:
            return None
        anchors = transform(traced.get_field("default_boxes"))
        if not isinstance(anchors, _BoxTrace) or len(anchors.ops) > MAX_DEFAULT_BOX_OPS:
            return None
    except Exception:   # noqa: BLE001 - a closure that does anything else than the three BoxList operations
        return None
    return traced.ops, anchors.ops, traced.image_size


def _probe_check(transform, img_size, traced):
    """The recorded chains against the entry itself on the probe boxes: bit-equal coordinates and the same output size."""
    try:
        probe = torch.tensor(_PROBE_BOXES, dtype=torch.float32)
        b, d, c, size = transform_level_boxes(transform, probe.clone(), probe.clone() + 0.5, probe.clone(), img_size)
        return bool(size == traced[2] and torch.equal(b, apply_box_ops(probe, traced[0])) and torch.equal(c, b)
                    and torch.equal(d, apply_box_ops(probe + 0.5, traced[1])))
    except Exception:   # noqa: BLE001
        return False


def _trace_box_transform(transform, img_size):
    """Uncached trace + probe check (tests)."""
    traced = _trace_chains(transform, img_size)
    return traced if traced is not None and _probe_check(transform, img_size, traced) else None


def _ops_tables(ops_per_level, width=MAX_BOX_OPS):
    """ctypes tables (counts [L], kinds [L][width], args [L][width][2]) of os2d_detect_level_ops / os2d_detect_pyramid_ops."""
    L = len(ops_per_level)
    counts = (ctypes.c_int * L)(*[len(o) for o in ops_per_level])
    kinds = (ctypes.c_int * (L * width))()
    args = (ctypes.c_float * (L * width * 2))()
    for l, ops in enumerate(ops_per_level):
        for k, (kind, ax, ay) in enumerate(ops):
            kinds[l * width + k] = kind
            args[(l * width + k) * 2] = ax
            args[(l * width + k) * 2 + 1] = ay
    return counts, kinds, args


class BoxGridGenerator(object):
    """reference box_coder.py:62-76: binds a box size and stride (both FeatureMapSize)."""

    def __init__(self, box_size, box_stride):
        self.box_size = box_size
        self.box_stride = box_stride

    def create_strided_boxes_columnfirst(self, fm_size):
        return create_strided_boxes_columnfirst(fm_size, self.box_size, self.box_stride)


def feature_map_size_c4(img_size):
    """Closed form of the ResNet-C4 output size (conv s2, maxpool s2, layer2 s2, layer3 s2, all with
    'same-ish' padding): four ceil-halvings.  Replaces the reference's dummy forward pass (model.py:91-114)."""
    def f(s):
        for _ in range(4):
            s = (s + 1) // 2
        return s
    return FeatureMapSize(w=f(img_size.w), h=f(img_size.h))


class Os2dBoxCoder(object):
    """Inference half of reference box_coder.py:169-536."""

    def __init__(self, positive_iou_threshold=0.5, negative_iou_threshold=0.1,
                 remap_classification_targets_iou_pos=0.5, remap_classification_targets_iou_neg=0.1,
                 output_box_grid_generator=None, function_get_feature_map_size=None, do_nms_across_classes=False):
        self.get_feature_map_size = function_get_feature_map_size or feature_map_size_c4
        self.output_box_grid_generator = output_box_grid_generator
        self.positive_iou_threshold = positive_iou_threshold
        self.negative_iou_threshold = negative_iou_threshold
        self.remap_classification_targets_iou_pos = remap_classification_targets_iou_pos
        self.remap_classification_targets_iou_neg = remap_classification_targets_iou_neg
        self.do_nms_across_classes = do_nms_across_classes
        self.nms_max_batch = 10000          # reference bounding_box.py:344 (``nms_max_batch_size``)
        self.use_fused_level_kernel = True  # single-level calls go through os2d_detect_level when the level fits,
                                            # several levels through os2d_detect_pyramid
        self.fused_pyramid_passes = 3       # chunked-NMS passes launched by os2d_detect_pyramid (more -> generic path)
        self.weights = BOX_ENCODING_WEIGHTS
        g = output_box_grid_generator
        if g is None:
            raise RuntimeError("output_box_grid_generator is required (head_creator.box_grid_generator_image_level)")
        if g.box_size.w != g.box_size.h or g.box_stride.w != g.box_stride.h:
            raise RuntimeError("anisotropic anchors are not supported")
        self._stride = int(g.box_stride.w)
        self._rec_field = int(g.box_size.w - self._stride * 14)

    def _get_default_boxes(self, img_size):
        """reference box_coder.py:191-203 (CPU anchors as a BoxList; the kernels use the closed form)."""
        fm = self.get_feature_map_size(img_size)
        return BoxList(self.output_box_grid_generator.create_strided_boxes_columnfirst(fm), image_size=img_size)

    def decode_level(self, loc_scores, img_size):
        """loc_scores [NB,4,HW] device tensor -> boxes [NB,HW,4] xyxy clipped to the level image
        (reference box_coder.py:319-330 + bounding_box.py:261-265) with os2d_decode_boxes."""
        lib = _lib.load()
        if not (loc_scores.is_cuda and loc_scores.dtype == torch.float32):
            raise RuntimeError("decode runs on the HIP device only (no CPU fallback)")
        loc_scores = loc_scores.contiguous()
        fm = self.get_feature_map_size(img_size)
        NB, four, HW = loc_scores.shape
        assert four == 4 and HW == fm.w * fm.h, "loc_scores {} do not match the feature map {}".format(tuple(loc_scores.shape), fm)
        boxes = torch.empty(NB, HW, 4, dtype=torch.float32, device=loc_scores.device)
        with torch.cuda.device(loc_scores.device):       # launches act on the CURRENT device
            _lib.check(lib.os2d_decode_boxes(_lib.ptr(loc_scores), NB, fm.h, fm.w, self._stride, self._rec_field,
                                             ctypes.c_float(img_size.w), ctypes.c_float(img_size.h), _lib.ptr(boxes),
                                             _lib.current_stream(loc_scores.device)), "os2d_decode_boxes")
        return boxes

    def build_boxes_from_loc_scores(self, loc_scores, default_boxes):
        """reference box_coder.py:319-330 for ONE class: loc_scores [HW,4], default_boxes a BoxList carrying the
        level's image size.  Returns an (unclipped-to-nothing-smaller) BoxList like the reference."""
        img_size = default_boxes.image_size
        boxes = self.decode_level(loc_scores.t().contiguous().unsqueeze(0), img_size)[0]
        return BoxList(boxes, image_size=img_size, mode="xyxy")

    # ------------------------------------------------------------------ NMS (os2d_nms)
    @staticmethod
    def nms_sorted(boxes_sorted, counts, iou_threshold):
        """boxes_sorted [NC,N,4] (each list by decreasing score, counts[c] valid) -> keep mask [NC,N] bool."""
        lib = _lib.load()
        if not boxes_sorted.is_cuda:
            raise RuntimeError("NMS runs on the HIP device only (no CPU fallback)")
        boxes_sorted = boxes_sorted.contiguous().float()
        NC, N, _ = boxes_sorted.shape
        dev = boxes_sorted.device
        counts = counts.to(device=dev, dtype=torch.int32).contiguous()
        keep = torch.empty(NC, N, dtype=torch.uint8, device=dev)
        num_keep = torch.empty(NC, dtype=torch.int32, device=dev)
        nbytes = ctypes.c_size_t()
        _lib.check(lib.os2d_nms_workspace_bytes(NC, N, ctypes.byref(nbytes)), "os2d_nms_workspace_bytes")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.os2d_nms(_lib.ptr(boxes_sorted), _lib.ptr(counts), NC, N, ctypes.c_float(iou_threshold),
                                    _lib.ptr(keep), _lib.ptr(num_keep), _lib.ptr(ws), ws.numel(),
                                    _lib.current_stream(dev)), "os2d_nms")
        return keep.bool()

    def _nms_lists(self, boxes, scores, valid, iou_threshold, nms_max_batch=10000):
        """Batched equivalent of reference bounding_box.py:343-374 for NC independent lists padded to a common
        length: boxes [NC,N,4], scores [NC,N], valid [NC,N] bool.  Lists longer than ``nms_max_batch`` go through
        the reference's chunk-and-repeat scheme: the current id list (at first the valid entries in list order) is cut
        into consecutive chunks of ``nms_max_batch``, every chunk is NMS-ed on its own, the survivors are concatenated
        chunk by chunk in score order, until a list needed one chunk only or lost nothing.  Per pass: ONE stable sort
        by (chunk, descending score), ONE gather and ONE ``os2d_nms`` launch over all NC x chunks lists, one host
        synchronisation.  Lists that are finished are stable under further passes (an NMS of its own output removes
        nothing), so all lists simply iterate until the last one is done.  Returns a bool keep mask [NC,N]."""
        NC, N = scores.shape
        dev = scores.device
        M = int(nms_max_batch)
        # descending-score sort key as a non-negative int64 < 2^32 (smaller = higher score; -0 and +0 tie like in a float sort)
        bits = (scores.float() + 0.0).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        asc = torch.where((bits >> 31) == 1, bits ^ 0xFFFFFFFF, bits | 0x80000000)
        desc_key = 0xFFFFFFFF - asc
        cnt = valid.sum(1)
        n_max = int(cnt.max())
        if n_max == 0:
            return torch.zeros_like(valid)
        ids = torch.argsort((~valid).to(torch.uint8), dim=1, stable=True)[:, :n_max]      # valid entries, list order
        while True:
            L = ids.size(1)
            num_chunks = (n_max + M - 1) // M
            pos = torch.arange(L, device=dev).unsqueeze(0)
            inlist = pos < cnt.unsqueeze(1)
            key = torch.where(inlist, ((pos // M) << 32) | torch.gather(desc_key, 1, ids),
                              torch.full((1, 1), (num_chunks + 1) << 32, dtype=torch.int64, device=dev))
            order = torch.argsort(key, dim=1, stable=True)
            ids_sorted = torch.gather(ids, 1, order)          # chunk ch of a list = sorted positions [ch*M, ch*M + its count)
            Lp = num_chunks * M
            padded = ids_sorted if L == Lp else torch.cat(
                [ids_sorted[:, :Lp], torch.zeros(NC, max(Lp - L, 0), dtype=ids.dtype, device=dev)], 1)
            b_sorted = torch.gather(boxes, 1, padded.unsqueeze(-1).expand(-1, -1, 4)).view(NC * num_chunks, M, 4)
            cnt_ch = (cnt.unsqueeze(1) - torch.arange(num_chunks, device=dev).unsqueeze(0) * M).clamp(0, M).reshape(-1)
            keep_sorted = self.nms_sorted(b_sorted, cnt_ch, iou_threshold).view(NC, Lp)
            keep_sorted = keep_sorted[:, :L] if Lp >= L else torch.cat(
                [keep_sorted, torch.zeros(NC, L - Lp, dtype=torch.bool, device=dev)], 1)
            new_cnt = keep_sorted.sum(1)
            compact = torch.argsort((~keep_sorted).to(torch.uint8), dim=1, stable=True)   # survivors first, arrangement kept
            ids = torch.gather(ids_sorted, 1, compact)
            more = (((cnt + M - 1) // M > 1) & (new_cnt != cnt)).any()
            new_max, more = torch.stack([new_cnt.max(), more.to(new_cnt.dtype)]).tolist()  # the pass's one synchronisation
            cnt, n_max = new_cnt, int(new_max)
            ids = ids[:, :max(n_max, 1)]
            if not more:
                break
        alive = torch.zeros(NC, N + 1, dtype=torch.bool, device=dev)
        pos = torch.arange(ids.size(1), device=dev).unsqueeze(0)
        alive.scatter_(1, torch.where(pos < cnt.unsqueeze(1), ids, torch.full_like(ids, N)), True)
        return alive[:, :N]

    def decode_pyramid(self, loc_scores_pyramid, cls_scores_pyramid, img_size_pyramid, class_ids,
                       nms_score_threshold=0.0, nms_iou_threshold=0.3, inverse_box_transforms=None,
                       transform_corners_pyramid=None):
        """reference box_coder.py:448-536 on the device, batched over classes:
        per level decode + clip (os2d_decode_boxes), score / empty-box mask, map to the original image
        (``inverse_box_transforms[l]`` is applied ONCE to a BoxList holding the level's boxes of all classes),
        concatenate levels, per-label NMS (os2d_nms), sort by score.  Entries of ``class_ids`` with the same id
        are merged before NMS, labels appear in the iteration order of ``set(class_ids)`` like in the reference (ascending
        for small ids).
        Returns a BoxList with fields scores, labels, default_boxes (and transform_corners if given)."""
        num_classes = len(class_ids)
        dev = cls_scores_pyramid[0].device
        if inverse_box_transforms is None:
            # the reference concatenates the levels of a label with cat_boxlist, which asserts one common image size
            # (bounding_box.py:390-437): levels of different sizes need transforms into a common frame
            if len({(s_.w, s_.h) for s_ in img_size_pyramid}) > 1:     # (not an ``assert``: must survive ``python -O``)
                raise ValueError("pyramid levels live on different image sizes ({}): pass inverse_box_transforms to map them to "
                                 "one image size before they are merged".format(sorted({(s_.w, s_.h) for s_ in img_size_pyramid})))
        fused = self._decode_single_level_fused(loc_scores_pyramid, cls_scores_pyramid, img_size_pyramid, class_ids,
                                                nms_score_threshold, nms_iou_threshold, inverse_box_transforms,
                                                transform_corners_pyramid)
        if fused is None:
            fused = self._decode_pyramid_fused(loc_scores_pyramid, cls_scores_pyramid, img_size_pyramid, class_ids,
                                               nms_score_threshold, nms_iou_threshold, inverse_box_transforms,
                                               transform_corners_pyramid)
        if fused is not None:
            return self._nms_across_classes(fused, nms_iou_threshold)
        boxes_l, scores_l, valid_l, dflt_l, corners_l = [], [], [], [], []
        for lvl, (loc, cls, img_size) in enumerate(zip(loc_scores_pyramid, cls_scores_pyramid, img_size_pyramid)):
            assert loc.device == dev and cls.device == dev, "scores and boxes should be on the same device"
            boxes = self.decode_level(loc, img_size)                                    # [B,HW,4]
            cls = cls.float()
            empty = (boxes[..., 3] <= boxes[..., 1]) | (boxes[..., 2] <= boxes[..., 0])
            valid = (cls > nms_score_threshold) & ~empty
            HW = boxes.size(1)
            dflt = self._get_default_boxes(img_size).bbox_xyxy.to(dev)                   # [HW,4]
            corners = None
            if transform_corners_pyramid is not None:
                corners = transform_corners_pyramid[lvl].transpose(1, 2).reshape(num_classes * HW * 2, 4)
            if inverse_box_transforms is not None:
                boxes, dflt, corners, _ = transform_level_boxes(inverse_box_transforms[lvl], boxes.reshape(-1, 4), dflt, corners, img_size)
                boxes = boxes.reshape(num_classes, HW, 4)
            boxes_l.append(boxes)
            scores_l.append(cls)
            valid_l.append(valid)
            dflt_l.append(dflt.unsqueeze(0).expand(num_classes, HW, 4))
            if corners is not None:
                corners_l.append(corners.reshape(num_classes, HW, 8))
        boxes = torch.cat(boxes_l, 1)
        scores = torch.cat(scores_l, 1)
        valid = torch.cat(valid_l, 1)
        dflt = torch.cat(dflt_l, 1)
        corners = torch.cat(corners_l, 1) if corners_l else None
        # merge rows that share a real label
        ids = [int(c) for c in class_ids]
        by_label = {}
        for i, c in enumerate(ids):
            by_label.setdefault(c, []).append(i)
        labels_sorted = list(set(ids))   # the reference iterates ``set(class_ids)`` (box_coder.py:483): same order, whatever it is
        groups = [by_label[l] for l in labels_sorted]
        if len(labels_sorted) != len(ids) or labels_sorted != ids:
            width = max(len(g) for g in groups) * boxes.size(1)

            def merge(t, fill):
                out = []
                for g in groups:
                    m = torch.cat([t[i] for i in g], 0)
                    pad = width - m.size(0)
                    if pad:
                        m = torch.cat([m, torch.full((pad,) + tuple(m.shape[1:]), fill, dtype=m.dtype, device=dev)], 0)
                    out.append(m)
                return torch.stack(out, 0)
            boxes, scores, valid, dflt = merge(boxes, 0.0), merge(scores, float("-inf")), merge(valid, False), merge(dflt, 0.0)
            corners = merge(corners, 0.0) if corners is not None else None
        if boxes.size(1) <= self.nms_max_batch:
            # every list fits one NMS batch (always true for a single level): ONE stable sort by score, NMS on the sorted
            # lists, and the survivors are already in the output order - no host synchronisation before the final
            # ``nonzero`` that sizes the result
            key = torch.where(valid, scores, torch.full_like(scores, float("-inf")))
            order = torch.argsort(key, dim=1, descending=True, stable=True)
            b_sorted = torch.gather(boxes, 1, order.unsqueeze(-1).expand(-1, -1, 4))
            keep_sorted = self.nms_sorted(b_sorted, valid.sum(1), nms_iou_threshold)
        else:
            keep = self._nms_lists(boxes, scores, valid, nms_iou_threshold, self.nms_max_batch)
            # per label: survivors by decreasing score (box_coder.py:431-437)
            key = torch.where(keep, scores, torch.full_like(scores, float("-inf")))
            order = torch.argsort(key, dim=1, descending=True, stable=True)
            keep_sorted = torch.gather(keep, 1, order)
        sel = keep_sorted.reshape(-1).nonzero().squeeze(1)
        flat = (order + torch.arange(order.size(0), device=dev).unsqueeze(1) * order.size(1)).reshape(-1)[sel]
        out_size = img_size_pyramid[0]
        if inverse_box_transforms is not None:
            out_size = inverse_box_transforms[0](BoxList(torch.zeros(1, 4, device=dev), img_size_pyramid[0])).image_size
        result = BoxList(boxes.reshape(-1, 4)[flat], out_size)
        result.add_field("scores", scores.reshape(-1)[flat])
        lab = torch.tensor(labels_sorted, dtype=torch.long, device=dev).unsqueeze(1).expand(-1, order.size(1)).reshape(-1)
        result.add_field("labels", lab[sel])
        result.add_field("default_boxes", BoxList(dflt.reshape(-1, 4)[flat], out_size))
        if corners is not None:
            result.add_field("transform_corners", corners.reshape(-1, 8)[flat])
        return self._nms_across_classes(result, nms_iou_threshold)

    def _nms_across_classes(self, result, nms_iou_threshold):
        """reference box_coder.py:530-534 (``eval.nms_across_classes``, off by default)."""
        if self.do_nms_across_classes and len(result) > 0:
            b = result.bbox_xyxy.unsqueeze(0)
            s = result.get_field("scores").unsqueeze(0)
            k = self._nms_lists(b, s, torch.ones_like(s, dtype=torch.bool), nms_iou_threshold)[0]
            idx = k.nonzero().squeeze(1)
            idx = idx[torch.argsort(s[0][idx], descending=True, stable=True)]
            result = result[idx]
        return result

    def _decode_single_level_fused(self, loc_pyr, cls_pyr, size_pyr, class_ids, score_thr, iou_thr, inverse, corners_pyr):
        """One level, one row per label, an identity / ``ResizeBoxes`` mapping, and a level that fits the kernel's LDS:
        the whole per-class chain runs in ONE launch (os2d_detect_level); None when the generic path has to be used."""
        if not self.use_fused_level_kernel or len(loc_pyr) != 1:
            return None
        ids = [int(c) for c in class_ids]
        if len(set(ids)) != len(ids):
            return None
        loc, cls, img_size = loc_pyr[0], cls_pyr[0], size_pyr[0]
        traced = trace_box_transform(inverse[0] if inverse is not None else None, img_size)
        if traced is None:
            return None
        ops, default_ops, out_size = traced
        fm = self.get_feature_map_size(img_size)
        lib = _lib.load()
        if not lib.os2d_detect_level_supported(fm.h, fm.w):
            return None
        if not (loc.is_cuda and cls.is_cuda and loc.dtype == torch.float32):
            raise RuntimeError("decode runs on the HIP device only (no CPU fallback)")
        dev = cls.device
        B, HW = len(ids), fm.h * fm.w
        loc = loc.contiguous()
        cls = cls.float().contiguous()
        assert tuple(loc.shape) == (B, 4, HW) and tuple(cls.shape) == (B, HW), "level tensors do not match class_ids / feature map"
        counts_c, kinds_c, args_c = _ops_tables([ops])
        out_boxes = torch.empty(B, HW, 4, dtype=torch.float32, device=dev)
        out_scores = torch.empty(B, HW, dtype=torch.float32, device=dev)
        out_index = torch.empty(B, HW, dtype=torch.int32, device=dev)
        out_count = torch.empty(B, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):     # hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device
            _lib.check(lib.os2d_detect_level_ops(_lib.ptr(loc), _lib.ptr(cls), B, fm.h, fm.w, self._stride, self._rec_field,
                                                 ctypes.c_float(img_size.w), ctypes.c_float(img_size.h), len(ops), kinds_c, args_c,
                                                 ctypes.c_float(score_thr), ctypes.c_float(iou_thr),
                                                 _lib.ptr(out_boxes), _lib.ptr(out_scores), _lib.ptr(out_index),
                                                 _lib.ptr(out_count), _lib.current_stream(dev)), "os2d_detect_level_ops")
        # rows in the label order of the reference (iteration order of ``set(class_ids)``), survivors of a row by score
        rank = {l: k for k, l in enumerate(set(ids))}      # the reference iterates ``set(class_ids)`` (box_coder.py:483)
        order = sorted(range(B), key=lambda i: rank[ids[i]])
        counts = out_count
        if order != list(range(B)):
            perm = torch.tensor(order, dtype=torch.long, device=dev)
            counts = out_count[perm]
        else:
            perm = None
        mask = torch.arange(HW, device=dev).unsqueeze(0) < counts.unsqueeze(1)
        row, pos = mask.nonzero(as_tuple=True)                   # the one host synchronisation: sizes the result
        src_row = perm[row] if perm is not None else row
        flat = src_row * HW + pos
        result = BoxList(out_boxes.view(-1, 4)[flat], out_size)
        result.add_field("scores", out_scores.view(-1)[flat])
        result.add_field("labels", torch.tensor([ids[i] for i in order], dtype=torch.long, device=dev)[row])
        loc_idx = out_index.view(-1)[flat].long()
        dflt = apply_box_ops(self._get_default_boxes(img_size).bbox_xyxy.to(dev)[loc_idx], default_ops)
        result.add_field("default_boxes", BoxList(dflt, out_size))
        if corners_pyr is not None:
            corners = corners_pyr[0][src_row, :, loc_idx]                                   # [n, 8]
            # like the reference, the corners go through the level's transform as the two "boxes" (x0, y0, x1, y1), (x2, y2, x3, y3)
            result.add_field("transform_corners", apply_box_ops(corners.reshape(-1, 4), ops).view(-1, 8))
        return result

    def _slot_rows(self, ids, labels, V, dev):
        """Device table [G][V] of the head rows of every label (-1: no such view), cached per (class ids, device): the rows of
        an evaluation's class batch are the same for every image, and building the table is a blocking host-to-device copy
        from pageable memory on the per-image path."""
        cache = self.__dict__.setdefault("_slot_rows_cache", {})
        key = (ids, V, str(dev))
        stream = torch.cuda.current_stream(dev)
        entry = cache.get(key)
        if entry is None:
            rows_of = {}
            for i, c in enumerate(ids):
                rows_of.setdefault(c, []).append(i)
            host = torch.tensor([(rows_of[l] + [-1] * V)[:V] for l in labels], dtype=torch.int32).pin_memory()
            table = host.to(dev, non_blocking=True)
            event = torch.cuda.Event()
            event.record(stream)
            if len(cache) >= 16:
                cache.pop(next(iter(cache)))
            # the pinned source lives as long as the copy may be in flight; the event marks the end of the copy
            entry = cache[key] = (table, host, event, stream.cuda_stream)
        table, _, event, producer = entry
        if stream.cuda_stream != producer:
            # a decode on ANOTHER stream than the one that filled the table: wait for the copy, and tell the caching
            # allocator that this stream reads the tensor too - an evicted entry must not be reused under a running kernel
            stream.wait_event(event)
            table.record_stream(stream)
        return table

    def _decode_pyramid_fused(self, loc_pyr, cls_pyr, size_pyr, class_ids, score_thr, iou_thr, inverse, corners_pyr):
        """Several levels and / or merged labels (class-image views: several head rows with one class id, reference
        box_coder.py:483-487), identity / ``ResizeBoxes`` mappings: the whole per-label chain, incl. the reference's
        chunk-and-repeat NMS for lists longer than ``nms_max_batch``, runs on the device (os2d_detect_pyramid /
        os2d_detect_pyramid_merged: one launch per NMS pass, a work-group per (chunk, label)); None when the generic path has
        to be used."""
        ids = [int(c) for c in class_ids]
        merged = len(set(ids)) != len(ids)
        if not self.use_fused_level_kernel:
            return None
        # (a single level with one row per label normally took os2d_detect_level before this is called; levels beyond that
        # kernel's LDS budget - more than ~5,900 locations, e.g. 72 x 96 - come here as a pyramid of one level)
        ts = list(inverse) if inverse is not None else [None] * len(loc_pyr)
        traced = [trace_box_transform(t, s_) for t, s_ in zip(ts, size_pyr)]
        if any(t is None for t in traced):
            return None      # an entry that is not a chain of BoxList.resize / transpose / crop: generic path
        if len({(sz.w, sz.h) for _, _, sz in traced}) != 1:
            return None      # levels that end on different image sizes: the generic path fails like the reference's cat_boxlist
        lib = _lib.load()
        L = len(loc_pyr)
        fms = [self.get_feature_map_size(s) for s in size_pyr]
        hws = [fm.h * fm.w for fm in fms]
        N1 = sum(hws)                                         # candidates per head row
        # labels in the reference's iteration order (``set(class_ids)``, box_coder.py:483), each with its rows in row order
        labels = list(set(ids))
        rows_of = {l: [i for i, c in enumerate(ids) if c == l] for l in labels}
        G, V = len(labels), max(len(r) for r in rows_of.values())
        N = N1 * V                                            # candidates per label
        if N > (1 << 22) or not lib.os2d_detect_pyramid_supported(L, N, int(self.nms_max_batch)):
            return None
        dev = cls_pyr[0].device
        B = len(ids)
        locs, clss = [], []
        for loc, cls, hw in zip(loc_pyr, cls_pyr, hws):
            if not (loc.is_cuda and cls.is_cuda and loc.dtype == torch.float32):
                raise RuntimeError("decode runs on the HIP device only (no CPU fallback)")
            assert loc.device == dev and cls.device == dev, "scores and boxes should be on the same device"
            loc, cls = loc.contiguous(), cls.float().contiguous()
            assert tuple(loc.shape) == (B, 4, hw) and tuple(cls.shape) == (B, hw), "level tensors do not match class_ids / feature map"
            locs.append(loc)
            clss.append(cls)
        out_size = traced[0][2]
        c_counts, c_kinds, c_args = _ops_tables([t_[0] for t_ in traced])
        d_counts, d_kinds, d_args = _ops_tables([t_[1] for t_ in traced], MAX_DEFAULT_BOX_OPS)
        passes = int(self.fused_pyramid_passes)
        nbytes = ctypes.c_size_t()
        _lib.check(lib.os2d_detect_pyramid_workspace_bytes(G, N, passes, ctypes.byref(nbytes)), "os2d_detect_pyramid_workspace_bytes")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        out_boxes = torch.empty(G, N, 4, dtype=torch.float32, device=dev)
        out_scores = torch.empty(G, N, dtype=torch.float32, device=dev)
        out_index = torch.empty(G, N, dtype=torch.int32, device=dev)
        out_count = torch.empty(G, dtype=torch.int32, device=dev)
        out_default = torch.empty(G, N, 4, dtype=torch.float32, device=dev)
        unfinished = torch.empty(1, dtype=torch.int32, device=dev)
        c_loc = (ctypes.c_void_p * L)(*[t.data_ptr() for t in locs])
        c_cls = (ctypes.c_void_p * L)(*[t.data_ptr() for t in clss])
        c_cor, out_corners, cors = None, None, None
        if corners_pyr is not None:
            cors = [k.reshape(B, 8, hw).float().contiguous() for k, hw in zip(corners_pyr, hws)]
            c_cor = (ctypes.c_void_p * L)(*[t.data_ptr() for t in cors])
            out_corners = torch.empty(G, N, 8, dtype=torch.float32, device=dev)
        c_hw = (ctypes.c_int * (2 * L))(*[v for fm in fms for v in (fm.h, fm.w)])
        c_img = (ctypes.c_float * (2 * L))(*[float(v) for s_ in size_pyr for v in (s_.w, s_.h)])
        identity = not merged and labels == ids          # one row per label, already in the reference's label order
        with torch.cuda.device(dev):
            slot_rows = None if identity else self._slot_rows(tuple(ids), tuple(labels), V, dev)
            _lib.check(lib.os2d_detect_pyramid_ops(c_loc, c_cls, c_cor, B, L, c_hw, self._stride, self._rec_field, c_img,
                                                   c_counts, c_kinds, c_args, d_counts, d_kinds, d_args,
                                                   ctypes.c_float(score_thr), ctypes.c_float(iou_thr),
                                                   int(self.nms_max_batch), passes, G, V, _lib.ptr(slot_rows),
                                                   _lib.ptr(out_boxes), _lib.ptr(out_scores), _lib.ptr(out_index),
                                                   _lib.ptr(out_default), _lib.ptr(out_corners), _lib.ptr(out_count),
                                                   _lib.ptr(unfinished), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)),
                       "os2d_detect_pyramid_ops")
        mask = torch.arange(N, device=dev).unsqueeze(0) < out_count.unsqueeze(1)
        row, pos = mask.nonzero(as_tuple=True)                   # the one host synchronisation: sizes the result
        if int(unfinished.item()) != 0:
            return None          # some label needs more NMS passes than were launched: generic path (exact, slower)
        flat = row * N + pos
        result = BoxList(out_boxes.view(-1, 4)[flat], out_size)
        result.add_field("scores", out_scores.view(-1)[flat])
        result.add_field("labels", torch.tensor(labels, dtype=torch.long, device=dev)[row])
        result.add_field("default_boxes", BoxList(out_default.view(-1, 4)[flat], out_size))
        if out_corners is not None:
            result.add_field("transform_corners", out_corners.view(-1, 8)[flat])
        return result
