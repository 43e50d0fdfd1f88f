"""Anchor grids and the decode side of the reference's box coder (os2d/modeling/box_coder.py) on the HIP library.

    create_strided_boxes_columnfirst / BoxGridGenerator   reference box_coder.py:16-76
    Os2dBoxCoder.build_boxes_from_loc_scores              reference box_coder.py:319-330   (os2d_decode_boxes)
    Os2dBoxCoder.decode_pyramid                           reference box_coder.py:448-536   (os2d_decode_boxes + os2d_nms)

Only the inference half of the reference class is mirrored: target encoding / anchor matching / hard-negative
remapping are training-only and out of scope.
"""
import ctypes
from functools import lru_cache

import torch

from .. import _lib
from ..structures.feature_map import FeatureMapSize
from ..structures.bounding_box import BoxList

BOX_ENCODING_WEIGHTS = (10.0, 10.0, 5.0, 5.0)   # reference box_coder.py:13


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
