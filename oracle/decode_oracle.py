"""CPU ORACLE for the per-location box decode + per-class NMS that follows the head.  TEST INFRASTRUCTURE ONLY
(same rules as ``head_oracle.py``: imported by tests / smoke / the cpu_baseline leg only).

Restates reference os2d/modeling/box_coder.py:319-330 (decode through torchvision's
``BoxCoder.decode_single``), :448-536 (``decode_pyramid``), :425-437 (``_nms_box_lists``) and
os2d/structures/bounding_box.py:261-281,344-387 (clip, empty-box mask, batched NMS wrapper).
torchvision's decode / clip / nms are restated from their published algorithms (pinned version
torchvision 0.5 per the reference's INSTALL.md:13): parity at that boundary is pinned to the formulas and
to the fixture ``tests/golden/decode_pyramid.npz`` produced by the reference's own ``decode_pyramid``.
"""
import math

import torch

from .head_oracle import LOC_WEIGHTS, TEMPLATE, anchor_grid

XFORM_CLIP = math.log(1000.0 / 16)    # torchvision BoxCoder default bbox_xform_clip


def decode_level(loc, H, W, img_w, img_h, stride=16, rec_field=16):
    """loc [B,4,H*W] -> boxes [B,H*W,4] (xyxy, clipped to the level's image).
    reference box_coder.py:319-330 + bounding_box.py:261-265."""
    box = float(stride * (TEMPLATE - 1) + rec_field)
    anchors = anchor_grid(H, W, box, float(stride))              # [HW,4]
    aw = anchors[:, 2] - anchors[:, 0]
    ah = anchors[:, 3] - anchors[:, 1]
    acx = anchors[:, 0] + 0.5 * aw
    acy = anchors[:, 1] + 0.5 * ah
    dx = loc[:, 0] / LOC_WEIGHTS[0]
    dy = loc[:, 1] / LOC_WEIGHTS[1]
    dw = torch.clamp(loc[:, 2] / LOC_WEIGHTS[2], max=XFORM_CLIP)
    dh = torch.clamp(loc[:, 3] / LOC_WEIGHTS[3], max=XFORM_CLIP)
    pcx = dx * aw + acx
    pcy = dy * ah + acy
    pw = torch.exp(dw) * aw
    ph = torch.exp(dh) * ah
    x1 = (pcx - 0.5 * pw).clamp(0, img_w)
    y1 = (pcy - 0.5 * ph).clamp(0, img_h)
    x2 = (pcx + 0.5 * pw).clamp(0, img_w)
    y2 = (pcy + 0.5 * ph).clamp(0, img_h)
    return torch.stack([x1, y1, x2, y2], dim=-1)


def box_iou_matrix(b):
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(b[:, None, :2], b[None, :, :2])
    rb = torch.min(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area[:, None] + area[None, :] - inter)


def greedy_nms(boxes, scores, iou_thr):
    """Standard greedy NMS (torchvision.ops.nms semantics): visit by decreasing score, drop boxes with
    IoU > iou_thr against a kept one; returns kept indices by decreasing score."""
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.long)
    order = torch.argsort(scores, descending=True, stable=True)
    iou = box_iou_matrix(boxes)
    dead = torch.zeros(boxes.size(0), dtype=torch.bool)
    keep = []
    for i in order.tolist():
        if not dead[i]:
            keep.append(i)
            dead |= iou[i] > iou_thr
    return torch.tensor(keep, dtype=torch.long)


def nms_chunked(boxes, scores, iou_thr, max_batch=10000, score_thr=float("-inf")):
    """reference bounding_box.py:343-374 (``nms`` with ``do_separate_per_label=False``): the memory-bounded NMS.
    Start from the ids with score > score_thr in list order; repeat { cut the current id list into consecutive chunks
    of ``max_batch``; greedy-NMS every chunk on its own (kept ids come back by decreasing score); concatenate the
    chunks' survivors in chunk order } until there was at most one chunk or no box was removed.  With more than one
    chunk this is NOT a global NMS: boxes of different chunks only meet if a later pass puts them in the same chunk.
    Returns the surviving ids in the reference's final order."""
    ids = torch.nonzero(scores > score_thr).squeeze(1)
    while True:
        n = ids.numel()
        num_batches = -(-n // max_batch)
        survived = []
        for start in range(0, n, max_batch):
            chunk = ids[start:start + max_batch]
            survived.append(chunk[greedy_nms(boxes[chunk], scores[chunk], iou_thr)])
        ids = torch.cat(survived, 0) if survived else torch.zeros(0, dtype=torch.long)
        if num_batches <= 1 or ids.numel() == n:
            return ids


def decode_pyramid(loc_pyramid, cls_pyramid, fm_sizes, img_sizes, orig_size=None,
                   score_thr=float("-inf"), iou_thr=0.3, nms_across_classes=False):
    """Per class: decode every level, clip, drop empty / low-score boxes, map to the original image
    (ratio resize, reference bounding_box.py:138-163), concatenate levels, NMS, sort by score.

    loc_pyramid[l] [B,4,HW_l], cls_pyramid[l] [B,HW_l], fm_sizes[l]=(H,W), img_sizes[l]=(w,h).
    Returns (boxes [N,4], scores [N], labels [N]) with classes in ascending order; with ``nms_across_classes``
    (reference box_coder.py:530-532, config ``eval.nms_across_classes``) a second NMS over the union of all labels,
    result by decreasing score.
    """
    B = cls_pyramid[0].size(0)
    out_b, out_s, out_l = [], [], []
    level_boxes = [decode_level(loc, H, W, iw, ih) for loc, (H, W), (iw, ih) in zip(loc_pyramid, fm_sizes, img_sizes)]
    for b in range(B):
        bb, ss = [], []
        for lvl, (boxes, (iw, ih)) in enumerate(zip(level_boxes, img_sizes)):
            bx = boxes[b]
            sc = cls_pyramid[lvl][b].float()
            empty = (bx[:, 3] <= bx[:, 1]) | (bx[:, 2] <= bx[:, 0])
            m = (sc > score_thr) & ~empty
            bx, sc = bx[m], sc[m]
            if orig_size is not None:
                rw = float(orig_size[0]) / iw
                rh = float(orig_size[1]) / ih
                if rw == rh:
                    bx = bx * rw
                else:
                    bx = bx * torch.tensor([rw, rh, rw, rh])
            bb.append(bx)
            ss.append(sc)
        bb, ss = torch.cat(bb, 0), torch.cat(ss, 0)
        keep = greedy_nms(bb, ss, iou_thr)
        out_b.append(bb[keep])
        out_s.append(ss[keep])
        out_l.append(torch.full((keep.numel(),), b, dtype=torch.long))
    bb, ss, ll = torch.cat(out_b, 0), torch.cat(out_s, 0), torch.cat(out_l, 0)
    if nms_across_classes:
        keep = greedy_nms(bb, ss, iou_thr)          # kept ids by decreasing score (box_coder.py:425-437)
        bb, ss, ll = bb[keep], ss[keep], ll[keep]
    return bb, ss, ll
