"""Minimal box container for the decode path, mirroring the parts of the reference's ``BoxList``
(os2d/structures/bounding_box.py:15-437) that ``Os2dBoxCoder.decode_pyramid`` and its callers touch:
xyxy storage, image size, named per-box fields, indexing, resize, clipping and the empty-box mask."""
import torch

from .feature_map import FeatureMapSize

# flip methods of ``BoxList.transpose`` (values of PIL.Image.FLIP_LEFT_RIGHT / FLIP_TOP_BOTTOM, reference bounding_box.py:11-12)
FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1


class BoxList(object):
    def __init__(self, bbox, image_size, mode="xyxy"):
        bbox = torch.as_tensor(bbox, dtype=torch.float32)
        if bbox.dim() != 2 or bbox.size(-1) != 4:
            raise ValueError("bbox should be of size [N, 4], got {}".format(tuple(bbox.shape)))
        if mode == "cx_cy_w_h":
            cx, cy, w, h = bbox.unbind(1)
            bbox = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], dim=1)
        elif mode != "xyxy":
            raise ValueError("mode should be 'xyxy' or 'cx_cy_w_h'")
        self.bbox_xyxy = bbox
        self.image_size = image_size
        self.extra_fields = {}

    # ---- fields (reference bounding_box.py:60-77)
    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def __len__(self):
        return self.bbox_xyxy.size(0)

    def __getitem__(self, item):
        sel = self.bbox_xyxy[item]
        if sel.dim() == 1:
            sel = sel.view(1, 4)
        out = BoxList(sel, self.image_size)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item])
        return out

    def to(self, device):
        out = BoxList(self.bbox_xyxy.to(device), self.image_size)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def cpu(self):
        return self.to("cpu")

    # ---- geometry
    def resize(self, target_size):
        """reference bounding_box.py:138-163: scale by target/current per axis."""
        rw = float(target_size.w) / self.image_size.w
        rh = float(target_size.h) / self.image_size.h
        if rw == rh:
            scaled = self.bbox_xyxy * rw
        else:
            scaled = self.bbox_xyxy * torch.tensor([rw, rh, rw, rh], dtype=torch.float32, device=self.bbox_xyxy.device)
        out = BoxList(scaled, target_size)
        out.extra_fields = dict(self.extra_fields)
        return out

    def transpose(self, method):
        """reference bounding_box.py:165-200: horizontal / vertical flip inside the image."""
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        x1, y1, x2, y2 = self.bbox_xyxy.unbind(1)
        if method == FLIP_LEFT_RIGHT:
            x1, x2 = self.image_size.w - x2, self.image_size.w - x1
        else:
            y1, y2 = self.image_size.h - y2, self.image_size.h - y1
        out = BoxList(torch.stack([x1, y1, x2, y2], dim=1), self.image_size)
        for k, v in self.extra_fields.items():
            out.add_field(k, v if isinstance(v, torch.Tensor) else v.transpose(method))
        return out

    def crop(self, box):
        """reference bounding_box.py:202-226: coordinates relative to the (left, upper, right, lower) region ``box``; the
        result lives on an image of the region's size and is NOT clipped to it."""
        x1, y1, x2, y2 = self.bbox_xyxy.unbind(1)
        out = BoxList(torch.stack([x1 - box[0], y1 - box[1], x2 - box[0], y2 - box[1]], dim=1),
                      FeatureMapSize(w=box[2] - box[0], h=box[3] - box[1]))
        for k, v in self.extra_fields.items():
            out.add_field(k, v if isinstance(v, torch.Tensor) else v.crop(box))
        return out

    def clip_to_image(self, remove_empty=True):
        """reference bounding_box.py:261-265."""
        b = self.bbox_xyxy
        self.bbox_xyxy = torch.stack([b[:, 0].clamp(0, self.image_size.w), b[:, 1].clamp(0, self.image_size.h),
                                      b[:, 2].clamp(0, self.image_size.w), b[:, 3].clamp(0, self.image_size.h)], dim=1)
        if remove_empty:
            return self[~self.get_mask_empty_boxes()]
        return self

    def get_mask_empty_boxes(self):
        """reference bounding_box.py:278-280."""
        b = self.bbox_xyxy
        return (b[:, 3] <= b[:, 1]) | (b[:, 2] <= b[:, 0])

    def __repr__(self):
        return "BoxList(num_boxes={}, image_size={})".format(len(self), self.image_size)


def cat_boxlist(bboxes):
    """reference bounding_box.py:390-437 (same image size, same field sets)."""
    if len(bboxes) == 0:
        raise ValueError("cannot concatenate an empty list")
    size = bboxes[0].image_size
    assert all(b.image_size == size for b in bboxes)
    out = BoxList(torch.cat([b.bbox_xyxy for b in bboxes], 0), size)
    for f in bboxes[0].fields():
        out.add_field(f, torch.cat([b.get_field(f) for b in bboxes], 0))
    return out
