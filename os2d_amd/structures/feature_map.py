"""Immutable (w, h) size value type, mirroring reference os2d/structures/feature_map.py:5-44
(same constructor forms, equality and hashing so it can key caches)."""
import torch


class FeatureMapSize(object):
    __slots__ = ("_w", "_h")

    def __init__(self, img=None, w=None, h=None):
        if w is None or h is None:
            if isinstance(img, torch.Tensor):
                w, h = img.size(-1), img.size(-2)
            elif hasattr(img, "size") and not callable(img.size) and len(img.size) == 2:  # PIL.Image: (w, h)
                w, h = img.size
            else:
                raise RuntimeError("Cannot initialize FeatureMapSize")
        object.__setattr__(self, "_w", int(w) if float(w).is_integer() else w)
        object.__setattr__(self, "_h", int(h) if float(h).is_integer() else h)

    w = property(lambda self: self._w)
    h = property(lambda self: self._h)

    def __setattr__(self, *args):
        raise AttributeError("Attributes of FeatureMapSize cannot be changed")

    def __delattr__(self, *args):
        raise AttributeError("Attributes of FeatureMapSize cannot be deleted")

    def __repr__(self):
        return "FeatureMapSize(w={}, h={})".format(self.w, self.h)

    def __eq__(self, other):
        return isinstance(other, FeatureMapSize) and (self.w, self.h) == (other.w, other.h)

    def __hash__(self):
        return hash((self.w, self.h))
