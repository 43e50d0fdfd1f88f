"""``Os2dModel`` with the reference's constructor, forward signatures, attribute names and state-dict keys
(reference os2d/modeling/model.py:72-426), driving the HIP head.

State-dict layout (unchanged, so ``{"net": ...}`` checkpoints of the reference load):
    net_feature_maps.*                                    backbone (PyTorch-ROCm)
    net_label_features.net_class_features.*               class-image backbone (shared when merge_branch_parameters)
    os2d_head_creator.aligner.parameter_regressor.{conv.0,conv.1,conv.3,conv.4,linear}.*   TransformNet
"""
import logging

import torch
import torch.nn as nn

from ..structures.feature_map import FeatureMapSize
from .box_coder import Os2dBoxCoder, feature_map_size_c4
from .feature_extractor import build_feature_extractor
from .head import build_os2d_head_creator


class LabelFeatureExtractor(nn.Module):
    """Backbone applied to a list of class images of different sizes (reference model.py:72-95)."""

    def __init__(self, feature_extractor):
        super(LabelFeatureExtractor, self).__init__()
        self.net_class_features = feature_extractor

    def forward(self, class_image_list):
        return [self.net_class_features(img.unsqueeze(0)) for img in class_image_list]

    def freeze_bn(self):
        self.net_class_features.freeze_bn()

    def freeze_blocks(self, num_blocks=0):
        self.net_class_features.freeze_blocks(num_blocks)


class Os2dModel(nn.Module):
    """The detector as a module (reference model.py:123-288): two ResNet-C4 branches (image / class images; one shared
    module when ``merge_branch_parameters``) and the head creator that owns the TransformNet.  Constructor arguments,
    attribute names and the state-dict layout are the reference's - they are the interface; the bodies are ours."""
    default_normalization = {"mean": (0.485, 0.456, 0.406), "std": (0.229, 0.224, 0.225)}

    def __init__(self, logger=None, is_cuda=False, merge_branch_parameters=False, use_group_norm=False,
                 backbone_arch="resnet50", use_inverse_geom_model=True, simplify_affine=False, img_normalization=None):
        super(Os2dModel, self).__init__()
        self.logger = logger if logger is not None else logging.getLogger("OS2D")
        # plain settings first ...
        self.is_cuda = bool(is_cuda)
        self.use_group_norm = use_group_norm
        self.merge_branch_parameters = merge_branch_parameters
        self.simplify_affine = simplify_affine
        self.use_inverse_geom_model = use_inverse_geom_model
        self.img_normalization = img_normalization or dict(self.default_normalization)
        # ... then the sub-modules, registered in the reference's order (it fixes the order of state_dict() keys, which
        # tests/test_checkpoints.py pins against the reference model): image branch, head creator, class-image branch
        image_branch = build_feature_extractor(backbone_arch, use_group_norm)
        class_branch = image_branch if merge_branch_parameters else build_feature_extractor(backbone_arch, use_group_norm)
        self.net_feature_maps = image_branch
        self.os2d_head_creator = build_os2d_head_creator(simplify_affine, self.is_cuda, use_inverse_geom_model,
                                                         image_branch.feature_map_stride,
                                                         image_branch.feature_map_receptive_field)
        self.net_label_features = LabelFeatureExtractor(feature_extractor=class_branch)
        self.eval()                         # inference is the only mode the HIP head implements
        if self.is_cuda:
            self.cuda()
        self.logger.info("OS2D model ({}, {} branches, {} transform{}) created on {}".format(
            backbone_arch, "shared" if merge_branch_parameters else "separate", "simplified affine" if simplify_affine else "affine",
            ", inverse geometric model" if use_inverse_geom_model else "",
            "the HIP device" if self.is_cuda else "the CPU (parameters only - the head itself needs a HIP device)"))

    # ---- training-time switches of the reference API (model.py:168-195); training itself is out of scope here
    def _transform_net(self):
        return self.os2d_head_creator.aligner.parameter_regressor

    def _branches(self):
        return (self.net_feature_maps, self.net_label_features)

    def train(self, mode=True, freeze_bn_in_extractor=False, freeze_transform_params=False, freeze_bn_transform=False):
        super(Os2dModel, self).train(mode)
        requested = ((freeze_bn_in_extractor, self.freeze_bn), (freeze_transform_params, self.freeze_transform_params),
                     (freeze_bn_transform, self._transform_net().freeze_bn))
        for wanted, action in requested:
            if wanted:
                action()
        return self

    def freeze_bn(self):
        for branch in self._branches():
            branch.freeze_bn()

    def freeze_transform_params(self):
        net = self._transform_net()
        net.eval()
        net.requires_grad_(False)

    def freeze_extractor_blocks(self, num_blocks=0):
        for branch in self._branches():
            branch.freeze_blocks(num_blocks)

    def get_num_blocks_in_feature_extractor(self):
        return self.net_feature_maps.get_num_blocks_in_feature_extractor()

    def apply_class_heads_to_feature_maps(self, feature_maps, class_head):
        """reference model.py:197-233: flatten H,W of the head outputs (views, no copies)."""
        num_images = feature_maps.size(0)
        loc, cls, cls_detached, corners = class_head(feature_maps)
        num_labels = cls.size(1)
        assert loc.size(-2) == cls.size(-2) and loc.size(-1) == cls.size(-1), \
            "Class and loc score should have same spatial sizes, but have {0} and {1}".format(cls.size(), loc.size())
        cls_flat = cls.reshape(num_images, num_labels, -1)
        cls_det_flat = cls_flat if cls_detached is cls else cls_detached.reshape(num_images, num_labels, -1)
        loc = loc.reshape(num_images, num_labels, 4, -1)
        corners = corners.reshape(num_images, num_labels, 8, -1)
        return loc, cls_flat, cls_det_flat, corners

    def forward(self, images=None, class_images=None, feature_maps=None, class_head=None, train_mode=False,
                fine_tune_features=True):
        """reference model.py:235-276.  Two calling conventions:
            forward(images=..., class_images=[...])           (app.py / demo / training-style signature)
            forward(feature_maps=..., class_head=...)         (evaluation: pre-extracted features + prebuilt head)
        Returns (loc [A,B,4,HW], cls [A,B,HW], cls_detached [A,B,HW], FeatureMapSize, corners [A,B,8,HW])."""
        if train_mode:
            raise RuntimeError("train_mode=True: training through the HIP head is out of scope (inference only)")
        # what is missing is computed from what was given; the messages are the reference's (callers match on them)
        need_features, need_head = feature_maps is None, class_head is None
        assert not need_features or images is not None, "If feature_maps is None than images cannot be None"
        assert not need_head or class_images is not None, "If class_conv_layer is None than class_images cannot be None"
        with torch.no_grad():
            if need_features:
                feature_maps = self.net_feature_maps(images)
            if need_head:
                class_head = self.os2d_head_creator.create_os2d_head(self.net_label_features(class_images))
            loc, cls, cls_det, corners = self.apply_class_heads_to_feature_maps(feature_maps, class_head)
        return loc, cls, cls_det, FeatureMapSize(img=feature_maps), corners

    def get_feature_map_size(self, img_size):
        """Closed form for the C4 backbone instead of the reference's dummy forward pass (model.py:278-288)."""
        return feature_map_size_c4(img_size)

    def build_box_coder(self, do_nms_across_classes=False):
        """The decode-side box coder wired like reference model.py:35-41."""
        return Os2dBoxCoder(output_box_grid_generator=self.os2d_head_creator.box_grid_generator_image_level,
                            function_get_feature_map_size=self.get_feature_map_size,
                            do_nms_across_classes=do_nms_across_classes)

    def init_model_from_file(self, path, init_affine_transform_path=""):
        """reference model.py:290-345, same outcomes for the same files: a whole-model ``{"net": ..., "optimizer": ...}``
        checkpoint is tried first; if ANYTHING about that fails (unreadable file, no "net", keys that do not fit the whole
        model - e.g. a "net" that holds only a feature extractor) the backbone-only chain of ``_load_network`` runs instead;
        afterwards an optional weakalign TransformNet, whose failure is reported and ignored.  Returns the optimizer state
        found in the checkpoint (or None)."""
        checkpoint, optimizer = None, None
        try:
            checkpoint = self._read_checkpoint(path)
            optimizer = self._load_whole_model(checkpoint)
        except (KeyboardInterrupt, SystemExit):
            raise
        except Exception as e:   # noqa: BLE001 - the reference's loader is this permissive (bare except, model.py:321)
            self.logger.info("whole-model load did not work ({}); initialising the feature extractors only".format(
                type(e).__name__))
            if checkpoint is not None:
                class_branch = self.net_label_features.net_class_features
                self._load_network(class_branch, checkpoint)
                if not self.merge_branch_parameters:            # separate branches start from the same weights
                    self._load_network(self.net_feature_maps, class_branch.state_dict())
        if init_affine_transform_path:
            self._load_weakalign_transform(init_affine_transform_path)
        return optimizer

    def _read_checkpoint(self, path):
        if not path:
            return None
        self.logger.info("reading checkpoint {}".format(path))
        return torch.load(path, map_location="cpu")

    def _load_whole_model(self, checkpoint):
        """checkpoint["net"] into the whole model (strict); -> checkpoint.get("optimizer").  Raises when it does not fit."""
        if not checkpoint or "net" not in checkpoint:
            raise KeyError("no 'net' entry in the checkpoint")
        self.load_state_dict(checkpoint["net"])
        has_optimizer = "optimizer" in checkpoint
        self.logger.info("whole model restored from the checkpoint{}".format(
            "; optimizer state found" if has_optimizer else "; no optimizer state in it (a new optimizer starts from scratch)"))
        return checkpoint["optimizer"] if has_optimizer else None

    def _load_weakalign_transform(self, path):
        """A weakalign checkpoint's FeatureRegression.* into the TransformNet (reference model.py:331-345)."""
        try:
            data = torch.load(path, map_location="cpu")
            init_from_weakalign_model(data["state_dict"], None, affine_regressor=self._transform_net())
            self.logger.info("TransformNet initialised from the weakalign model {}".format(path))
        except (KeyboardInterrupt, SystemExit):
            raise
        except Exception as e:   # noqa: BLE001 - reference model.py:344 ignores any failure here
            self.logger.info("TransformNet NOT initialised from {} ({})".format(path, type(e).__name__))

    def _load_network(self, net, model_data):
        """reference model.py:347-386 fall-back chain."""
        for attempt in (lambda: net.load_state_dict(model_data),
                        lambda: net.load_state_dict(model_data["net"], strict=False),
                        lambda: init_from_weakalign_model(model_data["state_dict"], self.net_feature_maps),
                        lambda: net.load_state_dict(model_data, strict=False)):
            try:
                attempt()
                return True
            except Exception:   # noqa: BLE001 - mirrors the reference's permissive loader
                continue
        self.logger.info("none of the known checkpoint layouts fits this network: it keeps its initial weights")
        return False


def init_from_weakalign_model(src_state_dict, feature_extractor=None, affine_regressor=None, tps_regressor=None):
    """Map a weakalign checkpoint (FeatureExtraction.model.N.* / FeatureRegression.*) onto our modules
    (reference model.py:389-426, including the FC -> 5x5 conv reshape of ``linear.weight``)."""
    prefix = {"conv1.": "FeatureExtraction.model.0.", "bn1.": "FeatureExtraction.model.1."}
    for stage, (n, idx) in enumerate(((3, 4), (4, 5), (23, 6)), start=1):
        for i in range(n):
            prefix["layer{}.{}".format(stage, i)] = "FeatureExtraction.model.{}.{}".format(idx, i)
    with torch.no_grad():
        if feature_extractor is not None:
            for k, v in feature_extractor.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    continue
                for tgt, src in prefix.items():
                    if k.startswith(tgt):
                        v.copy_(src_state_dict[k.replace(tgt, src)])
                        break
        for regressor, pre in ((affine_regressor, "FeatureRegression."), (tps_regressor, "FeatureRegression2.")):
            if regressor is None:
                continue
            for k, v in regressor.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    continue
                src = src_state_dict[pre + k]
                v.copy_(src.view(-1, 64, 5, 5) if k == "linear.weight" else src)
