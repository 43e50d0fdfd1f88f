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
    """reference model.py:123-288."""
    default_normalization = {"mean": (0.485, 0.456, 0.406), "std": (0.229, 0.224, 0.225)}

    def __init__(self, logger=None, is_cuda=False, merge_branch_parameters=False, use_group_norm=False,
                 backbone_arch="resnet50", use_inverse_geom_model=True, simplify_affine=False, img_normalization=None):
        super(Os2dModel, self).__init__()
        self.logger = logger or logging.getLogger("OS2D")
        self.use_group_norm = use_group_norm
        self.img_normalization = img_normalization if img_normalization else self.default_normalization
        self.net_feature_maps = build_feature_extractor(backbone_arch, use_group_norm)
        self.merge_branch_parameters = merge_branch_parameters
        extractor = self.net_feature_maps if merge_branch_parameters else build_feature_extractor(backbone_arch, use_group_norm)
        self.simplify_affine = simplify_affine
        self.use_inverse_geom_model = use_inverse_geom_model
        self.os2d_head_creator = build_os2d_head_creator(self.simplify_affine, is_cuda, self.use_inverse_geom_model,
                                                         self.net_feature_maps.feature_map_stride,
                                                         self.net_feature_maps.feature_map_receptive_field)
        self.net_label_features = LabelFeatureExtractor(feature_extractor=extractor)
        self.eval()
        self.is_cuda = is_cuda
        if self.is_cuda:
            self.logger.info("Creating model on one GPU")
            self.cuda()
        else:
            self.logger.info("Creating model on CPU (parameters only: the head itself needs a HIP device)")

    def train(self, mode=True, freeze_bn_in_extractor=False, freeze_transform_params=False, freeze_bn_transform=False):
        super(Os2dModel, self).train(mode)
        if freeze_bn_in_extractor:
            self.freeze_bn()
        if freeze_transform_params:
            self.freeze_transform_params()
        if freeze_bn_transform:
            self.os2d_head_creator.aligner.parameter_regressor.freeze_bn()
        return self

    def freeze_bn(self):
        self.net_feature_maps.freeze_bn()
        self.net_label_features.freeze_bn()

    def freeze_transform_params(self):
        self.os2d_head_creator.aligner.parameter_regressor.eval()
        for p in self.os2d_head_creator.aligner.parameter_regressor.parameters():
            p.requires_grad = False

    def freeze_extractor_blocks(self, num_blocks=0):
        self.net_feature_maps.freeze_blocks(num_blocks)
        self.net_label_features.freeze_blocks(num_blocks)

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
        with torch.no_grad():
            if feature_maps is None:
                assert images is not None, "If feature_maps is None than images cannot be None"
                feature_maps = self.net_feature_maps(images)
            if class_head is None:
                assert class_images is not None, "If class_conv_layer is None than class_images cannot be None"
                class_feature_maps = self.net_label_features(class_images)
                class_head = self.os2d_head_creator.create_os2d_head(class_feature_maps)
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
        """reference model.py:290-345: full ``{"net":..., "optimizer":...}`` checkpoint first; if ANYTHING about that
        fails (no 'net' key, keys that do not fit the whole model - e.g. a checkpoint whose 'net' holds only a feature
        extractor) fall through to the backbone-only chain of ``_load_network`` like the reference does; optionally a
        weakalign TransformNet afterwards, whose failure is logged and ignored (reference model.py:331-345).
        Returns the optimizer state (or None)."""
        optimizer = None
        checkpoint = None
        try:
            if path:
                self.logger.info("Reading model file {}".format(path))
                checkpoint = torch.load(path, map_location="cpu")
            if checkpoint and "net" in checkpoint:
                self.load_state_dict(checkpoint["net"])
                self.logger.info("Loaded complete model from checkpoint")
            else:
                self.logger.info("Cannot find 'net' in the checkpoint file")
                raise RuntimeError()
            if "optimizer" in checkpoint:
                optimizer = checkpoint["optimizer"]
                self.logger.info("Loaded optimizer from checkpoint")
            else:
                self.logger.info("Cannot find 'optimizer' in the checkpoint file. Initializing optimizer from scratch.")
        except (KeyboardInterrupt, SystemExit):
            raise
        except Exception:   # noqa: BLE001 - the reference's permissive loader (bare except, model.py:321)
            self.logger.info("Failed to load the full model, trying to init feature extractors")
            if checkpoint is not None:
                self._load_network(self.net_label_features.net_class_features, checkpoint)
                if not self.merge_branch_parameters:
                    self._load_network(self.net_feature_maps, self.net_label_features.net_class_features.state_dict())
        if init_affine_transform_path:
            try:
                self.logger.info("Trying to init affine transform from {}".format(init_affine_transform_path))
                data = torch.load(init_affine_transform_path, map_location="cpu")
                init_from_weakalign_model(data["state_dict"], None,
                                          affine_regressor=self.os2d_head_creator.aligner.parameter_regressor)
                self.logger.info("Successfully initialized the affine transform from the provided weakalign model.")
            except (KeyboardInterrupt, SystemExit):
                raise
            except Exception:   # noqa: BLE001 - reference model.py:344
                self.logger.info("Could not init affine transform from {0}.".format(init_affine_transform_path))
        return optimizer

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
        self.logger.info("Could not init anything. Starting from scratch.")
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
