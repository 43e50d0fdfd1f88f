"""CPU: checkpoint compatibility of the model wrapper (reference os2d/modeling/model.py:290-426, os2d/utils/logger.py:137-160
format: {"net": state_dict, "optimizer": ...}; backbone-only files; weakalign FeatureExtraction / FeatureRegression maps)."""
import os

import pytest
import torch

from os2d_amd.modeling.model import Os2dModel, init_from_weakalign_model


def _perturbed(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.randn(p.shape, generator=g))
    return net


def test_full_checkpoint_roundtrip(tmp_path):
    src = _perturbed(Os2dModel(is_cuda=False, merge_branch_parameters=True, backbone_arch="resnet50"), 1)
    path = tmp_path / "checkpoint.pth"
    torch.save({"net": src.state_dict(), "optimizer": {"state": {}, "param_groups": []}, "epoch": 3}, str(path))
    dst = Os2dModel(is_cuda=False, merge_branch_parameters=True, backbone_arch="resnet50")
    opt = dst.init_model_from_file(str(path))
    assert opt == {"state": {}, "param_groups": []}
    a, b = src.state_dict(), dst.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_backbone_only_checkpoint(tmp_path):
    """A plain torchvision-style ResNet state dict (incl. layer4 / fc, as downloaded) initialises both branches."""
    src = _perturbed(Os2dModel(is_cuda=False, merge_branch_parameters=False, backbone_arch="resnet50"), 2)
    sd = {k: v.clone() for k, v in src.net_feature_maps.state_dict().items()}
    sd["fc.weight"] = torch.zeros(1000, 2048)            # extra keys of a full classification net are tolerated
    sd["layer4.0.conv1.weight"] = torch.zeros(512, 1024, 1, 1)
    path = tmp_path / "resnet50.pth"
    torch.save(sd, str(path))
    dst = Os2dModel(is_cuda=False, merge_branch_parameters=False, backbone_arch="resnet50")
    dst.init_model_from_file(str(path))
    for k, v in src.net_feature_maps.state_dict().items():
        assert torch.equal(dst.net_label_features.net_class_features.state_dict()[k], v)
        assert torch.equal(dst.net_feature_maps.state_dict()[k], v)


def test_weakalign_transform_mapping():
    """FeatureRegression.* -> TransformNet, with the FC weight [6, 64*5*5] reshaped to the 5x5 conv (model.py:415-426)."""
    net = Os2dModel(is_cuda=False, merge_branch_parameters=True, backbone_arch="resnet101")
    reg = net.os2d_head_creator.aligner.parameter_regressor
    g = torch.Generator().manual_seed(4)
    src = {}
    for k, v in reg.state_dict().items():
        if k.endswith("num_batches_tracked"):
            continue
        t = torch.randn(v.shape, generator=g)
        src["FeatureRegression." + k] = t.reshape(6, -1) if k == "linear.weight" else t
    init_from_weakalign_model(src, None, affine_regressor=reg)
    got = reg.state_dict()
    assert torch.equal(got["linear.weight"], src["FeatureRegression.linear.weight"].view(6, 64, 5, 5))
    assert torch.equal(got["conv.0.weight"], src["FeatureRegression.conv.0.weight"])
    assert torch.equal(got["conv.4.running_var"], src["FeatureRegression.conv.4.running_var"])


@pytest.mark.parametrize("name,merge,simplify,inverse,arch", [("v2_merged", True, False, True, "resnet50"), ("v1_split", False, True, False, "resnet50"),
                                                              ("v1_r101", False, True, False, "resnet101")])
def test_state_dict_layout_equals_the_reference_model(name, merge, simplify, inverse, arch):
    """Keys, ORDER and shapes of ``Os2dModel.state_dict()`` against those recorded from the reference's own Os2dModel
    (tests/golden/model_forward.npz): reference checkpoints load key for key, and aliases of shared modules
    (merge_branch_parameters) appear under both names in the same order."""
    import numpy as np
    from os2d_amd.modeling.model import Os2dModel
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_forward.npz"))
    net = Os2dModel(is_cuda=False, merge_branch_parameters=merge, backbone_arch=arch,
                    use_inverse_geom_model=inverse, simplify_affine=simplify)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in d["keys_" + name]]
    assert [",".join(str(x) for x in v.shape) for v in sd.values()] == [str(s) for s in d["shapes_" + name]]


def test_checkpoint_whose_net_holds_only_a_backbone_falls_back(tmp_path):
    """{"net": <feature extractor only>}: the whole-model load fails and the reference falls through to the backbone
    chain (model.py:321-325 -> _load_network step 2: net.load_state_dict(model_data["net"], strict=False))."""
    src = _perturbed(Os2dModel(is_cuda=False, merge_branch_parameters=False, backbone_arch="resnet50"), 5)
    path = tmp_path / "backbone_ckpt.pth"
    torch.save({"net": src.net_feature_maps.state_dict(), "epoch": 1, "loss": 0.5}, str(path))
    dst = Os2dModel(is_cuda=False, merge_branch_parameters=False, backbone_arch="resnet50")
    assert dst.init_model_from_file(str(path)) is None
    for k, v in src.net_feature_maps.state_dict().items():
        assert torch.equal(dst.net_label_features.net_class_features.state_dict()[k], v)
        assert torch.equal(dst.net_feature_maps.state_dict()[k], v)


def test_unreadable_affine_transform_file_is_ignored(tmp_path):
    """A missing / malformed init_affine_transform_path is logged and ignored (reference model.py:331-345)."""
    src = _perturbed(Os2dModel(is_cuda=False, merge_branch_parameters=True, backbone_arch="resnet50"), 6)
    path = tmp_path / "checkpoint.pth"
    torch.save({"net": src.state_dict()}, str(path))
    bad = tmp_path / "weakalign.pth"
    torch.save({"not_state_dict": 1}, str(bad))
    for affine in (str(bad), str(tmp_path / "does_not_exist.pth")):
        dst = Os2dModel(is_cuda=False, merge_branch_parameters=True, backbone_arch="resnet50")
        dst.init_model_from_file(str(path), init_affine_transform_path=affine)
        a, b = src.state_dict(), dst.state_dict()
        assert all(torch.equal(a[k], b[k]) for k in a)
