"""Host-side mirror of the reference's head API (os2d/modeling/head.py) on top of libos2d_hip.so.

Same class names, constructor arguments, attribute names and state-dict keys as the reference so that a caller of
``Os2dHeadCreator.create_os2d_head`` / ``Os2dHead.forward`` can switch over:

    build_os2d_head_creator  reference head.py:12-15
    Os2dAlignment            reference head.py:43-193   (holds the TransformNet; transform assembly runs in HIP)
    Os2dHeadCreator          reference head.py:204-268
    Os2dHead                 reference head.py:271-435  (forward = ONE call of os2d_head_forward for all classes)
    TransformationNet        reference head.py:604-661

All arithmetic of the path runs in the hand-written gfx950 kernels (os2d_amd/csrc); PyTorch only owns device
memory, the stream and the parameter containers.  There is no CPU / eager fallback: CPU tensors raise.
Training through the head (autograd) is out of scope and raises as well.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn

from .. import _lib
from ..structures.feature_map import FeatureMapSize
from .box_coder import BoxGridGenerator

TEMPLATE = 15
QROWS = 256
PRECISIONS = {"f32": 0, "f16x3": 1, "f16x2": 2}     # OS2D_PRECISION_* of include/os2d_hip.h


def resolve_precision(precision=None):
    """Arithmetic of the two large TransformNet convolutions: "f32" (exact fp32 MFMA) or "f16x3" (fp16 hi/lo split on
    the half-precision matrix cores, fp32-equivalent results: same 2.4e-7 agreement with the reference on every
    parity case), or "f16x2" (as f16x3, but the 7x7 layer takes its weights as fp16 roundings only: 2/3 of the
    matrix-core work, box regression within 5e-5 and scores within 1e-6 of fp32, inside the 1e-4 parity bound).
    Default from $OS2D_PRECISION, else "f16x3"."""
    precision = precision or os.environ.get("OS2D_PRECISION", "f16x3")
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of {}, got {!r}".format(sorted(PRECISIONS), precision))
    return precision


def build_os2d_head_creator(do_simple_affine, is_cuda, use_inverse_geom_model, feature_map_stride,
                            feature_map_receptive_field):
    """reference head.py:12-15."""
    aligner = Os2dAlignment(do_simple_affine, is_cuda, use_inverse_geom_model)
    return Os2dHeadCreator(aligner, feature_map_stride, feature_map_receptive_field)


# --------------------------------------------------------------------------------------------- workspace
_WORKSPACES = {}


def workspace_cap_bytes():
    """Upper bound of the per-device scratch buffer (classes are processed in chunks that fit).  An MI355X has
    288 GB of HBM3E, so the default is generous: 32 GiB holds ~2300 classes at 60x80 in one chunk."""
    return int(float(os.environ.get("OS2D_WORKSPACE_GB", "32")) * (1 << 30))


def get_workspace(device, wanted, minimum):
    """Grow-only scratch buffer, one per (device, stream): concurrent heads on different HIP streams (the pyramid
    runner) must not share intermediates."""
    size = max(min(wanted, workspace_cap_bytes()), minimum)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < size:
        _WORKSPACES[key] = None
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        _WORKSPACES[key] = buf
    return buf


def release_workspaces():
    _WORKSPACES.clear()


def _require_device_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("{} must be a torch.Tensor".format(name))
    if not t.is_cuda:
        raise RuntimeError("{} is on {}: the OS2D head runs only on a HIP device (libos2d_hip.so); there is no "
                           "CPU fallback".format(name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("{} must be float32, got {}".format(name, t.dtype))
    return t.contiguous()


# --------------------------------------------------------------------------------------------- TransformNet
class TransformationNet(nn.Module):
    """Parameter container with the reference's layout (head.py:604-646): ``conv`` = Sequential(Conv 225->128 k7,
    BatchNorm, ReLU, Conv 128->64 k5, BatchNorm, ReLU) and ``linear`` = Conv 64->output_dim k5, the latter
    initialised to the identity transform.  The compute is the MFMA implicit-GEMM kernels; ``packed()`` folds
    eval-mode BatchNorm and re-lays the filters out for them (cached until a parameter changes)."""

    def __init__(self, output_dim=6, use_cuda=True, normalization="batchnorm", kernel_sizes=[7, 5],
                 channels=[128, 64], input_feature_dim=15 * 15, num_groups=16):
        super(TransformationNet, self).__init__()
        if list(kernel_sizes) != [7, 5] or list(channels) != [128, 64] or input_feature_dim != 225:
            raise RuntimeError("the HIP TransformNet kernels are built for the OS2D architecture "
                               "(225 -> 128 k7 -> 64 k5 -> P k5)")
        if normalization.lower() != "batchnorm":
            raise RuntimeError("only 'batchnorm' TransformNet normalisation is supported (the reference "
                               "hard-codes it, head.py:72)")
        mods = []
        ch_in = input_feature_dim
        for ch_out, k in zip(channels, kernel_sizes):
            mods += [nn.Conv2d(ch_in, ch_out, kernel_size=k, padding=k // 2), nn.BatchNorm2d(ch_out),
                     nn.ReLU(inplace=True)]
            ch_in = ch_out
        self.conv = nn.Sequential(*mods)
        k = kernel_sizes[-1]
        self.linear = nn.Conv2d(ch_in, output_dim, kernel_size=(k, k), padding=k // 2)
        # identity transform at initialisation (head.py:632-642)
        with torch.no_grad():
            self.linear.weight.zero_()
            self.linear.bias.zero_()
            if output_dim == 6:
                self.linear.bias[0] = 1
                self.linear.bias[4] = 1
            elif output_dim == 4:
                self.linear.bias[0] = 1
                self.linear.bias[2] = 1
        self.output_dim = output_dim
        self._packed_cache = {}
        if use_cuda:
            self.conv.cuda()
            self.linear.cuda()

    def freeze_bn(self):
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    def _state_key(self):
        ts = list(self.parameters()) + list(self.buffers())
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in ts)

    def packed(self, precision=None):
        """Packed TransformNet for the kernels: (w1, b1, w2, b2, w3, b3, scale_log2[3]).
        precision "f32": os2d_pack_conv layouts (scales are 0); "f16x3": the split-half layout of
        os2d_pack_conv_f16x3, each layer pre-scaled by the largest power of two that keeps max|w| <= 16384."""
        precision = resolve_precision(precision)
        if precision == "f16x2":
            precision = "f16x3"        # same packed weights and scales; the kernel just skips the lo halves of layer 1
        key = (precision,) + self._state_key()
        cached = self._packed_cache.get(precision)
        if cached is not None and cached[0] == key:
            return cached[1]
        lib = _lib.load()
        dev = self.linear.weight.device
        if dev.type != "cuda":
            raise RuntimeError("TransformationNet parameters are on {}: move the model to the HIP device "
                               "(no CPU fallback)".format(dev))
        if self.training and any(isinstance(m, nn.BatchNorm2d) and m.training for m in self.modules()):
            raise RuntimeError("TransformationNet is in training mode: the HIP path implements eval-mode "
                               "BatchNorm (running statistics) only; call .eval()")
        stream = _lib.current_stream(dev)
        out, scales = [], []
        P = self.output_dim
        layers = ((1, self.conv[0], self.conv[1]), (2, self.conv[3], self.conv[4]), (3, self.linear, None))
        for layer, conv, bn in layers:
            w = _require_device_f32(conv.weight.detach(), "conv weight")
            b = _require_device_f32(conv.bias.detach(), "conv bias")
            pb = torch.empty(lib.os2d_packed_bias_floats(layer), dtype=torch.float32, device=dev)
            if bn is not None:
                bnp = [_require_device_f32(t.detach(), "bn") for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
                eps = float(bn.eps)
            else:
                bnp, eps = [None] * 4, 0.0
            if precision == "f16x3":
                wmax = w.abs().amax(dim=(1, 2, 3))
                if bn is not None:
                    wmax = wmax * (bnp[0] / torch.sqrt(bnp[3] + eps)).abs()
                folded_max = float(wmax.max())
                scale_log2 = int(math.floor(math.log2(16384.0 / folded_max))) if folded_max > 0 else 0
                scale_log2 = max(-60, min(60, scale_log2))
                pw = torch.empty(lib.os2d_packed_conv_bytes(layer, PRECISIONS[precision]), dtype=torch.uint8, device=dev)
                _lib.check(lib.os2d_pack_conv_f16x3(layer, P, _lib.ptr(w), _lib.ptr(b), *[_lib.ptr(t) for t in bnp],
                                                    ctypes.c_float(eps), scale_log2, _lib.ptr(pw), _lib.ptr(pb), stream),
                           "os2d_pack_conv_f16x3")
                scales.append(scale_log2)
            else:
                pw = torch.empty(lib.os2d_packed_conv_floats(layer), dtype=torch.float32, device=dev)
                _lib.check(lib.os2d_pack_conv(layer, P, _lib.ptr(w), _lib.ptr(b), *[_lib.ptr(t) for t in bnp],
                                              ctypes.c_float(eps), _lib.ptr(pw), _lib.ptr(pb), stream), "os2d_pack_conv")
                scales.append(0)
            out += [pw, pb]
        result = tuple(out) + ((ctypes.c_int * 3)(*scales),)
        self._packed_cache[precision] = (key, result)
        return result

    def forward(self, corr_maps):
        """corr_maps [N,225,H,W] -> transform parameters [N,P,H,W] (reference head.py:648-655), via
        os2d_corr_normalize + the three MFMA conv kernels."""
        corr_maps = _require_device_f32(corr_maps, "corr_maps")
        if corr_maps.dim() != 4 or corr_maps.size(1) != 225:
            raise RuntimeError("corr_maps must be [N,225,H,W], got {}".format(tuple(corr_maps.shape)))
        if torch.is_grad_enabled() and (corr_maps.requires_grad or any(p.requires_grad for p in self.parameters())) and self.training:
            raise RuntimeError("autograd through the HIP TransformNet is not implemented (training is out of scope)")
        lib = _lib.load()
        N, _, H, W = corr_maps.shape
        dev = corr_maps.device
        w1, b1, w2, b2, w3, b3 = self.packed("f32")[:6]
        plane = lib.os2d_plane_floats(H, W)
        stream = _lib.current_stream(dev)
        r = torch.empty(N * 226 * plane, dtype=torch.float32, device=dev)
        h1 = torch.empty(N * 128 * plane, dtype=torch.float32, device=dev)
        h2 = torch.empty(N * 64 * plane, dtype=torch.float32, device=dev)
        out = torch.empty(N, self.output_dim, H, W, dtype=torch.float32, device=dev)
        _lib.check(lib.os2d_corr_normalize(_lib.ptr(corr_maps), _lib.ptr(r), N, H, W, stream), "os2d_corr_normalize")
        _lib.check(lib.os2d_transform_conv(1, _lib.ptr(r), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(h1), N, self.output_dim, H, W, stream), "conv1")
        _lib.check(lib.os2d_transform_conv(2, _lib.ptr(h1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(h2), N, self.output_dim, H, W, stream), "conv2")
        _lib.check(lib.os2d_transform_conv(3, _lib.ptr(h2), _lib.ptr(w3), _lib.ptr(b3), _lib.ptr(out), N, self.output_dim, H, W, stream), "conv3")
        return out


class Os2dAlignment(nn.Module):
    """reference head.py:43-193.  Owns the TransformNet and the transformation-model flags.  The reference's
    ``forward`` returns a materialised [N,H,W,15,15,2] grid tensor; here the grid never exists: the transform
    assembly (head.py:81-153), F.affine_grid (head.py:184) and everything downstream are fused in
    os2d_sample_decode, driven from ``Os2dHead.forward``."""

    def __init__(self, do_simple_affine, is_cuda, use_inverse_geom_model):
        super(Os2dAlignment, self).__init__()
        self.model_type = "affine" if not do_simple_affine else "simple_affine"
        self.use_inverse_geom_model = use_inverse_geom_model
        transform_net_output_dim = 6 if self.model_type == "affine" else 4
        self.out_grid_size = FeatureMapSize(w=TEMPLATE, h=TEMPLATE)
        self.reference_feature_map_size = FeatureMapSize(w=TEMPLATE, h=TEMPLATE)
        self.network_stride = FeatureMapSize(w=1, h=1)
        self.network_receptive_field = FeatureMapSize(w=TEMPLATE, h=TEMPLATE)
        self.input_feature_dim = TEMPLATE * TEMPLATE
        self.parameter_regressor = TransformationNet(output_dim=transform_net_output_dim, use_cuda=is_cuda,
                                                     normalization="batchnorm", kernel_sizes=[7, 5],
                                                     channels=[128, 64], input_feature_dim=self.input_feature_dim)

    @property
    def num_transform_params(self):
        return self.parameter_regressor.output_dim

    def forward(self, corr_maps):
        raise NotImplementedError("Os2dAlignment.forward would materialise the [N,H,W,15,15,2] sampling grids; the "
                                  "HIP path fuses them into Os2dHead.forward (use parameter_regressor(corr_maps) "
                                  "for the raw transform parameters)")


# --------------------------------------------------------------------------------------------- head creator
class Os2dHeadCreator(nn.Module):
    """reference head.py:204-268: a sub-module of the model (it owns the TransformNet parameters) that creates
    ``Os2dHead`` objects holding class features."""

    def __init__(self, aligner, feature_map_stride, feature_map_receptive_field):
        super(Os2dHeadCreator, self).__init__()
        self.aligner = aligner
        rec_field, stride = self.get_rec_field_and_stride_after_concat_nets(
            feature_map_receptive_field, feature_map_stride,
            self.aligner.network_receptive_field, self.aligner.network_stride)
        self.feature_map_stride = feature_map_stride
        self.feature_map_receptive_field = feature_map_receptive_field
        self.box_grid_generator_image_level = BoxGridGenerator(box_size=rec_field, box_stride=stride)
        self.box_grid_generator_feature_map_level = BoxGridGenerator(box_size=self.aligner.network_receptive_field,
                                                                     box_stride=self.aligner.network_stride)

    @staticmethod
    def get_rec_field_and_stride_after_concat_nets(receptive_field_netA, stride_netA, receptive_field_netB, stride_netB):
        """net(x) = netB(netA(x)): rf = strideA*(rfB-1)+rfA, stride = strideA*strideB (reference head.py:223-238)."""
        if isinstance(receptive_field_netA, FeatureMapSize):
            f = Os2dHeadCreator.get_rec_field_and_stride_after_concat_nets
            rw, sw = f(receptive_field_netA.w, stride_netA.w, receptive_field_netB.w, stride_netB.w)
            rh, sh = f(receptive_field_netA.h, stride_netA.h, receptive_field_netB.h, stride_netB.h)
            return FeatureMapSize(w=rw, h=rh), FeatureMapSize(w=sw, h=sh)
        return stride_netA * (receptive_field_netB - 1) + receptive_field_netA, stride_netA * stride_netB

    @staticmethod
    def resize_feature_maps_to_reference_size(ref_size, feature_maps):
        """reference head.py:241-259, as a HIP kernel: returns the resized (NOT normalised) maps [B,C,15,15]."""
        q15, _ = _prepare_class_maps(feature_maps, normalise=False)
        return q15

    def create_os2d_head(self, class_feature_maps):
        """class_feature_maps: list of [1,C,h_i,w_i] device tensors (reference head.py:261-268)."""
        q15, qp = _prepare_class_maps(class_feature_maps, normalise=True)
        return Os2dHead(q15, self.aligner, self.box_grid_generator_image_level,
                        self.box_grid_generator_feature_map_level, _prepared=qp,
                        _stride=self.feature_map_stride, _rec_field=self.feature_map_receptive_field)


def _prepare_class_maps(class_feature_maps, normalise=True):
    """Run os2d_class_prepare on each [1,C,h,w] (or [C,h,w]) map.  Returns (q15 [B,C,15,15], qp [B,C,256])."""
    lib = _lib.load()
    if isinstance(class_feature_maps, torch.Tensor):
        class_feature_maps = [m.unsqueeze(0) for m in class_feature_maps]
    if len(class_feature_maps) == 0:
        raise RuntimeError("need at least one class feature map")
    maps = []
    for fm in class_feature_maps:
        fm = _require_device_f32(fm, "class feature map")
        if fm.dim() == 4:
            assert fm.size(0) == 1, "Can process only batches of size 1, but have {0}".format(fm.size(0))
            fm = fm[0]
        maps.append(fm)
    C = maps[0].size(0)
    dev = maps[0].device
    B = len(maps)
    q15 = torch.empty(B, C, TEMPLATE, TEMPLATE, dtype=torch.float32, device=dev)
    qp = torch.empty(B, C, QROWS, dtype=torch.float32, device=dev)
    stream = _lib.current_stream(dev)
    for b, fm in enumerate(maps):
        if fm.size(0) != C:
            raise RuntimeError("class feature maps disagree on the feature dimension: {} vs {}".format(fm.size(0), C))
        _lib.check(lib.os2d_class_prepare(_lib.ptr(fm), C, fm.size(1), fm.size(2), 1 if normalise else 0,
                                          _lib.ptr(q15[b]), _lib.ptr(qp[b]), stream), "os2d_class_prepare")
    return q15, qp


# --------------------------------------------------------------------------------------------- head
class Os2dHead(nn.Module):
    """reference head.py:271-435.  Holds the L2-normalised 15x15 class feature maps of B classes and computes, for
    a batch of image feature maps, the localisation / recognition outputs of every (image, class) pair.

    Differences from the reference, by design:
      * one ``forward`` handles all B classes in a single library call (the reference's evaluation loops B=1
        heads, os2d/engine/evaluate.py:323-331) - the per-class results are identical;
      * eval mode only: ``output_recognition_transform_detached`` is the same tensor as ``output_recognition``
        (as in the reference when no gradient is required, head.py:400-402).
    """

    def __init__(self, class_feature_maps, aligner, box_grid_generator_image_level,
                 box_grid_generator_feature_map_level, pool_border_width=2, _prepared=None, _stride=None,
                 _rec_field=None):
        super(Os2dHead, self).__init__()
        if pool_border_width != 2:
            raise RuntimeError("the HIP resampling kernel is built for pool_border_width=2 (the only value the "
                               "reference uses, head.py:280)")
        class_feature_maps = _require_device_f32(class_feature_maps, "class_feature_maps")
        if class_feature_maps.dim() != 4 or class_feature_maps.size(2) != TEMPLATE or class_feature_maps.size(3) != TEMPLATE:
            raise RuntimeError("class_feature_maps must be [B,C,15,15], got {}".format(tuple(class_feature_maps.shape)))
        if _prepared is None:
            # public constructor, as in the reference: maps are resized but not yet normalised (head.py:293)
            class_feature_maps, _prepared = _prepare_class_maps(class_feature_maps, normalise=True)
        self.class_feature_maps = class_feature_maps          # normalised, [B,C,15,15]
        self._qp = _prepared                                  # GEMM operand [B,C,256]
        self._qs = None                                       # its fp16 hi/lo split (f16x3 mode), built on first use
        self.class_batch_size = self.class_feature_maps.size(0)
        self.box_grid_generator_image_level = box_grid_generator_image_level
        self.box_grid_generator_feature_map_level = box_grid_generator_feature_map_level
        # pooling mask (head.py:296-302): informational - the kernel hard-codes the 11x11 inner window
        mask = torch.zeros(self.class_batch_size, 1, TEMPLATE, TEMPLATE, dtype=torch.float32,
                           device=self.class_feature_maps.device)
        mask[:, :, pool_border_width:TEMPLATE - pool_border_width, pool_border_width:TEMPLATE - pool_border_width] = 1
        self.class_pool_mask = mask / mask.sum(dim=(2, 3), keepdim=True)
        self.aligner = aligner
        self.precision = None      # None: follow $OS2D_PRECISION (default "f16x3"); or "f32" / "f16x3" / "f16x2"
        box = box_grid_generator_image_level
        self._stride = int(box.box_stride.w)
        # image-level box = stride*(15-1) + receptive field (head.py:223-238)
        self._rec_field = int(box.box_size.w - self._stride * (TEMPLATE - 1))
        if box.box_stride.w != box.box_stride.h or box.box_size.w != box.box_size.h:
            raise RuntimeError("anisotropic strides / receptive fields are not supported by the HIP head")

    def _split_class_operand(self):
        """qs [B, C/8, hi|lo, 256] x 8 halves for the f16x3 correlation (os2d_class_split), built on first use."""
        if self._qs is None:
            lib = _lib.load()
            B, C = self._qp.size(0), self._qp.size(1)
            qs = torch.empty(B * ((C + 7) // 8) * 2 * 256 * 16, dtype=torch.uint8, device=self._qp.device)
            _lib.check(lib.os2d_class_split(_lib.ptr(self._qp), _lib.ptr(qs), B, C, _lib.current_stream(self._qp.device)),
                       "os2d_class_split")
            self._qs = qs
        return self._qs

    @classmethod
    def cat(cls, heads):
        """Merge per-class heads (as the reference's evaluation creates them) into one class-batched head."""
        h0 = heads[0]
        q15 = torch.cat([h.class_feature_maps for h in heads], 0)
        qp = torch.cat([h._qp for h in heads], 0)
        return cls(q15, h0.aligner, h0.box_grid_generator_image_level, h0.box_grid_generator_feature_map_level,
                   _prepared=qp)

    def forward(self, feature_maps, out=None, stage_events=None, precision=None):
        """feature_maps [A,C,H,W] -> (loc [A,B,4,H,W], cls [A,B,1,H,W], cls_detached (same), corners [A,B,8,H,W]).

        ``out``: optional preallocated (loc, cls, corners) device tensors of exactly those shapes (contiguous) - used
        by the class-sharded wrapper to let the kernels write straight into the all-gather buffer.
        ``stage_events``: optional ctypes array of 10 event handles for os2d_head_forward_profiled (bench.py)."""
        feature_maps = _require_device_f32(feature_maps, "feature_maps")
        if feature_maps.dim() != 4:
            raise RuntimeError("feature_maps must be [A,C,H,W], got {}".format(tuple(feature_maps.shape)))
        if torch.is_grad_enabled() and feature_maps.requires_grad:
            raise RuntimeError("autograd through the HIP head is not implemented (training is out of scope): "
                               "call under torch.no_grad()")
        A, C, H, W = feature_maps.shape
        B = self.class_batch_size
        class_feature_dim = self.class_feature_maps.size(1)
        assert C == class_feature_dim, \
            "Feature dimensionality of input={0} and class={1} feature maps has to equal".format(C, class_feature_dim)
        if self._qp.device != feature_maps.device:
            raise RuntimeError("class features are on {} but feature_maps on {}".format(self._qp.device, feature_maps.device))
        lib = _lib.load()
        dev = feature_maps.device
        regressor = self.aligner.parameter_regressor
        P = regressor.output_dim
        precision = resolve_precision(precision or self.precision)
        w1, b1, w2, b2, w3, b3, scales = regressor.packed(precision)
        if out is None:
            loc = torch.empty(A, B, 4, H, W, dtype=torch.float32, device=dev)
            cls = torch.empty(A, B, 1, H, W, dtype=torch.float32, device=dev)
            corners = torch.empty(A, B, 8, H, W, dtype=torch.float32, device=dev)
        else:
            loc, cls, corners = out
            for t, k in ((loc, 4), (cls, 1), (corners, 8)):
                if tuple(t.shape) != (A, B, k, H, W) or not t.is_contiguous() or t.device != dev or t.dtype != torch.float32:
                    raise RuntimeError("out tensors must be contiguous float32 [A,B,{},H,W] on {}".format(k, dev))
        full = ctypes.c_size_t()
        one = ctypes.c_size_t()
        _lib.check(lib.os2d_head_workspace_bytes(A, B, C, H, W, P, ctypes.byref(full)), "os2d_head_workspace_bytes")
        _lib.check(lib.os2d_head_workspace_bytes(A, 1, C, H, W, P, ctypes.byref(one)), "os2d_head_workspace_bytes")
        ws = get_workspace(dev, full.value, one.value)
        _lib.check(lib.os2d_head_forward_ex(
            _lib.ptr(feature_maps), _lib.ptr(self._qp), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2),
            _lib.ptr(w3), _lib.ptr(b3), A, B, C, H, W, P, 1 if self.aligner.use_inverse_geom_model else 0,
            self._stride, self._rec_field, _lib.ptr(loc), _lib.ptr(cls), _lib.ptr(corners),
            _lib.ptr(ws), ws.numel(), _lib.current_stream(dev), PRECISIONS[precision],
            _lib.ptr(self._split_class_operand()) if precision != "f32" else None, scales, stage_events, None),
            "os2d_head_forward_ex")
        return loc, cls, cls, corners


def normalize_feature_map_L2(feature_maps, epsilon=1e-6):
    """reference head.py:597-601.  Provided for API completeness (plain tensor expression; NOT used by the HIP
    path, which folds both normalisations into its GEMM epilogue)."""
    return feature_maps / (feature_maps.norm(dim=1, keepdim=True) + epsilon)
