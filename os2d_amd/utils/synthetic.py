"""Deterministic synthetic inputs for parity tests, golden fixtures and the benchmark.

Everything is drawn from ``numpy.random.RandomState`` (the legacy MT19937 stream, frozen by numpy's
compatibility policy), so the same seed gives bit-identical tensors in the development container
(where the golden fixtures are produced) and on the GPU box (where they are consumed).  That lets
the fixtures store only *seeds* for the 6.5 MB TransformNet weights plus small checksums.

Why these distributions ("hard parts"):
  * features are ``relu(randn)``: real ResNet-C4 features are post-ReLU; i.i.d. N(0,1) features are
    nearly orthogonal in 1024-d and give |score| < 0.01, useless for a 1e-4 parity check;
  * the last TransformNet layer (``linear``) is perturbed: the reference initialises it to weight 0 /
    identity bias (reference os2d/modeling/head.py:632-642) which makes every transform the
    identity and the three convolutions dead code for the outputs;
  * BatchNorm running statistics and affine parameters are perturbed so that BN folding is exercised.
"""
import numpy as np
import torch

TRANSFORM_NET_CHANNELS = (225, 128, 64)
TRANSFORM_NET_KERNELS = (7, 5, 5)


def _rs(seed):
    return np.random.RandomState(int(seed) & 0x7FFFFFFF)


def relu_randn(shape, seed):
    """Post-ReLU Gaussian tensor, float32."""
    x = _rs(seed).standard_normal(size=tuple(shape)).astype(np.float32)
    return torch.from_numpy(np.maximum(x, 0.0))


def randn_tensor(shape, seed, scale=1.0):
    """Sign-mixed Gaussian tensor (NOT post-ReLU), float32: negative correlations / scores."""
    return torch.from_numpy((scale * _rs(seed).standard_normal(size=tuple(shape))).astype(np.float32))


def make_feature_map(C, H, W, seed=0, A=1):
    """Image feature map [A, C, H, W] (stand-in for the ResNet-C4 output)."""
    return relu_randn((A, C, H, W), seed)


def make_class_feature_maps(B, C, sizes=None, seed=1000):
    """List of B class feature maps [1, C, h_b, w_b]; ``sizes`` cycles over (h, w) pairs."""
    if sizes is None:
        sizes = [(15, 15)]
    out = []
    for b in range(B):
        h, w = sizes[b % len(sizes)]
        out.append(relu_randn((1, C, h, w), seed + b))
    return out


def make_transform_net_state(P, seed=1, linear_std=0.02, linear_bias=None):
    """State dict of the TransformNet (keys as in reference head.py:612-629):

    conv.0 (Conv 225->128 k7), conv.1 (BN 128), conv.3 (Conv 128->64 k5), conv.4 (BN 64),
    linear (Conv 64->P k5).  Conv weights ~ U(+-1/sqrt(fan_in)) (the PyTorch default scale),
    BN perturbed, linear.weight ~ N(0, linear_std), linear.bias = identity transform + N(0, 0.02), or ``linear_bias``
    verbatim when given (the random stream is consumed identically either way).
    """
    rs = _rs(seed)
    sd = {}

    def conv(name, cout, cin, k):
        bound = 1.0 / np.sqrt(cin * k * k)
        sd[name + ".weight"] = rs.uniform(-bound, bound, size=(cout, cin, k, k)).astype(np.float32)
        sd[name + ".bias"] = rs.uniform(-bound, bound, size=(cout,)).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = (1.0 + 0.1 * rs.standard_normal(c)).astype(np.float32)
        sd[name + ".bias"] = (0.1 * rs.standard_normal(c)).astype(np.float32)
        sd[name + ".running_mean"] = (0.1 * rs.standard_normal(c)).astype(np.float32)
        sd[name + ".running_var"] = rs.uniform(0.5, 1.5, size=c).astype(np.float32)
        sd[name + ".num_batches_tracked"] = np.array(1, dtype=np.int64)

    c0, c1, c2 = TRANSFORM_NET_CHANNELS
    k1, k2, k3 = TRANSFORM_NET_KERNELS
    conv("conv.0", c1, c0, k1)
    bn("conv.1", c1)
    conv("conv.3", c2, c1, k2)
    bn("conv.4", c2)
    sd["linear.weight"] = (linear_std * rs.standard_normal((P, c2, k3, k3))).astype(np.float32)
    bias = np.zeros(P, dtype=np.float32)
    if P == 6:
        bias[0] = 1.0
        bias[4] = 1.0
    elif P == 4:
        bias[0] = 1.0
        bias[2] = 1.0
    else:
        raise ValueError("P must be 6 (affine) or 4 (simplified affine), got {}".format(P))
    sd["linear.bias"] = (bias + 0.02 * rs.standard_normal(P)).astype(np.float32)
    if linear_bias is not None:     # a chosen transform instead of (identity + noise): fixtures far from the identity
        sd["linear.bias"] = np.asarray(linear_bias, dtype=np.float32).reshape(P)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def state_checksum(state):
    """Order-independent float64 checksum of a state dict (to verify regenerated weights)."""
    total = 0.0
    for k in sorted(state):
        v = state[k].double()
        total += float((v * v).sum()) + float(v.sum()) * 1e-3
    return total


def fill_model_state(state_dict, seed, P):
    """Deterministic non-trivial weights for a whole ``Os2dModel`` state dict (backbone + TransformNet), a function of
    the sorted key names and shapes only - the golden generator applies it to the REFERENCE model, the tests to ours,
    so equal keys/shapes give equal weights.  Conv weights ~ N(0, sqrt(2 / fan_out)) (the ResNet initialisation scale),
    norm layers perturbed around identity, running statistics live; the TransformNet comes from
    ``make_transform_net_state`` so that the alignment path is exercised."""
    rs = _rs(seed)
    out = {}
    tn_prefix = "os2d_head_creator.aligner.parameter_regressor."
    tn = make_transform_net_state(P, seed=seed + 1)
    for k in sorted(state_dict):
        v = state_dict[k]
        shape = tuple(v.shape)
        if k.startswith(tn_prefix):
            out[k] = tn[k[len(tn_prefix):]].clone().reshape(shape)
        elif k.endswith("num_batches_tracked"):
            out[k] = torch.ones(shape, dtype=v.dtype)
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy(rs.uniform(0.5, 1.5, size=shape).astype(np.float32))
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy((0.1 * rs.standard_normal(shape)).astype(np.float32))
        elif len(shape) == 4:
            std = np.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))
            out[k] = torch.from_numpy((std * rs.standard_normal(shape)).astype(np.float32))
        elif k.endswith("weight"):
            out[k] = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32))
        else:
            out[k] = torch.from_numpy((0.1 * rs.standard_normal(shape)).astype(np.float32))
        assert tuple(out[k].shape) == shape, k
    return out
