"""CPU: the range plan of the split-fp16 path (``TransformationNet.range_plan``): its bounds hold, no operand leaves the
fp16 range, and the scheme stays fp32-equivalent on networks whose intermediate ranges span many decades - checked with a
float64 model of the kernels' arithmetic (tests/f16x3_model.py)."""
import pytest
import torch
import torch.nn.functional as F

import f16x3_model as M
import util
from oracle import head_oracle as O
from os2d_amd.modeling.head import TransformationNet
from os2d_amd.utils import synthetic


def _net(state, P):
    net = TransformationNet(output_dim=P, use_cuda=False)
    net.load_state_dict(state)
    return net.eval()


def _rnorm(C=32, H=12, W=14, B=2, seed=3):
    fm = synthetic.make_feature_map(C, H, W, seed=seed)
    q = O.prepare_class_maps(synthetic.make_class_feature_maps(B, C, sizes=[(15, 15), (13, 17)], seed=seed + 100))
    corr = O.correlation(q, fm).double()
    r = corr.clamp(min=0)
    return r / (r.pow(2).sum(1, keepdim=True).sqrt() + 1e-6)


def _fp64_params(rnorm, folded):
    (w1, b1), (w2, b2), (w3, b3) = folded
    x = F.relu(F.conv2d(rnorm, w1, b1, padding=3))
    h1 = x
    x = F.relu(F.conv2d(x, w2, b2, padding=2))
    return F.conv2d(x, w3, b3, padding=2), h1, x


@pytest.mark.parametrize("kind", ["plain", "adversarial"])
@pytest.mark.parametrize("P", [6, 4])
def test_bounds_hold_and_scheme_is_fp32_equivalent(kind, P):
    state = synthetic.make_transform_net_state(P, seed=11) if kind == "plain" else util.adversarial_transform_net_state(P, seed=11)
    net = _net(state, P)
    plan, folded = net.range_plan(), net._folded()
    rnorm = _rnorm()
    p64, h1, h2 = _fp64_params(rnorm, folded)
    b1, b2 = plan["bounds"]
    # the bounds are rigorous: every activation of this input sits below its channel's bound
    assert bool((h1.amax(dim=(0, 2, 3)) <= b1).all()) and bool((h2.amax(dim=(0, 2, 3)) <= b2).all())
    # ... and not absurdly loose: the largest activation of a channel is within 2^14 of the bound (22 bits are kept down
    # to 2^-18 of it)
    live1 = h1.amax(dim=(0, 2, 3)) > 0
    assert float((b1[live1] / h1.amax(dim=(0, 2, 3))[live1]).max()) < 2 ** 14
    # scaled bounds / weights fit fp16 with margin
    for e, b in zip(plan["out_exp"][:2], (b1, b2)):
        assert float((b * torch.exp2(e.double())).max()) <= 32768.0
    p16, mids = M.transform_net_f16x3(rnorm, folded, plan)          # raises on any fp16 overflow
    # fp32-equivalence: the model of the split arithmetic agrees with plain fp64 to a few fp32 ulps of the parameters
    assert float((p16 - p64).abs().max()) < 3e-6, float((p16 - p64).abs().max())


def test_adversarial_network_equals_its_base_network():
    """The hostile rescaling is function preserving (that is what makes it a fair test): fp32 torch agrees on both."""
    base = synthetic.make_transform_net_state(6, seed=11)
    adv = util.adversarial_transform_net_state(6, seed=11)
    corr = torch.rand(2, 225, 9, 10, generator=torch.Generator().manual_seed(5)) - 0.3
    with torch.no_grad():
        a, b = O.transform_net(corr, base), O.transform_net(corr, adv)
    assert util.maxdiff(a, b) < 2e-5
    s = adv["conv.1.weight"] / base["conv.1.weight"]
    assert float(s.abs().max() / s.abs().min()) > 1e8                   # nine decades between channels
    assert float(adv["conv.1.running_var"].min()) == pytest.approx(1e-6)


def test_plain_scalar_scaling_would_overflow_on_the_adversarial_network():
    """Why the exponents are per channel and bound-derived: with the activations stored unscaled (round 1) the hostile
    network drives conv 7x7 outputs beyond 65504."""
    adv = util.adversarial_transform_net_state(6, seed=11)
    net = _net(adv, 6)
    _, h1, _ = _fp64_params(_rnorm(), net._folded())
    assert float(h1.max()) > M.FP16_MAX


def test_degenerate_weights_give_a_valid_plan():
    """All-zero layers / channels (the reference initialises ``linear.weight`` to zero, head.py:632-642) and huge
    magnitudes: exponents stay finite and clamped."""
    net = TransformationNet(output_dim=6, use_cuda=False).eval()        # linear.weight == 0
    plan = net.range_plan()
    assert int(plan["weight_exp"][2].abs().max()) == 0
    with torch.no_grad():
        net.conv[0].weight[3].zero_()
        net.conv[0].bias[3] = 0.0
        net.conv[1].bias[3] = 0.0
        net.conv[1].running_mean[3] = 0.0
        net.conv[3].weight.mul_(1e30)
    plan = net.range_plan()
    for group in ("in_exp", "out_exp", "weight_exp"):
        for e in plan[group]:
            if e is not None:
                assert e.dtype == torch.int32 and int(e.abs().max()) <= 60
    assert int(plan["out_exp"][0][3]) == 0 and int(plan["weight_exp"][0][3]) == 0


def test_split_rows_f16_scales_and_reconstructs():
    """Host logic of precision "fftx3": per-row power-of-two scales keep the largest entry of every row in (16384, 32768]
    (no fp16 overflow, also not through rounding), and hi + lo reproduces the scaled value to 2^-22 relative / 2^-25 absolute -
    for rows whose magnitudes differ by twelve decades."""
    import torch
    from os2d_amd.modeling.head import split_rows_f16
    g = torch.Generator().manual_seed(5)
    T = torch.randn(6, 40, 30, 2, generator=g, dtype=torch.float64)
    T *= torch.tensor([1e-6, 1e-3, 1.0, 37.5, 1e3, 1e6], dtype=torch.float64).view(-1, 1, 1, 1)
    T[2, 0, 0, 0] = 0.0
    T[3, 1] *= 1e-7                                   # entries far below the row maximum: lo halves subnormal
    ref = T.clone()
    hi, lo, wexp = split_rows_f16(T)
    scaled = ref * torch.exp2(wexp).view(-1, 1, 1, 1)
    amax = scaled.abs().amax(dim=(1, 2, 3))
    assert bool((amax <= 32768).all()) and bool((amax > 16384).all())
    assert bool(torch.isfinite(hi.float()).all()) and bool(torch.isfinite(lo.float()).all())
    err = (hi.double() + lo.double() - scaled).abs()
    assert bool((err <= torch.maximum(scaled.abs() * 2.0 ** -21, torch.full_like(err, 2.0 ** -24))).all())
    assert float(err.max()) <= 32768 * 2.0 ** -21
