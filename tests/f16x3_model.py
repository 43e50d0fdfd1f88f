"""CPU model of the split-fp16 ("f16x3") arithmetic of the TransformNet kernels (os2d_amd/csrc/conv_f16x3.hip), test
infrastructure only: every operand is x * 2^e rounded to fp16 hi + fp16 lo with the exponents of
``TransformationNet.range_plan``, a product is hi*hi + hi*lo + lo*hi (the MFMA accumulates exact fp16 products in fp32;
float64 here).  It checks the range plan - no operand may leave the fp16 range - and gives the error of the scheme
against plain fp64 without a GPU."""
import torch
import torch.nn.functional as F

FP16_MAX = 65504.0


def split(x):
    """float64 tensor -> (hi, lo) float64 tensors holding fp16 values; raises if the fp16 range is left."""
    assert float(x.abs().max()) <= FP16_MAX, "fp16 overflow: {}".format(float(x.abs().max()))
    hi = x.to(torch.float16)
    lo = (x - hi.double()).to(torch.float16)
    return hi.double(), lo.double()


def conv_f16x3(x_scaled, w, b, in_exp, out_exp, weight_exp, relu, pad, terms=3):
    """x_scaled: the layer's input as the kernels hold it (x[c] * 2^in_exp[c], float64); returns the output as they
    write it (y[o] * 2^out_exp[o]), or the plain fp32-like output when out_exp is None."""
    w_eff = w * torch.exp2((weight_exp.double().view(-1, 1) - in_exp.double().view(1, -1))).view(w.size(0), w.size(1), 1, 1)
    wh, wl = split(w_eff)
    xh, xl = split(x_scaled)
    acc = F.conv2d(xh, wh, None, padding=pad) + F.conv2d(xl, wh, None, padding=pad)
    if terms == 3:
        acc = acc + F.conv2d(xh, wl, None, padding=pad)
    y = acc * torch.exp2(-weight_exp.double()).view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    if relu:
        y = y.clamp(min=0)
    if out_exp is not None:
        y = y * torch.exp2(out_exp.double()).view(1, -1, 1, 1)
    return y


def transform_net_f16x3(rnorm, folded, plan, terms1=3):
    """rnorm [N,225,H,W] float64 (relu + L2-normalised correlation); folded = TransformationNet._folded();
    plan = TransformationNet.range_plan().  Returns (params float64, dict of the intermediate scaled activations)."""
    (w1, b1), (w2, b2), (w3, b3) = folded
    e_in, e_out, e_w = plan["in_exp"], plan["out_exp"], plan["weight_exp"]
    x0 = rnorm * torch.exp2(e_in[0].double()).view(1, -1, 1, 1)
    h1 = conv_f16x3(x0, w1, b1, e_in[0], e_out[0], e_w[0], True, 3, terms1)
    h2 = conv_f16x3(h1, w2, b2, e_in[1], e_out[1], e_w[1], True, 2)
    p = conv_f16x3(h2, w3, b3, e_in[2], None, e_w[2], False, 2)
    return p, dict(x0=x0, h1=h1, h2=h2)
