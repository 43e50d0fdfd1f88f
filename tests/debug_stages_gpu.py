#!/usr/bin/env python3
"""Stage-by-stage diagnostic of the HIP path against the oracle on one golden fixture (GPU box).
Prints max abs errors of: class maps, sumsq, corr, normalised corr, conv1, conv2, params, final outputs."""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import util  # noqa: E402
from oracle import head_oracle as O  # noqa: E402
from os2d_amd import _lib  # noqa: E402


def unpad(x, NB, Cst, H, W, plane):
    """plane layout (os2d_common.h): cell(h,w) = BASE + h*(W+3) + w, BASE = round_up(3*(W+3)+3, 4)."""
    Ws = W + 3
    base = (3 * Ws + 3 + 3) // 4 * 4
    x = x.view(NB, Cst, plane)
    data = x[:, :, base:base + H * Ws].reshape(NB, Cst, H, Ws)
    inner = data[:, :, :, :W].contiguous()
    border = x.clone()
    border[:, :, base:base + H * Ws].view(NB, Cst, H, Ws)[:, :, :, :W] = 0
    return inner, float(border.abs().max())


def synthetic_fixture(C, H, W, B, P=6, inverse=True):
    """Fixture-like dict built from the oracle for an arbitrary shape: 'C,H,W,B' on the command line."""
    from os2d_amd.utils import synthetic
    state = synthetic.make_transform_net_state(P, seed=31)
    fm = synthetic.make_feature_map(C, H, W, seed=41)
    class_fms = synthetic.make_class_feature_maps(B, C, sizes=[(15, 15), (11, 19)], seed=900)
    q = O.prepare_class_maps(class_fms)
    with torch.no_grad():
        loc, cls, _, corners, mid = O.head_forward(fm, q, state, inverse, return_intermediate=True)
    return dict(P=P, inverse=inverse, state=state, fm=fm, class_fms=class_fms, ref_q15=q, ref_corr=mid["corr"],
                ref_params=mid["params"], ref_loc=loc, ref_cls=cls, ref_corners=corners)


def main(name):
    dev = torch.device("cuda:0")
    lib = _lib.load()
    fx = synthetic_fixture(*[int(v) for v in name.split(",")]) if "," in name else util.load_head_fixture(name)
    P, inverse, state = fx["P"], fx["inverse"], fx["state"]
    fm = fx["fm"].to(dev)
    A, C, H, W = fm.shape
    creator = util.make_head_creator(P, inverse, state, dev)
    head = creator.create_os2d_head([c.to(dev) for c in fx["class_fms"]])
    B = head.class_batch_size
    NB = A * B
    st = _lib.current_stream(dev)
    print("fixture", name, "A,B,C,H,W,P,inv =", A, B, C, H, W, P, inverse)
    print("q15       ", util.maxdiff(head.class_feature_maps, fx["ref_q15"]))
    q_or = O.prepare_class_maps(fx["class_fms"])
    qp_ref = torch.zeros(B, C, 256)
    qp_ref[:, :, :225] = q_or.permute(0, 1, 3, 2).reshape(B, C, 225)
    print("qp        ", util.maxdiff(head._qp, qp_ref))

    HW = H * W
    plane = lib.os2d_plane_floats(H, W)
    sumsq = torch.empty(A, HW, device=dev)
    _lib.check(lib.os2d_fm_sumsq(_lib.ptr(fm), _lib.ptr(sumsq), A, C, H, W, st), "sumsq")
    print("sumsq     ", util.maxdiff(sumsq, fx["fm"].pow(2).sum(1).view(A, HW)))
    corr = torch.empty(NB, 225, HW, device=dev)
    rpad = torch.full((NB * 226 * plane,), float("nan"), device=dev)
    _lib.check(lib.os2d_corr(_lib.ptr(fm), _lib.ptr(head._qp), _lib.ptr(sumsq), _lib.ptr(corr), _lib.ptr(rpad), A, B, C, H, W, st), "corr")
    print("corr      ", util.maxdiff(corr.view(NB, 225, H, W), fx["ref_corr"]))
    rn_ref = O.l2_normalize_channels(F.relu(fx["ref_corr"]), 1e-6)
    rn, bmax = unpad(rpad, NB, 226, H, W, plane)
    print("rnorm     ", util.maxdiff(rn[:, :225], rn_ref), "pad-ch", float(rn[:, 225].abs().max()), "border", bmax)

    w1, b1, w2, b2, w3, b3 = creator.aligner.parameter_regressor.packed("f32")[:6]
    fw = O.fold_batchnorm(state, torch.float32)
    h1 = torch.full((NB * 128 * plane,), float("nan"), device=dev)
    _lib.check(lib.os2d_transform_conv(1, _lib.ptr(rpad), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(h1), NB, P, H, W, st), "conv1")
    h1_ref = F.relu(F.conv2d(rn_ref, fw[0], fw[1], padding=3))
    g, bmax = unpad(h1, NB, 128, H, W, plane)
    print("conv1     ", util.maxdiff(g, h1_ref), "border", bmax, "ref absmax", float(h1_ref.abs().max()))
    h2 = torch.full((NB * 64 * plane,), float("nan"), device=dev)
    _lib.check(lib.os2d_transform_conv(2, _lib.ptr(h1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(h2), NB, P, H, W, st), "conv2")
    h2_ref = F.relu(F.conv2d(h1_ref, fw[2], fw[3], padding=2))
    g, bmax = unpad(h2, NB, 64, H, W, plane)
    print("conv2     ", util.maxdiff(g, h2_ref), "border", bmax, "ref absmax", float(h2_ref.abs().max()))
    params = torch.full((NB, P, HW), float("nan"), device=dev)
    _lib.check(lib.os2d_transform_conv(3, _lib.ptr(h2), _lib.ptr(w3), _lib.ptr(b3), _lib.ptr(params), NB, P, H, W, st), "conv3")
    print("params    ", util.maxdiff(params.view(NB, P, H, W), fx["ref_params"]))
    loc = torch.empty(NB, 4, HW, device=dev)
    cls = torch.empty(NB, 1, HW, device=dev)
    corners = torch.empty(NB, 8, HW, device=dev)
    # feed the REFERENCE params/corr to isolate the sampler
    _lib.check(lib.os2d_sample_decode(_lib.ptr(fx["ref_corr"].to(dev).contiguous()), _lib.ptr(fx["ref_params"].to(dev).contiguous()),
                                      NB, H, W, P, int(inverse), 16, 16, _lib.ptr(loc), _lib.ptr(cls), _lib.ptr(corners), st), "sample")
    print("sampler(ref in): cls", util.maxdiff(cls.view_as(fx["ref_cls"]), fx["ref_cls"]),
          "loc", util.maxdiff(loc.view_as(fx["ref_loc"]), fx["ref_loc"]),
          "corners", util.maxdiff(corners.view_as(fx["ref_corners"]), fx["ref_corners"]))
    with torch.no_grad():
        l2, c2, _, k2 = head(fm)
    print("head      : cls", util.maxdiff(c2, fx["ref_cls"]), "loc", util.maxdiff(l2, fx["ref_loc"]),
          "corners", util.maxdiff(k2, fx["ref_corners"]))
    torch.cuda.synchronize()


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["v2_affine_inv", "v1_simple"]):
        main(n)
