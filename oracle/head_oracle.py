"""CPU ORACLE for the OS2D correlation + alignment head.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped path (``os2d_amd``) never does and
raises if the HIP library is missing.

It restates, with stock PyTorch CPU operators in fp32 (fp64 inside the resampler, as the reference),
what ``Os2dHead.forward`` of aosokin/os2d computes in eval mode.  Every function cites the reference
lines it follows (paths relative to the reference checkout).

Parity pinning: the reference has no tests.  The oracle is pinned against outputs
of the reference itself run in the development container: ``tests/golden/make_golden.py`` imports the
reference and records inputs + outputs; ``tests/test_oracle_golden.py`` checks this file against them
(head outputs, correlation tensor, TransformNet parameters).  The torchvision pieces on the path
(``encode_boxes``) are restated from torchvision's published closed form -- see DESIGN.md.

Two statements are provided:
  * ``head_forward``            -- operator-for-operator twin of the reference (materialised sampling
                                   grids, float64 grid_sample).  This is what ``cpu_baseline`` times.
  * ``head_forward_closed_form`` -- the closed form of SURVEY.md appendix A evaluated in float64
                                   (no 15x15x2 grid tensor, 121 taps, 4 corners, analytic inverse): an
                                   independent derivation used as high-precision truth for error budgets.
"""

import torch
import torch.nn.functional as F

TEMPLATE = 15            # reference head.py:66-69 (out_grid_size = reference_feature_map_size = 15x15)
POOL_BORDER = 2          # reference head.py:280 (pool_border_width)
LOC_WEIGHTS = (10.0, 10.0, 5.0, 5.0)   # reference box_coder.py:13
BN_EPS = 1e-5            # torch.nn.BatchNorm2d default used at reference head.py:623


def l2_normalize_channels(x, eps):
    """reference head.py:597-601 : x / (||x||_2 over dim 1 + eps)."""
    return x / (x.norm(dim=1, keepdim=True) + eps)


def resize_class_maps(class_feature_maps, size=TEMPLATE):
    """reference head.py:241-259 : bilinear resize of each [1,C,h,w] class map to [1,C,15,15]
    (identity affine grid, align_corners=True, zero padding), concatenated to [B,C,15,15]."""
    out = []
    for fm in class_feature_maps:
        assert fm.size(0) == 1
        theta = torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]], dtype=fm.dtype, device=fm.device)
        grid = F.affine_grid(theta, [1, fm.size(1), size, size], align_corners=True)
        out.append(F.grid_sample(fm, grid, mode="bilinear", padding_mode="zeros", align_corners=True))
    return torch.cat(out, dim=0)


def prepare_class_maps(class_feature_maps):
    """resize + L2 normalise (eps 1e-5): reference head.py:261-268 and :293."""
    return l2_normalize_channels(resize_class_maps(class_feature_maps), 1e-5)


def pool_mask(size=TEMPLATE, border=POOL_BORDER):
    """reference head.py:296-302, :196-201 : 1 on the inner (15-2*2)^2 window, normalised to sum 1."""
    m = torch.zeros(size, size)
    m[border:size - border, border:size - border] = 1
    return m / m.sum()


def correlation(q_hat, fm):
    """reference head.py:339-350 : normalise the image map (eps 1e-5) and correlate all-to-all.
    Output channel index is x_T * 15 + y_T (the reference's "abwhxy" ordering)."""
    f_hat = l2_normalize_channels(fm, 1e-5)
    corr = torch.einsum("bchw,acxy->abwhxy", q_hat, f_hat)
    A, B = fm.size(0), q_hat.size(0)
    return corr.contiguous().view(A * B, TEMPLATE * TEMPLATE, fm.size(2), fm.size(3))


def transform_net(corr, state):
    """reference head.py:604-655 : relu -> L2 over the 225 channels (eps 1e-6) -> conv7+BN+ReLU ->
    conv5+BN+ReLU -> conv5.  ``state`` uses the reference's state-dict keys."""
    x = l2_normalize_channels(F.relu(corr), 1e-6)
    x = F.conv2d(x, state["conv.0.weight"], state["conv.0.bias"], padding=3)
    x = F.batch_norm(x, state["conv.1.running_mean"], state["conv.1.running_var"],
                     state["conv.1.weight"], state["conv.1.bias"], training=False, eps=BN_EPS)
    x = F.relu(x)
    x = F.conv2d(x, state["conv.3.weight"], state["conv.3.bias"], padding=2)
    x = F.batch_norm(x, state["conv.4.running_mean"], state["conv.4.running_var"],
                     state["conv.4.weight"], state["conv.4.bias"], training=False, eps=BN_EPS)
    x = F.relu(x)
    return F.conv2d(x, state["linear.weight"], state["linear.bias"], padding=2)


def params_to_theta(params, inverse):
    """reference head.py:81-153 : [N,P,H,W] -> theta [N*H*W, 2, 3] ordered (n,h,w); P=6 full affine,
    P=4 scale+translation; optional inversion of the homogeneous 3x3 matrix."""
    P = params.size(1)
    flat = params.permute(0, 2, 3, 1).reshape(-1, P)
    if P == 6:
        theta = flat.view(-1, 2, 3)
    elif P == 4:
        z = torch.zeros_like(flat[:, 0])
        theta = torch.stack([flat[:, 0], z, flat[:, 1], z, flat[:, 2], flat[:, 3]], dim=1).view(-1, 2, 3)
    else:
        raise RuntimeError("P must be 6 or 4")
    if inverse:
        last = torch.zeros(theta.size(0), 1, 3, dtype=theta.dtype)
        last[:, :, 2] = 1
        theta = torch.inverse(torch.cat([theta, last], dim=1))[:, :2, :].contiguous()
    return theta


def anchor_grid(H, W, box, stride):
    """reference box_coder.py:17-59 : row-major (h*W+w) xyxy anchors, centre ((w+.5)*stride, (h+.5)*stride)."""
    cy = (torch.arange(H, dtype=torch.float32) + 0.5) * stride
    cx = (torch.arange(W, dtype=torch.float32) + 0.5) * stride
    cx = cx.view(1, W).expand(H, W).reshape(-1)
    cy = cy.view(H, 1).expand(H, W).reshape(-1)
    half = box / 2.0
    # the reference converts cx_cy_w_h -> xyxy as (cx - w/2, cy - h/2, cx + w/2, cy + h/2)
    return torch.stack([cx - half, cy - half, cx + half, cy + half], dim=1)


def local_to_global(grids, boxes):
    """reference head.py:18-40 : x_g = (x2-x1)/2 * x_l + (x1+x2)/2 (same for y); boxes broadcast [..,H,W,4]."""
    ax = ((boxes[..., 2] - boxes[..., 0]) / 2)[..., None, None]
    bx = ((boxes[..., 2] + boxes[..., 0]) / 2)[..., None, None]
    ay = ((boxes[..., 3] - boxes[..., 1]) / 2)[..., None, None]
    by = ((boxes[..., 3] + boxes[..., 1]) / 2)[..., None, None]
    return torch.stack([grids[..., 0] * ax + bx, grids[..., 1] * ay + by], dim=-1)


def resample_and_pool(corr, grids_unit, mask):
    """reference head.py:439-520 (``resample_of_correlation_map_fast``).
    corr [A,B,225,H,W]; grids_unit [A,B,H,W,15,15,2] in [-1,1] feature-map coords; mask [15,15].
    The channel index is folded into the Y coordinate of one tall [225*H, W] image and sampled in fp64."""
    A, B, K, H, W = corr.shape
    tall = corr.reshape(A * B, 1, K * H, W).double()
    g = grids_unit.clamp(-1, 1).double()
    gx = g[..., 0]
    gy = (g[..., 1] + 1) / 2 * (H - 1)
    yy, xx = torch.meshgrid(torch.arange(TEMPLATE), torch.arange(TEMPLATE), indexing="ij")
    chan = (yy + xx * TEMPLATE).double()                     # channel of template point (row yy, col xx)
    gy = (gy + chan * H) / (H * K - 1) * 2 - 1
    pts = torch.stack([gx, gy], dim=-1).view(A * B, -1, 1, 2)
    vals = F.grid_sample(tall, pts, mode="bilinear", padding_mode="border", align_corners=True)
    vals = vals.view(A, B, H * W, K).float()
    return (vals * mask.reshape(1, 1, 1, K)).sum(-1).view(A, B, 1, H, W)


def encode_boxes(boxes, anchors, weights=LOC_WEIGHTS):
    """torchvision ``encode_boxes`` (published closed form), called at reference box_coder.py:316."""
    ew, eh = anchors[:, 2] - anchors[:, 0], anchors[:, 3] - anchors[:, 1]
    ecx, ecy = anchors[:, 0] + 0.5 * ew, anchors[:, 1] + 0.5 * eh
    gw, gh = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    gcx, gcy = boxes[:, 0] + 0.5 * gw, boxes[:, 1] + 0.5 * gh
    return torch.stack([weights[0] * (gcx - ecx) / ew, weights[1] * (gcy - ecy) / eh,
                        weights[2] * torch.log(gw / ew), weights[3] * torch.log(gh / eh)], dim=1)


def clip_to_min_size(b, min_size=1.0):
    """reference bounding_box.py:267-277."""
    b = b.clone()
    m = (b[:, 0] + min_size) > b[:, 2]
    b[m, 2] = b[m, 0] + min_size
    m = (b[:, 1] + min_size) > b[:, 3]
    b[m, 3] = b[m, 1] + min_size
    return b


def head_forward(fm, q_hat, state, inverse, stride=16, rec_field=16, return_intermediate=False):
    """Operator-level twin of reference head.py:308-435 (eval mode).

    fm [A,C,H,W]; q_hat [B,C,15,15] (already resized + L2-normalised, see ``prepare_class_maps``);
    ``state`` TransformNet state dict; P is inferred from ``linear.weight``.
    Returns loc [A,B,4,H,W], cls [A,B,1,H,W], cls (again; detached alias in eval), corners [A,B,8,H,W].
    """
    A, C, H, W = fm.shape
    B = q_hat.size(0)
    corr = correlation(q_hat, fm)                                            # head.py:339-350
    params = transform_net(corr, state)                                      # head.py:176
    theta = params_to_theta(params, inverse)                                 # head.py:179
    grids = F.affine_grid(theta, [theta.size(0), 1, TEMPLATE, TEMPLATE], align_corners=True)  # head.py:184
    grids = grids.view(A, B, H, W, TEMPLATE, TEMPLATE, 2)
    # feature-map level anchors: box 15, stride 1 (head.py:68-69,219-220)
    boxes_fm = anchor_grid(H, W, float(TEMPLATE), 1.0).view(1, 1, H, W, 4)
    g_fm = local_to_global(grids, boxes_fm)                                  # head.py:371-376
    g_unit = torch.stack([g_fm[..., 0] / (W - 1) * 2 - 1, g_fm[..., 1] / (H - 1) * 2 - 1], dim=-1).clamp(-1, 1)
    cls = resample_and_pool(corr.view(A, B, TEMPLATE * TEMPLATE, H, W), g_unit, pool_mask())  # head.py:393
    # image level anchors: box stride*(15-1)+rf, stride (head.py:223-238)
    box_img = float(stride * (TEMPLATE - 1) + rec_field)
    anchors = anchor_grid(H, W, box_img, float(stride))
    g_img = local_to_global(grids, anchors.view(1, 1, H, W, 4))              # head.py:410
    gx = g_img[..., 0].reshape(-1, TEMPLATE * TEMPLATE)
    gy = g_img[..., 1].reshape(-1, TEMPLATE * TEMPLATE)
    boxes = torch.stack([gx.min(1)[0], gy.min(1)[0], gx.max(1)[0], gy.max(1)[0]], dim=1)   # head.py:416-419
    corners = g_img[:, :, :, :, [0, -1]][:, :, :, :, :, [0, -1]].reshape(A, B, H, W, 8)    # head.py:422-425
    corners = corners.permute(0, 1, 4, 2, 3).contiguous()
    all_anchors = anchors.repeat(A * B, 1)
    loc = encode_boxes(clip_to_min_size(boxes), clip_to_min_size(all_anchors))             # box_coder.py:306-317
    loc = loc.view(A, B, H, W, 4).permute(0, 1, 4, 2, 3).contiguous()
    if return_intermediate:
        return loc, cls, cls, corners, dict(corr=corr, params=params)
    return loc, cls, cls, corners


def head_forward_looped(fm, q_hat, state, inverse, **kw):
    """The way the reference's evaluation drives the head: one class at a time
    (reference os2d/engine/evaluate.py:323-331, class_batch_size == 1), outputs concatenated."""
    outs = [head_forward(fm, q_hat[b:b + 1], state, inverse, **kw) for b in range(q_hat.size(0))]
    return tuple(torch.cat([o[i] for o in outs], dim=1) for i in range(4))


# ----------------------------------------------------------------------------- closed form (fp64 truth)
def fold_batchnorm(state, dtype=torch.float64):
    """Fold eval-mode BatchNorm into the preceding convolution: w' = w*g/sqrt(var+eps), b' = (b-mean)*g/sqrt(var+eps)+beta."""
    out = []
    for conv, bn in (("conv.0", "conv.1"), ("conv.3", "conv.4")):
        s = state[bn + ".weight"].to(dtype) / torch.sqrt(state[bn + ".running_var"].to(dtype) + BN_EPS)
        w = state[conv + ".weight"].to(dtype) * s.view(-1, 1, 1, 1)
        b = (state[conv + ".bias"].to(dtype) - state[bn + ".running_mean"].to(dtype)) * s + state[bn + ".bias"].to(dtype)
        out += [w, b]
    out += [state["linear.weight"].to(dtype), state["linear.bias"].to(dtype)]
    return out


def head_forward_closed_form(fm, q_hat, state, inverse, stride=16, rec_field=16, dtype=torch.float64,
                             return_intermediate=False):
    """SURVEY.md appendix A evaluated directly (default float64).  Independent of ``head_forward``:
    no grid tensors, no grid_sample, analytic 2x2 inverse, box from the 4 corner points."""
    fm = fm.to(dtype)
    q_hat = q_hat.to(dtype)
    A, C, H, W = fm.shape
    B = q_hat.size(0)
    T = TEMPLATE
    f_hat = fm / (fm.pow(2).sum(1, keepdim=True).sqrt() + 1e-5)
    # corr[a,b,k=j*15+i,h,w] = sum_c q[b,c,i,j] f[a,c,h,w]
    qk = q_hat.permute(0, 3, 2, 1).reshape(B, T * T, C)          # [b, j*15+i, c]
    corr = torch.einsum("bkc,achw->abkhw", qk, f_hat)
    r = corr.clamp(min=0)
    r = r / (r.pow(2).sum(2, keepdim=True).sqrt() + 1e-6)
    w1, b1, w2, b2, w3, b3 = fold_batchnorm(state, dtype)
    x = r.reshape(A * B, T * T, H, W)
    x = F.relu(F.conv2d(x, w1, b1, padding=3))
    x = F.relu(F.conv2d(x, w2, b2, padding=2))
    p = F.conv2d(x, w3, b3, padding=2)                            # [A*B, P, H, W]
    P = p.size(1)
    if P == 6:
        t00, t01, t02, t10, t11, t12 = [p[:, i] for i in range(6)]
    else:
        zero = torch.zeros_like(p[:, 0])
        t00, t01, t02, t10, t11, t12 = p[:, 0], zero, p[:, 1], zero, p[:, 2], p[:, 3]
    if inverse:
        det = t00 * t11 - t01 * t10
        i00, i01, i10, i11 = t11 / det, -t01 / det, -t10 / det, t00 / det
        i02 = -(i00 * t02 + i01 * t12)
        i12 = -(i10 * t02 + i11 * t12)
        t00, t01, t02, t10, t11, t12 = i00, i01, i02, i10, i11, i12
    hh = torch.arange(H, dtype=dtype).view(1, H, 1)
    ww = torch.arange(W, dtype=dtype).view(1, 1, W)
    corr_n = corr.reshape(A * B, T * T, H * W)
    score = torch.zeros(A * B, H, W, dtype=dtype)
    lo, hi = POOL_BORDER, T - POOL_BORDER
    for i in range(lo, hi):
        yi = -1 + 2.0 * i / (T - 1)
        for j in range(lo, hi):
            xj = -1 + 2.0 * j / (T - 1)
            gx = t00 * xj + t01 * yi + t02
            gy = t10 * xj + t11 * yi + t12
            X = (ww + 0.5 + (T / 2.0) * gx).clamp(0, W - 1)
            Y = (hh + 0.5 + (T / 2.0) * gy).clamp(0, H - 1)
            x0 = X.floor().clamp(max=W - 1)
            y0 = Y.floor().clamp(max=H - 1)
            x1 = (x0 + 1).clamp(max=W - 1)
            y1 = (y0 + 1).clamp(max=H - 1)
            fx, fy = X - x0, Y - y0
            ch = corr_n[:, j * T + i]                                  # [A*B, H*W]

            def at(yy, xx):
                return torch.gather(ch, 1, (yy.long() * W + xx.long()).view(A * B, -1)).view(A * B, H, W)

            score += (at(y0, x0) * (1 - fx) * (1 - fy) + at(y0, x1) * fx * (1 - fy)
                      + at(y1, x0) * (1 - fx) * fy + at(y1, x1) * fx * fy)
    score = score / float((hi - lo) ** 2)
    half = (stride * (T - 1) + rec_field) / 2.0
    ecx = stride * (ww + 0.5)
    ecy = stride * (hh + 0.5)
    us, vs = [], []
    for yi in (-1.0, 1.0):        # template row 0, 14
        for xj in (-1.0, 1.0):    # template col 0, 14
            us.append(half * (t00 * xj + t01 * yi + t02) + ecx)
            vs.append(half * (t10 * xj + t11 * yi + t12) + ecy)
    U, V = torch.stack(us, 0), torch.stack(vs, 0)
    x1, x2, y1, y2 = U.min(0)[0], U.max(0)[0], V.min(0)[0], V.max(0)[0]
    x2 = torch.where(x1 + 1 > x2, x1 + 1, x2)
    y2 = torch.where(y1 + 1 > y2, y1 + 1, y2)
    bw, bh = x2 - x1, y2 - y1
    size = 2 * half
    loc = torch.stack([LOC_WEIGHTS[0] * (x1 + 0.5 * bw - ecx) / size, LOC_WEIGHTS[1] * (y1 + 0.5 * bh - ecy) / size,
                       LOC_WEIGHTS[2] * torch.log(bw / size), LOC_WEIGHTS[3] * torch.log(bh / size)], dim=1)
    corners = torch.stack([U[0], V[0], U[1], V[1], U[2], V[2], U[3], V[3]], dim=1)
    out = (loc.view(A, B, 4, H, W), score.view(A, B, 1, H, W), score.view(A, B, 1, H, W), corners.view(A, B, 8, H, W))
    if return_intermediate:
        return out + (dict(corr=corr.reshape(A * B, T * T, H, W), params=p),)
    return out
