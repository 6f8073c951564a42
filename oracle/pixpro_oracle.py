"""CPU restatement of the PixPro+optical-flow hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: a plain torch-CPU (fp32) restatement of the algorithm of
rioyokotalab/PixPro-with-OpticalFlow for the one path this repository accelerates.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / timed CPU baseline.  The product
(``pixpro-with-opticalflow_amd/``) never imports it and has no CPU fallback.

Pinning: the reference has no tests or golden vectors of its own, so the
oracle is pinned by vectors captured from the reference's own Python, imported in the
build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and checked
in ``tests/test_oracle_golden.py``.  All arithmetic of the reference lives in torch
(pinned by the reference to 1.8.2, here 2.10.0): the samplers below restate ATen's
``grid_sampler_2d`` / ``upsample_bilinear2d`` CPU formulas explicitly (no call to
``F.grid_sample`` / ``F.interpolate``) so that the HIP kernels can follow the same
fp32 operation order.

Every function cites the reference lines (relative to /root/reference) it follows.
"""
import math

import torch

EPS_NORM = 1e-12  # F.normalize default eps

# The explicit samplers below are what parity is defined on.  For *timing* the CPU baseline (bench.py) the
# same two operations can be delegated to the ATen kernels the reference itself calls (F.grid_sample,
# F.interpolate): identical results on this build (tests/test_oracle_golden.py::test_aten_samplers_agree), and
# the speed of the reference's own CPU path rather than of the gather-based restatement.
_ATEN_SAMPLERS = False


class aten_samplers:
    """with aten_samplers(): ... -> bilinear_sample / upflow8 call torch's own CPU kernels."""

    def __enter__(self):
        global _ATEN_SAMPLERS
        self.prev, _ATEN_SAMPLERS = _ATEN_SAMPLERS, True

    def __exit__(self, *exc):
        global _ATEN_SAMPLERS
        _ATEN_SAMPLERS = self.prev


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def l2_normalize(t, dim=1):
    """F.normalize(t, dim=1): t / max(||t||_2, 1e-12).  contrast/models/PixPro.py:344,348,380"""
    n = t.pow(2).sum(dim, keepdim=True).sqrt().clamp_min(EPS_NORM)
    return t / n


def _fma(a, b, c):
    """fp32 fused multiply-add a*b+c (single rounding), emulated through fp64: the product of
    two fp32 values is exact in fp64.  fp64 inputs (used by some tests) fall back to a*b+c."""
    if a.dtype != torch.float32:
        return a * b + c
    return (a.double() * b.double() + c.double()).float()


def _gather_zeros(img, iy, ix):
    """img [N,C,Hf,Wf]; iy, ix integer tensors [N,...] -> values [N,C,...], 0 outside."""
    N, C, Hf, Wf = img.shape
    inb = (ix >= 0) & (ix < Wf) & (iy >= 0) & (iy < Hf)
    lin = (iy.clamp(0, Hf - 1) * Wf + ix.clamp(0, Wf - 1)).reshape(N, 1, -1).expand(N, C, -1)
    val = img.reshape(N, C, Hf * Wf).gather(2, lin).reshape(N, C, *ix.shape[1:])
    return val * inb.unsqueeze(1).to(img.dtype)


def bilinear_sample(img, gx, gy):
    """grid_sample(img, stack(gx,gy), mode='bilinear', padding_mode='zeros', align_corners=True).

    img [N,C,Hf,Wf]; gx, gy [N,...] normalised coordinates in [-1,1].  ATen CPU operation
    order (x86 build with FMA contraction, verified bit-for-bit against torch 2.10 CPU):
    ix=(gx+1)*((Wf-1)/2); w=ix-floor(ix); e=1-w; n=iy-floor(iy); s=1-n;
    out = fma(se_val, n*w, fma(sw_val, n*e, fma(ne_val, s*w, nw_val*(s*e)))).
    Used at contrast/models/PixPro.py:64 and contrast/util.py:278,318,322.
    """
    if _ATEN_SAMPLERS:
        lead = gx.shape[1:]
        grid = torch.stack([gx, gy], -1).reshape(gx.shape[0], 1, -1, 2)
        out = torch.nn.functional.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        return out.reshape(img.shape[0], img.shape[1], *lead)
    Hf, Wf = img.shape[-2:]
    ix = (gx + 1) * ((Wf - 1) / 2)
    iy = (gy + 1) * ((Hf - 1) / 2)
    x_w = ix.floor()
    y_n = iy.floor()
    w = ix - x_w
    e = 1 - w
    n = iy - y_n
    s = 1 - n
    nw, ne, sw, se = s * e, s * w, n * e, n * w
    x0 = x_w.long()
    y0 = y_n.long()
    v_nw = _gather_zeros(img, y0, x0)
    v_ne = _gather_zeros(img, y0, x0 + 1)
    v_sw = _gather_zeros(img, y0 + 1, x0)
    v_se = _gather_zeros(img, y0 + 1, x0 + 1)
    acc = v_nw * nw.unsqueeze(1)
    acc = _fma(v_ne, ne.unsqueeze(1), acc)
    acc = _fma(v_sw, sw.unsqueeze(1), acc)
    return _fma(v_se, se.unsqueeze(1), acc)


def nearest_sample(img, gx, gy):
    """grid_sample(..., mode='nearest', padding_mode='zeros', align_corners=True):
    index = nearbyint(unnormalised coordinate) (round-half-even).  PixPro.py:68-69."""
    Hf, Wf = img.shape[-2:]
    ix = (gx + 1) * ((Wf - 1) / 2)
    iy = (gy + 1) * ((Hf - 1) / 2)
    return _gather_zeros(img, torch.round(iy).long(), torch.round(ix).long())


# --------------------------------------------------------------------------------------
# K1  Pixel Propagation Module          contrast/models/PixPro.py:339-363 (+ :380)
# --------------------------------------------------------------------------------------
def featprop(x, v, p=2.0, clamp=0.0):
    """x [N,C,H,W] projector output, v = value_transform(x) [N,C,H,W] -> Y [N,C,H,W].

    v_hat = normalize(v); x_hat = normalize(x); S = x_hat^T x_hat; A = clamp(S, min=clamp);
    if p < 1: A += 1e-6; A = A**p; Y = v_hat @ A^T.           PixPro.py:343-363
    """
    N, C, H, W = x.shape
    vh = l2_normalize(v).reshape(N, C, -1)
    xh = l2_normalize(x).reshape(N, C, -1)
    S = torch.bmm(xh.transpose(1, 2), xh)
    A = torch.clamp(S, min=clamp)
    if p < 1.0:
        A = A + 1e-6
    A = A ** p
    Y = torch.bmm(vh, A.transpose(1, 2))
    return Y.reshape(N, C, H, W)


# --------------------------------------------------------------------------------------
# K4  add_optical_flow                   contrast/models/PixPro.py:46-89
# --------------------------------------------------------------------------------------
def add_optical_flow(flow, x_grid, y_grid, size, mask=None):
    """flow [N,2,Hf,Wf] (pixels of the flow map), x_grid/y_grid [N,H,W] (original-frame
    pixels), size=(Ho,Wo), mask [N,Hf,Wf] bool or None -> out_x, out_y [N,H,W], mask_grid
    [N,1,H,W] bool or None."""
    Hf, Wf = flow.shape[-2:]
    Ho, Wo = size
    gx = 2 * (x_grid / (Wo - 1)) - 1                       # :61
    gy = 2 * (y_grid / (Ho - 1)) - 1                       # :62
    f = bilinear_sample(flow, gx, gy)                      # :64
    mask_grid = None
    if mask is not None:                                   # :65-70
        mask_grid = nearest_sample(mask.unsqueeze(1).float(), gx, gy).to(torch.bool)
    if Hf != Ho or Wf != Wo:                               # :49-51,76-80
        rh, rw = Hf / Ho, Wf / Wo
        out_x = (x_grid * rw + f[:, 0]) / rw
        out_y = (y_grid * rh + f[:, 1]) / rh
    else:                                                  # :81-83
        out_x = x_grid + f[:, 0]
        out_y = y_grid + f[:, 1]
    return out_x, out_y, mask_grid


# --------------------------------------------------------------------------------------
# K2  regression_loss                    contrast/models/PixPro.py:92-247
# --------------------------------------------------------------------------------------
def unpack_coord_arg(coord_q, coord_k):
    """The reference's argument polymorphism (PixPro.py:99-124), debug tuple form excluded.
    Returns coord_q[N,10], coord_k[N,10], flow or None, size (Ho,Wo), mask or None."""
    flow = mask = None
    if isinstance(coord_q, list):
        coord_q, flow = coord_q
        coord_k, _ = coord_k
        if isinstance(flow, list):
            flow, size, mask = flow
            if isinstance(size, torch.Tensor):
                size = (int(size[0].item()), int(size[1].item()))
            if isinstance(mask, list):
                mask = mask[0]
        else:
            size = tuple(flow.shape[-2:])
    else:
        size = (coord_q[0][9].item(), coord_q[0][8].item())
    return coord_q, coord_k, flow, size, mask


def positive_mask(coord_q, coord_k, H, W, size, pos_ratio, flow=None, mask=None):
    """Boolean positive-pair matrix M [N,P,P] (P=H*W).  PixPro.py:132-223."""
    Ho, Wo = size
    dt = coord_q.dtype
    xs = torch.arange(0., float(W), dtype=dt).view(1, 1, -1).repeat(1, H, 1)      # :132
    ys = torch.arange(0., float(H), dtype=dt).view(1, -1, 1).repeat(1, 1, W)      # :133
    q_bw = ((coord_q[:, 2] - coord_q[:, 0]) / W).view(-1, 1, 1)                   # :140-143
    q_bh = ((coord_q[:, 3] - coord_q[:, 1]) / H).view(-1, 1, 1)
    k_bw = ((coord_k[:, 2] - coord_k[:, 0]) / W).view(-1, 1, 1)
    k_bh = ((coord_k[:, 3] - coord_k[:, 1]) / H).view(-1, 1, 1)
    q_x0, q_y0 = coord_q[:, 0].view(-1, 1, 1), coord_q[:, 1].view(-1, 1, 1)        # :145-148
    k_x0, k_y0 = coord_k[:, 0].view(-1, 1, 1), coord_k[:, 1].view(-1, 1, 1)
    q_diag = torch.sqrt((q_bw * (Wo - 1)) ** 2 + (q_bh * (Ho - 1)) ** 2)          # :155-157
    k_diag = torch.sqrt((k_bw * (Wo - 1)) ** 2 + (k_bh * (Ho - 1)) ** 2)
    max_diag = torch.max(q_diag, k_diag)
    qx = ((xs + 0.5) * q_bw + q_x0) * (Wo - 1)                                    # :168-175 / :192-199
    qy = ((ys + 0.5) * q_bh + q_y0) * (Ho - 1)
    kx = ((xs + 0.5) * k_bw + k_x0) * (Wo - 1)
    ky = ((ys + 0.5) * k_bh + k_y0) * (Ho - 1)
    mask_grid = None
    if flow is not None:                                                           # :200-201
        qx, qy, mask_grid = add_optical_flow(flow, qx, qy, size, mask)
    P = H * W
    dist = torch.sqrt((qx.reshape(-1, P, 1) - kx.reshape(-1, 1, P)) ** 2
                      + (qy.reshape(-1, P, 1) - ky.reshape(-1, 1, P)) ** 2) / max_diag   # :217-218
    M = dist < pos_ratio                                                           # :219
    if mask_grid is not None:                                                      # :220-222
        M = M & mask_grid.reshape(-1, P, 1)
    return M


def regression_loss(q, k, coord_q, coord_k, pos_ratio=0.5):
    """Same call forms as the reference (Tensor coords, or [coords, flow] /
    [coords, [flow, size, mask]] lists).  Returns (loss, [pos_num[N], pos_mean[N]]).
    PixPro.py:92-247."""
    N, C, H, W = q.shape
    coord_q, coord_k, flow, size, mask = unpack_coord_arg(coord_q, coord_k)
    M = positive_mask(coord_q, coord_k, H, W, size, pos_ratio, flow, mask).float()
    qf = q.reshape(N, C, -1)
    kf = k.reshape(N, C, -1)
    logit = torch.bmm(qf.transpose(1, 2), kf)                                      # :239
    loss = (logit * M).sum(-1).sum(-1) / (M.sum(-1).sum(-1) + 1e-6)                # :241
    pos_num = M.sum(-1).sum(-1)                                                    # :243-245
    pos_mean = M.mean(-1).mean(-1)
    return -2 * loss.mean(), [pos_num, pos_mean]


# --------------------------------------------------------------------------------------
# K1+K2(+K4) fused, with the analytic backward K3
# --------------------------------------------------------------------------------------
def ppm_loss(x, v, k, coord_q, coord_k, p, clamp, pos_ratio, normalize_k=False):
    """One loss direction exactly as PixPro.forward composes it (PixPro.py:376-380,429):
    pred = normalize(featprop(x)); loss = regression_loss(pred, k, coord_q, coord_k).
    Returns loss (scalar, differentiable w.r.t. x and v), pred, pos_num, pos_mean."""
    if normalize_k:
        k = l2_normalize(k)                                                        # :403,407
    pred = l2_normalize(featprop(x, v, p, clamp))
    loss, (pos_num, pos_mean) = regression_loss(pred, k, coord_q, coord_k, pos_ratio)
    return loss, pred, pos_num, pos_mean


def ppm_loss_grad_analytic(x, v, k, M, p, clamp, upstream=1.0):
    """Closed-form d(loss)/dx, d(loss)/dv for loss = -2*mean_n(<Yhat, K M^T>/(cnt+1e-6))
    (x-path only: the v = value_transform(x) chain rule is left to the caller).
    x, v, k [N,C,P]; M [N,P,P] float.  Returns loss, pred, dx, dv."""
    N, C, P = x.shape
    nx = x.pow(2).sum(1, keepdim=True).sqrt().clamp_min(EPS_NORM)
    nv = v.pow(2).sum(1, keepdim=True).sqrt().clamp_min(EPS_NORM)
    xh, vh = x / nx, v / nv
    S = torch.bmm(xh.transpose(1, 2), xh)
    B = torch.clamp(S, min=clamp)
    if p < 1.0:
        B = B + 1e-6
    A = B ** p
    Y = torch.bmm(vh, A.transpose(1, 2))
    ny = Y.pow(2).sum(1, keepdim=True).sqrt().clamp_min(EPS_NORM)
    yh = Y / ny
    cnt = M.sum(-1).sum(-1)
    den = cnt + 1e-6
    G = torch.bmm(k, M.transpose(1, 2))
    loss_n = (yh * G).sum(1).sum(1) / den
    loss = -2 * loss_n.mean()
    gy = (upstream * -2.0 / N / den).view(N, 1, 1) * G
    dY = (gy - yh * (yh * gy).sum(1, keepdim=True)) / ny
    dvh = torch.bmm(dY, A)
    dA = torch.bmm(dY.transpose(1, 2), vh)
    dS = dA * (p * B ** (p - 1.0)) * (S >= clamp).to(x.dtype)
    dxh = torch.bmm(xh, dS + dS.transpose(1, 2))
    dx = (dxh - xh * (xh * dxh).sum(1, keepdim=True)) / nx
    dv = (dvh - vh * (vh * dvh).sum(1, keepdim=True)) / nv
    return loss, yh, dx, dv


# --------------------------------------------------------------------------------------
# K5  upflow8                             contrast/flow/utils/utils.py:87-89
# --------------------------------------------------------------------------------------
UP8_SMALL_OUTPUT = 128   # ATen: (output_H + output_W) <= 128 selects the other CPU kernel (see upflow8)


def upflow8(flow):
    """8 * F.interpolate(flow, size=(8h,8w), mode='bilinear', align_corners=True).

    ATen's CPU bilinear up-sampling is TWO kernels with different operation orders, chosen by the output size
    (`_use_vectorized_kernel_cond_2d`: output_H + output_W <= 128 -> `cpu_upsample_linear_channels_last`, else
    `upsample_generic_Nd_kernel_impl`); both verified bit for bit against torch 2.10 (x86, FMA build) over h, w in
    2..90 / 2..160 (tests/test_oracle_golden.py::test_aten_samplers_agree*).  Common part: src = dst*((in-1)/(out-1));
    i0 = int(src); l1 = src-i0; l0 = 1-l1.  With a,b = row y0 (x0,x1) and c,d = row y1:
      small outputs (8h+8w <= 128): weights w_yx = l_y*l_x rounded to fp32,
          out = fma(d, w11, fma(c, w10, fma(a, w00, b*w01)))
      larger outputs (every production size, e.g. 720x1280): separable, x first,
          t0 = fma(a, lx0, b*lx1);  t1 = fma(c, lx0, d*lx1);  out = fma(t0, ly0, t1*ly1)
    Round 1 restated only the first form (all its goldens were <= 48x64); the second was found in round 2 by the
    720x1280 test against the ATen kernels and is pinned by the goldens upflow8_big / applyflow_*big."""
    M, C, h, w = flow.shape
    Ho, Wo = 8 * h, 8 * w
    if _ATEN_SAMPLERS:
        return 8 * torch.nn.functional.interpolate(flow, size=(Ho, Wo), mode="bilinear", align_corners=True)

    def axis(n_in, n_out):
        scale = torch.tensor((n_in - 1) / (n_out - 1) if n_out > 1 else 0.0, dtype=torch.float32)
        src = torch.arange(n_out, dtype=torch.float32) * scale
        i0 = src.long().clamp_max(n_in - 1)
        i1 = (i0 + 1).clamp_max(n_in - 1)
        l1 = (src - i0.float()).clamp(0, 1)
        return i0, i1, 1 - l1, l1

    y0, y1, ly0, ly1 = axis(h, Ho)
    x0, x1, lx0, lx1 = axis(w, Wo)
    a = flow[:, :, y0][:, :, :, x0]
    b = flow[:, :, y0][:, :, :, x1]
    c = flow[:, :, y1][:, :, :, x0]
    d = flow[:, :, y1][:, :, :, x1]
    ly0, ly1 = ly0.view(1, 1, -1, 1), ly1.view(1, 1, -1, 1)
    lx0, lx1 = lx0.view(1, 1, 1, -1), lx1.view(1, 1, 1, -1)
    if Ho + Wo <= UP8_SMALL_OUTPUT:
        acc = b * (ly0 * lx1)
        acc = _fma(a, ly0 * lx0, acc)
        acc = _fma(c, ly1 * lx0, acc)
        acc = _fma(d, ly1 * lx1, acc)
    else:
        t0 = _fma(a, lx0, b * lx1)
        t1 = _fma(c, lx0, d * lx1)
        acc = _fma(t0, ly0, t1 * ly1)
    return 8 * acc


# --------------------------------------------------------------------------------------
# K6/K7  flow chain and forward-backward consistency      contrast/util.py:252-357
# --------------------------------------------------------------------------------------
def _pixel_grid(nb, ht, wd):
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    return torch.stack([xs, ys], 0).float().repeat(nb, 1, 1, 1)                    # util.py:270-271


def normalize_coord(c):
    """util.py:333-339"""
    ht, wd = c.shape[-2:]
    o = c.clone()
    o[:, 0] = 2 * o[:, 0] / (wd - 1) - 1
    o[:, 1] = 2 * o[:, 1] / (ht - 1) - 1
    return o


def normalize_flow(f):
    """util.py:342-348"""
    ht, wd = f.shape[-2:]
    o = f.clone()
    o[:, 0] = 2 * o[:, 0] / (wd - 1)
    o[:, 1] = 2 * o[:, 1] / (ht - 1)
    return o


def denormalize_flow(f):
    """util.py:351-357"""
    ht, wd = f.shape[-2:]
    o = f.clone()
    o[:, 0] = (o[:, 0] * (wd - 1)) / 2
    o[:, 1] = (o[:, 1] * (ht - 1)) / 2
    return o


def concat_flow(flows, is_norm=False):
    """flows [F-1,N,2,Hf,Wf] -> composed flow [N,2,Hf,Wf].  util.py:300-330."""
    num, nb, _, ht, wd = flows.shape
    if num == 1:                                                                   # :303-308
        out = flows[0].clone()
        return normalize_flow(out) if is_norm else out
    c0 = _pixel_grid(nb, ht, wd)
    c0n = normalize_coord(c0)
    c1, c1n = c0.clone(), c0n.clone()
    for f in flows:                                                                # :315-323
        if is_norm:
            c1n = c1n + bilinear_sample(normalize_flow(f), c1n[:, 0], c1n[:, 1])
        else:
            g = normalize_coord(c1)
            c1 = c1 + bilinear_sample(f, g[:, 0], g[:, 1])
    return (c1n - c0n) if is_norm else (c1 - c0)                                    # :325-328


def forward_backward_consistency(flow_fwd, flow_bwd, alpha_1=0.01, alpha_2=0.5, is_norm=False):
    """Validity mask [N,Hf,Wf] (True = consistent and inside the frame) and the cycle flow.
    util.py:252-297."""
    if is_norm:                                                                    # :259-266
        fn, bn = flow_fwd.clone(), flow_bwd.clone()
    else:
        fn, bn = normalize_flow(flow_fwd), normalize_flow(flow_bwd)
    nb, _, ht, wd = flow_fwd.shape
    c0n = normalize_coord(_pixel_grid(nb, ht, wd))
    c1n = c0n + fn                                                                 # :276
    inside = (torch.abs(c1n[:, 0]) < 1) & (torch.abs(c1n[:, 1]) < 1)               # :277
    bi = bilinear_sample(bn, c1n[:, 0], c1n[:, 1])                                 # :279
    cyc = fn + bi                                                                  # :280
    diag = torch.sqrt(torch.tensor(ht) ** 2 + torch.tensor(wd) ** 2).item()        # :290-292
    a2 = alpha_2 / diag
    cyc_sq = (cyc ** 2).sum(1)                                                     # :294
    eps = alpha_1 * ((fn ** 2).sum(1) + (bi ** 2).sum(1)) + a2                     # :295
    return inside & ((cyc_sq - eps) <= 0), cyc                                     # :297


def calc_mask_ratio(mask):
    """Invalid-pixel ratio per sample.  util.py:360-366."""
    if mask is None:
        return None
    return torch.logical_not(mask).float().mean(-1).mean(-1)


def apply_optical_flow_file(flow_fwds, flow_bwds, size, flow_up=True, alpha1=0.01, alpha2=0.5,
                            flow_cat_norm=False):
    """File-mode branch of util.apply_optical_flow with use_flow_frames=False
    (util.py:174-248).  flow_fwds/bwds [N,F-1,2,h,w] as the loader yields them.
    Returns [flow_fwd, size, mask_fwd], [flow_bwd, size, mask_bwd]."""
    ff = flow_fwds.permute(1, 0, 2, 3, 4)                                          # :183-184
    fb = flow_bwds.permute(1, 0, 2, 3, 4)
    if flow_up:                                                                    # :185-191
        num, nb, c, h, w = ff.shape
        ff = upflow8(ff.reshape(-1, c, h, w)).reshape(num, nb, c, 8 * h, 8 * w)
        fb = upflow8(fb.reshape(-1, c, h, w)).reshape(num, nb, c, 8 * h, 8 * w)
    f = concat_flow(ff, flow_cat_norm)                                             # :192-194 -> :105-109
    b = concat_flow(fb, flow_cat_norm)
    mf = mb = None
    if alpha1 is not None and alpha2 is not None:                                  # :206-222
        mf, _ = forward_backward_consistency(f, b, alpha1, alpha2, flow_cat_norm)
        mb, _ = forward_backward_consistency(b, f, alpha1, alpha2, flow_cat_norm)
    if flow_cat_norm:                                                              # :229-231
        f, b = denormalize_flow(f), denormalize_flow(b)
    return [f, size, mf], [b, size, mb]


# --------------------------------------------------------------------------------------
# EMA of the key encoder                 contrast/models/PixPro.py:321-337
# --------------------------------------------------------------------------------------
def momentum_coefficient(base_m, k, K):
    """1 - (1-m)*(cos(pi*k/K)+1)/2, evaluated in double as the reference does (:326)."""
    return 1. - (1. - base_m) * (math.cos(math.pi * k / K) + 1) / 2.


def momentum_update(params_q, params_k, m):
    """param_k = param_k*m + param_q*(1-m)   (:329-333)"""
    return [pk * m + pq * (1. - m) for pq, pk in zip(params_q, params_k)]


# --------------------------------------------------------------------------------------
# LARS + SGD momentum step               contrast/lars.py:109-151 + torch.optim.SGD
# --------------------------------------------------------------------------------------
def lars_sgd_step(p, g, buf, lr, momentum, weight_decay, adapt, trust_coef=0.001, eps=1e-8):
    """One parameter tensor.  g += wd*p; if adapt: g *= trust*||p||/(||g||+eps) when both
    norms > 0; buf = momentum*buf + g (buf=g on the first step, buf None); p -= lr*buf.
    Returns new (p, buf)."""
    if weight_decay > 0:
        g = g.add(p, alpha=weight_decay)
    if adapt:
        pn, gn = p.norm(), g.norm()
        if pn > 0 and gn > 0:
            g = g * (trust_coef * pn / (gn + eps))
    buf = g.clone() if buf is None else buf * momentum + g
    return p - lr * buf, buf
