"""torch.autograd bindings of the HIP hot path.

`ppm_loss`        fused featprop + normalize + regression_loss (both loss directions, fwd+bwd,
                  one launch)                      <- PixPro.forward, PixPro.py:376-385,429-431
`featprop`        PixPro.featprop alone           <- PixPro.py:339-363
`regression_loss` module-level loss alone         <- PixPro.py:92-247
`add_optical_flow`                                 <- PixPro.py:46-89

Everything here needs CUDA tensors and the built shared library; nothing falls back to torch ops.
"""
import ctypes

import torch

from . import _native as nv


def _size_fields(size, flow, device):
    """(Ho, Wo, size_dev_tensor).  size: None | (Ho, Wo) | Tensor[2].  A device tensor is handed to the
    kernel as-is (no .item() sync, unlike PixPro.py:116); a host tensor / tuple is read here."""
    if size is None:
        if flow is not None:
            return int(flow.shape[-2]), int(flow.shape[-1]), None
        return 0, 0, None
    if isinstance(size, torch.Tensor):
        if size.is_cuda:
            return 0, 0, size.reshape(-1)[:2].to(torch.int32).contiguous()
        return int(size[0].item()), int(size[1].item()), None
    return int(size[0]), int(size[1]), None


class Direction:
    """Operands of one loss direction (q-side features x/v, k-side momentum projection k)."""

    def __init__(self, x, v, k, coord_q, coord_k, flow=None, mask=None, size=None):
        self.x, self.v, self.k = x, v, k
        self.coord_q, self.coord_k = coord_q, coord_k
        self.flow, self.mask, self.size = flow, mask, size       # flow may be a util.LazyFlow (then mask is ignored)


def _problem(dirs, p, clamp, pos_ratio, normalize_k, grad_scale=1.0, allow_nhwc=False):
    d0 = dirs[0]
    N, C, H, W = d0.x.shape
    pb = nv.Problem()
    pb.N, pb.C, pb.H, pb.W = N, C, H, W
    # channels_last activations are consumed in place by the single-launch kernels and by the tiled path (no NCHW copies)
    nhwc = False
    if allow_nhwc and nv.is_nhwc(d0.x):
        pb.layout, pb.ndir = 1, len(dirs)
        nhwc = bool(nv.load().pixpro_ppm_loss_layout_supported(ctypes.byref(pb)))
    pb.layout = int(nhwc)
    pb.p, pb.clamp, pb.pos_ratio = float(p), float(clamp), float(pos_ratio)
    pb.normalize_k, pb.ndir, pb.grad_scale = int(bool(normalize_k)), len(dirs), float(grad_scale)
    keep = []
    cds = (nv.Dir * 2)()
    for i, d in enumerate(dirs):
        shape = (N, C, H, W)
        x = nv.dev_f32(d.x, "x", shape, nhwc)
        v = nv.dev_f32(d.v, "v", shape, nhwc) if d.v is not None else None
        k = nv.dev_f32(d.k, "k", shape, nhwc)
        cq = nv.dev_f32(d.coord_q, "coord_q", (N, 10))
        ck = nv.dev_f32(d.coord_k, "coord_k", (N, 10))
        flow = mask = lo_own = lo_opp = None
        if d.flow is not None and hasattr(d.flow, "lo_own"):          # util.LazyFlow: evaluated inside the kernel
            lz = d.flow
            lo_own = nv.dev_f32(lz.lo_own, "lo_own")
            lo_opp = nv.dev_f32(lz.lo_opp, "lo_opp", lo_own.shape)
            if lo_own.dim() != 5 or lo_own.shape[0] != N or lo_own.shape[1] < 1 or lo_own.shape[2] != 2:
                raise RuntimeError(f"lazy flow: expected low-res maps [N,F-1,2,h,w] with N={N}, got {tuple(lo_own.shape)}")
            lf, lh, lw = int(lo_own.shape[1]), int(lo_own.shape[-2]), int(lo_own.shape[-1])
            if i and (lf, lh, lw) != (pb.lo_frames, pb.lo_h, pb.lo_w):
                raise RuntimeError("both directions must use the same low-res flow stack shape")
            pb.lo_frames, pb.lo_h, pb.lo_w = lf, lh, lw
            pb.lazy_mask = int(lz.with_mask)
            pb.alpha1 = float(lz.alpha1) if lz.with_mask else 0.0
            pb.alpha2 = float(lz.alpha2) if lz.with_mask else 0.0
        elif d.flow is not None:
            if d.flow.dim() != 4 or d.flow.shape[0] != N or d.flow.shape[1] != 2:
                raise RuntimeError(f"flow: expected [N,2,Hf,Wf] with N={N}, got {tuple(d.flow.shape)}")
            flow = nv.dev_f32(d.flow, "flow")
            Hf, Wf = int(flow.shape[-2]), int(flow.shape[-1])
            if i and (Hf, Wf) != (pb.Hf, pb.Wf):
                raise RuntimeError("both directions must use the same flow map size")
            pb.Hf, pb.Wf = Hf, Wf
            if d.mask is not None:
                mask = nv.dev_mask(d.mask, "mask", (N, Hf, Wf))
        elif d.mask is not None:
            raise RuntimeError("mask given without flow")
        if lo_own is not None and d.size is None:
            raise RuntimeError("lazy flow needs the frame size")
        Ho, Wo, size_dev = _size_fields(d.size, flow, x.device)
        if i and (Ho, Wo) != (pb.Ho, pb.Wo):
            raise RuntimeError("both directions must use the same frame size")
        pb.Ho, pb.Wo = Ho, Wo
        keep += [x, v, k, cq, ck, flow, mask, size_dev, lo_own, lo_opp]
        c = cds[i]
        c.x, c.v, c.k = nv.ptr(x), nv.ptr(v), nv.ptr(k)
        c.coord_q, c.coord_k = nv.ptr(cq), nv.ptr(ck)
        c.flow, c.mask, c.size_dev = nv.ptr(flow), nv.ptr(mask), nv.ptr(size_dev)
        c.lo_own, c.lo_opp = nv.ptr(lo_own), nv.ptr(lo_opp)
    return pb, cds, keep


class _PPMLoss(torch.autograd.Function):
    """inputs: x_0, v_0, [x_1, v_1]; everything else (k, coords, flows, masks) is constant."""

    @staticmethod
    def forward(ctx, meta, *xv):
        dirs, p, clamp, pos_ratio, normalize_k, want_pred, generic = meta
        nd = len(dirs)
        for i, d in enumerate(dirs):
            d.x, d.v = xv[2 * i], xv[2 * i + 1]
        pb, cds, keep = _problem(dirs, p, clamp, pos_ratio, normalize_k, allow_nhwc=not generic)
        x0 = keep[0]
        dev = x0.device
        N = pb.N
        want_grad = any(ctx.needs_input_grad[1:])
        out = torch.empty(3 * nd * N + 1, dtype=torch.float32, device=dev)
        loss_n, pos_num, pos_mean = out[:nd * N], out[nd * N:2 * nd * N], out[2 * nd * N:3 * nd * N]
        loss = out[3 * nd * N:]
        def like_x(lead):
            n_, c_, h_, w_ = x0.shape
            if pb.layout:   # [lead, N, H, W, C] storage viewed as [lead, N, C, H, W]: every slice is channels_last
                return torch.empty((lead, n_, h_, w_, c_), dtype=torch.float32, device=dev).permute(0, 1, 4, 2, 3)
            return torch.empty((lead, n_, c_, h_, w_), dtype=torch.float32, device=dev)
        grads = like_x(2 * nd) if want_grad else None
        preds = like_x(nd) if want_pred else None
        for i in range(nd):
            if want_grad:
                cds[i].dx, cds[i].dv = nv.ptr(grads[2 * i]), nv.ptr(grads[2 * i + 1])
            if want_pred:
                cds[i].pred = nv.ptr(preds[i])
        lib = nv.load()
        query = lib.pixpro_ppm_loss_generic_workspace_bytes if generic else lib.pixpro_ppm_loss_workspace_bytes
        ws_bytes = query(ctypes.byref(pb))
        ws = nv.workspace(ws_bytes, dev) if ws_bytes else None   # 0: the single-launch MFMA kernel applies
        nv.call("pixpro_ppm_loss_generic" if generic else "pixpro_ppm_loss", ctypes.byref(pb), cds,
                nv.ptr(loss_n), nv.ptr(pos_num), nv.ptr(pos_mean), nv.ptr(loss), nv.ptr(ws),
                0 if ws is None else ws.numel(), nv.stream())
        ctx.grads = grads
        ctx.nd = nd
        loss_s = loss.reshape(())
        outs = [loss_s, pos_num.reshape(nd, N), pos_mean.reshape(nd, N), loss_n.reshape(nd, N)]
        ctx.mark_non_differentiable(*outs[1:])
        if want_pred:
            ctx.mark_non_differentiable(preds)
            outs.append(preds)
        return tuple(outs)

    @staticmethod
    def backward(ctx, gloss, *_):
        grads = ctx.grads
        if grads is None:
            return (None,) * (1 + 2 * ctx.nd)
        # The kernel wrote d(loss)/d(x,v) for an upstream of 1.  First call: bring the buffer to the upstream scalar in place,
        # on the device (the kernel returns at once when it is 1, the usual case) - no host sync - and hand out views of it.
        # The views stay referenced from ctx: AccumulateGrad keeps ("steals") a gradient tensor nobody else holds, and a
        # leaf's .grad that aliased this buffer would be accumulated into IN PLACE by a second backward (retain_graph) -
        # corrupting what that call rescales.  With the reference held, a leaf gets its own copy; the usual consumer (the
        # projector's conv backward) only reads.  Later calls never write the buffer: their result is a fresh tensor.
        g = gloss.detach().to(torch.float32).reshape(1)
        applied = getattr(ctx, "applied", None)
        if applied is None:
            nv.call("pixpro_scale_by_device_scalar", nv.ptr(grads), grads.numel(), nv.ptr(g.contiguous()), nv.stream())  # dense buffer
            ctx.applied = g.clone()
            ctx.handed = tuple(grads[i] for i in range(2 * ctx.nd))
            return (None,) + ctx.handed
        out = grads * (g / applied)          # same strides (dense), channels_last slices stay channels_last
        return (None,) + tuple(out[i] for i in range(2 * ctx.nd))


def ppm_loss(dirs, p, clamp, pos_ratio, normalize_k=True, want_pred=False, generic=False):
    """dirs: list of 1 or 2 `Direction`.  Returns (loss, pos_num[nd,N], pos_mean[nd,N], loss_n[nd,N]
    [, preds[nd,N,C,H,W]]).  loss is differentiable w.r.t. every x and v."""
    xv = []
    for d in dirs:
        xv += [d.x, d.v]
    return _PPMLoss.apply((dirs, p, clamp, pos_ratio, normalize_k, want_pred, generic), *xv)


class _FeatProp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, v, p, clamp):
        N, C, H, W = x.shape
        x = nv.dev_f32(x, "feat")
        v = nv.dev_f32(v, "value", x.shape)
        pb = nv.Problem()
        pb.N, pb.C, pb.H, pb.W, pb.p, pb.clamp, pb.ndir = N, C, H, W, float(p), float(clamp), 1
        Y = torch.empty_like(x)
        lib = nv.load()
        ws = nv.workspace(lib.pixpro_featprop_workspace_bytes(ctypes.byref(pb)), x.device)
        nv.call("pixpro_featprop_fwd", ctypes.byref(pb), nv.ptr(x), nv.ptr(v), nv.ptr(Y), nv.ptr(ws), ws.numel(),
                nv.stream())
        ctx.save_for_backward(x, v)
        ctx.pc = (float(p), float(clamp))
        return Y

    @staticmethod
    def backward(ctx, dY):
        x, v = ctx.saved_tensors
        N, C, H, W = x.shape
        pb = nv.Problem()
        pb.N, pb.C, pb.H, pb.W, pb.p, pb.clamp, pb.ndir = N, C, H, W, ctx.pc[0], ctx.pc[1], 1
        dY = nv.dev_f32(dY, "dY", x.shape)
        dx, dv = torch.empty_like(x), torch.empty_like(x)
        lib = nv.load()
        ws = nv.workspace(lib.pixpro_featprop_workspace_bytes(ctypes.byref(pb)), x.device)
        nv.call("pixpro_featprop_bwd", ctypes.byref(pb), nv.ptr(x), nv.ptr(v), nv.ptr(dY), nv.ptr(dx), nv.ptr(dv),
                nv.ptr(ws), ws.numel(), nv.stream())
        return dx, dv, None, None


def featprop(x, v, p, clamp):
    """Y = normalize(v) @ (clamp(normalize(x)^T normalize(x)) ** p)^T    (PixPro.py:339-363)."""
    return _FeatProp.apply(x, v, p, clamp)


class _RegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, meta):
        d, pos_ratio = meta
        d.x, d.v = q, None
        pb, cds, keep = _problem([d], 1.0, 0.0, pos_ratio, False)
        dev = keep[0].device
        N = pb.N
        out = torch.empty(3 * N + 1, dtype=torch.float32, device=dev)
        dq = torch.empty_like(keep[0]) if ctx.needs_input_grad[0] else None
        cds[0].dx = nv.ptr(dq)
        lib = nv.load()
        ws = nv.workspace(lib.pixpro_regression_loss_workspace_bytes(ctypes.byref(pb)), dev)
        nv.call("pixpro_regression_loss", ctypes.byref(pb), cds, nv.ptr(out[:N]), nv.ptr(out[N:2 * N]),
                nv.ptr(out[2 * N:3 * N]), nv.ptr(out[3 * N:]), nv.ptr(ws), ws.numel(), nv.stream())
        ctx.dq = dq
        pos_num, pos_mean = out[N:2 * N], out[2 * N:3 * N]
        ctx.mark_non_differentiable(pos_num, pos_mean)
        return out[3 * N:].reshape(()), pos_num, pos_mean

    @staticmethod
    def backward(ctx, gloss, *_):
        if ctx.dq is None:
            return None, None
        return ctx.dq * gloss, None


def regression_loss_op(q, k, coord_q, coord_k, pos_ratio, flow=None, mask=None, size=None):
    d = Direction(q, None, k, coord_q, coord_k, flow, mask, size)
    return _RegLoss.apply(q, (d, pos_ratio))


def add_optical_flow_op(flow, x_grid, y_grid, size, mask=None):
    N, H, W = x_grid.shape
    flow = nv.dev_f32(flow, "flow")
    Hf, Wf = int(flow.shape[-2]), int(flow.shape[-1])
    xg = nv.dev_f32(x_grid, "x_grid")
    yg = nv.dev_f32(y_grid, "y_grid", xg.shape)
    pb = nv.Problem()
    pb.N, pb.C, pb.H, pb.W, pb.Hf, pb.Wf, pb.Ho, pb.Wo, pb.ndir = N, 1, H, W, Hf, Wf, int(size[0]), int(size[1]), 1
    ox, oy = torch.empty_like(xg), torch.empty_like(yg)
    mg = None
    if mask is not None:
        mask = nv.dev_mask(mask, "mask", (N, Hf, Wf))
        mg = torch.empty((N, 1, H, W), dtype=torch.bool, device=xg.device)
    nv.call("pixpro_add_optical_flow", ctypes.byref(pb), nv.ptr(flow), nv.ptr(xg), nv.ptr(yg), nv.ptr(mask),
            nv.ptr(ox), nv.ptr(oy), nv.ptr(mg), nv.stream())
    return ox, oy, mg
