"""`contrast.util` surface: flow preparation on the HIP library + the small training helpers.

Signatures follow the reference's contrast/util.py (apply_optical_flow :175, all_concat_flow :105,
forward_backward_consistency :253, concat_flow :301, normalize_coord :334, normalize_flow :343,
denormalize_flow :352, calc_mask_ratio :361, AverageMeter :10); `upflow8` is the
reference's contrast/flow/utils/utils.py:87.  `calc_frame_ratio` is called by the reference's
main_pretrain.py:230 but defined nowhere (defect D1); it is defined here with the contract its caller
needs.  The online branch (a RAFT flow model instead of flow files, util.py:76-171,200-203) runs contrast/flow/raft.py
per frame pair and then the same fused preparation as the file branch.
"""
import argparse
import os

import torch

from . import _native as nv


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class MyHelpFormatter(argparse.MetavarTypeHelpFormatter, argparse.ArgumentDefaultsHelpFormatter):
    pass


# ------------------------------------------------------------------------------------------
# element-wise helpers of the reference API (tiny; stock torch, device-agnostic)
# ------------------------------------------------------------------------------------------
@torch.no_grad()
def normalize_coord(coords):
    _, _, ht, wd = coords.shape
    out = coords.clone()
    out[:, 0] = 2 * out[:, 0] / (wd - 1) - 1
    out[:, 1] = 2 * out[:, 1] / (ht - 1) - 1
    return out


@torch.no_grad()
def normalize_flow(flow):
    _, _, ht, wd = flow.shape
    out = flow.clone()
    out[:, 0] = 2 * out[:, 0] / (wd - 1)
    out[:, 1] = 2 * out[:, 1] / (ht - 1)
    return out


@torch.no_grad()
def denormalize_flow(flow_norm):
    _, _, ht, wd = flow_norm.shape
    out = flow_norm.clone()
    out[:, 0] = (out[:, 0] * (wd - 1)) / 2
    out[:, 1] = (out[:, 1] * (ht - 1)) / 2
    return out


# ------------------------------------------------------------------------------------------
# HIP kernels
# ------------------------------------------------------------------------------------------
@torch.no_grad()
def upflow8(flow, mode='bilinear'):
    """[M,2,h,w] -> [M,2,8h,8w] = 8 * bilinear(align_corners=True)."""
    if mode != 'bilinear':
        raise NotImplementedError("upflow8: only bilinear is used by the pre-training path")
    flow = nv.dev_f32(flow, "flow")
    M, C, h, w = flow.shape
    out = torch.empty((M, C, 8 * h, 8 * w), dtype=torch.float32, device=flow.device)
    nv.call("pixpro_upflow8", nv.ptr(flow), nv.ptr(out), M * C, h, w, nv.stream())
    return out


@torch.no_grad()
def concat_flow(flows, is_norm=False):
    """[F-1,N,2,Hf,Wf] -> composed flow [N,2,Hf,Wf] (one launch; the reference runs F-1 grid_samples)."""
    flows = nv.dev_f32(flows, "flows")
    num, nb, two, ht, wd = flows.shape
    if two != 2:
        raise RuntimeError(f"flows: expected [F-1,N,2,H,W], got {tuple(flows.shape)}")
    out = torch.empty((nb, 2, ht, wd), dtype=torch.float32, device=flows.device)
    nv.call("pixpro_concat_flow", nv.ptr(flows), nv.ptr(out), num, nb, ht, wd, int(bool(is_norm)), nv.stream())
    return out


def all_concat_flow(flow_fwds, flow_bwds, is_norm=False, use_flow_frames=True):
    if use_flow_frames:
        raise NotImplementedError("use_flow_frames builds a 5-D flow the loss cannot consume in the "
                                  "reference either (defect D5); only use_flow_frames=False is supported")
    return concat_flow(flow_fwds, is_norm), concat_flow(flow_bwds, is_norm)


@torch.no_grad()
def forward_backward_consistency(flow_fwd, flow_bwd, coords0=None, alpha_1=0.01, alpha_2=0.5, is_norm=False):
    """Returns (coords0_norm, coords1_norm, [mask, flow_cycle]) like the reference; the two coordinate
    grids are debug outputs there and are returned as None here."""
    if alpha_1 is None or alpha_2 is None:
        return flow_fwd.clone(), flow_bwd.clone(), [None, None]
    f = nv.dev_f32(flow_fwd.detach(), "flow_fwd")
    b = nv.dev_f32(flow_bwd.detach(), "flow_bwd", f.shape)
    nb, _, ht, wd = f.shape
    mask = torch.empty((nb, ht, wd), dtype=torch.bool, device=f.device)
    cycle = torch.empty_like(f)
    nv.call("pixpro_fbc_mask", nv.ptr(f), nv.ptr(b), nv.ptr(mask), nv.ptr(cycle), nb, ht, wd, float(alpha_1),
            float(alpha_2), int(bool(is_norm)), nv.stream())
    return None, None, [mask, cycle]


@torch.no_grad()
def calc_mask_ratio(mask):
    if mask is None:
        return None
    if isinstance(mask, LazyMask):          # logging only: materialise with the dense kernel
        mask = mask.dense()
    if isinstance(mask, MaskWithCounts):
        return mask.invalid_ratio()
    m = nv.dev_mask(mask, "mask", mask.shape)
    lead = m.shape[:-2]
    n = 1
    for s in lead:
        n *= s
    out = torch.empty(n, dtype=torch.float32, device=m.device)
    nv.call("pixpro_mask_ratio", nv.ptr(m), nv.ptr(out), n, m.shape[-2] * m.shape[-1], nv.stream())
    return out.reshape(lead)


class LazyFlow:
    """Composed flow of one direction that has NOT been materialised (--flow_up, --lazy-flow): it carries the
    loader's 1/8-resolution stacks [N,F-1,2,h,w] of this and of the opposite direction plus alpha1 / alpha2, and the loss
    kernel evaluates flow and consistency mask at its P sample points only.  `.dense()` materialises
    (flow[N,2,Hf,Wf], mask[N,Hf,Wf] or None) with the dense kernel when something else needs the tensors."""

    def __init__(self, lo_own, lo_opp, alpha1, alpha2):
        self.lo_own, self.lo_opp = lo_own, lo_opp
        self.alpha1, self.alpha2 = alpha1, alpha2
        self._dense = None

    @property
    def with_mask(self):
        return self.alpha1 is not None and self.alpha2 is not None

    @property
    def map_size(self):
        return 8 * int(self.lo_own.shape[-2]), 8 * int(self.lo_own.shape[-1])

    def dense(self):
        if self._dense is None:
            args = argparse.Namespace(use_flow_file=True, use_flow_frames=False, flow_cat_norm=False, flow_up=True,
                                      alpha1=self.alpha1, alpha2=self.alpha2, lazy_flow=False, debug=False)
            data = [None] * 5 + [[None, self.lo_own, self.lo_opp], [[None], [torch.tensor(self.lo_own.shape[1] + 1)]]]
            fwd, _ = apply_optical_flow(data, None, args)
            self._dense = (fwd[0], fwd[2])
        return self._dense


class LazyMask:
    """Place-holder for the consistency mask of a LazyFlow (third element of the [flow, size, mask] triple)."""

    def __init__(self, flow):
        self.flow = flow

    def dense(self):
        return self.flow.dense()[1]


class MaskWithCounts(torch.Tensor):
    """Bool mask [N,Hf,Wf] that remembers the per-sample invalid-pixel counts produced by the fused
    flow-prepare launch, so that calc_mask_ratio (logging only) does not re-read the mask from HBM."""

    @staticmethod
    def wrap(mask, counts):
        out = mask.as_subclass(MaskWithCounts)
        out._counts = counts
        return out

    def invalid_ratio(self):
        counts = getattr(self, "_counts", None)
        if counts is None:
            return calc_mask_ratio(self.as_subclass(torch.Tensor))
        return counts.to(torch.float32) / float(self.shape[-2] * self.shape[-1])


def calc_frame_ratio(cur_n_frames):
    """Defined for the reference's call site main_pretrain.py:230-231,305,315-337: returns
    (mean_n_frames, no_optical_flow_ratio, num_per_frame[n,2]) where num_per_frame[i] =
    (share of samples that used i+1 frames, their count)."""
    n = cur_n_frames.detach().reshape(-1).to(torch.float32).cpu()
    total = max(n.numel(), 1)
    mean_n_frames = float(n.mean()) if n.numel() else 1.0
    no_of_r = float((n <= 1).float().mean()) if n.numel() else 1.0
    top = int(n.max()) if n.numel() else 1
    rows = []
    for i in range(1, top + 1):
        c = float((n == i).sum())
        rows.append([c / total, c])
    return mean_n_frames, no_of_r, torch.tensor(rows)


def load_flow_model(args, device="cuda"):
    """The flow-model set-up of the reference's main_pretrain.py:44-58: RAFT from `args.flow_model` (`small` in the file
    name selects RAFT-small, option.py:107-110), DataParallel `module.` prefixes stripped, eval mode, frozen."""
    from .flow import RAFT
    if args.flow_model is None or not os.path.isfile(args.flow_model):
        raise FileNotFoundError(f"not exit flow model path {args.flow_model}")
    if "small" in os.path.basename(args.flow_model):
        args.small = True
    args.mixed_precision = False
    model = RAFT(args)
    weights = torch.load(args.flow_model, map_location="cpu", weights_only=True)
    model.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in weights.items()})
    model = model.to(device).eval()
    for p in model.parameters():
        p.requires_grad = False
    return model


@torch.no_grad()
def calc_optical_flow(imgs, flow_model, up=False, verbose=False):
    """util.py:76-102: forward flows of consecutive frame pairs and backward flows of the time-reversed clip,
    [F-1, N, 2, h', w'] each (1/8 resolution, or full resolution when `up`)."""
    assert len(imgs) >= 2
    i = 1 if up else 0
    flow_model.eval()
    fwd = torch.stack([flow_model(a, b, upsample=False, test_mode=True)[i] for a, b in zip(imgs[:-1], imgs[1:])])
    bwd = torch.stack([flow_model(a, b, upsample=False, test_mode=True)[i] for a, b in zip(imgs[1:][::-1], imgs[:-1][::-1])])
    return fwd, bwd


def _online_flows(imgs, flow_model, args):
    """Flows of every frame pair in loader layout [N,F-1,2,h',w'], computed `flow_bs` samples at a time (the all-pairs
    volume of one 720x1280 pair is 1.1 GB; util.py:129-171).  RAFT-small up-samples with plain x8 bilinear interpolation
    (update_block returns no mask), so with --flow_up its 1/8-resolution output is handed on and the fused preparation
    up-samples lazily; the full model's convex up-sampling is materialised.  Returns (lo_f, lo_b, still_to_upsample)."""
    lazy_up = bool(args.flow_up) and getattr(flow_model, "update_block").mask is None
    want_up = bool(args.flow_up) and not lazy_up
    bs = imgs[0].shape[0]
    step = getattr(args, "flow_bs", None) or 8
    fs, bs_ = [], []
    for s in range(0, bs, step):
        part = [im[s:s + step] for im in imgs]
        f, b = calc_optical_flow(part, flow_model, up=want_up, verbose=getattr(args, "verbose", False))
        fs.append(f.permute(1, 0, 2, 3, 4))
        bs_.append(b.permute(1, 0, 2, 3, 4))
    return torch.cat(fs).contiguous(), torch.cat(bs_).contiguous(), lazy_up


@torch.no_grad()
def mem_reduce_calc_optical_flow(orig_imgs, flow_model, args):
    """util.py:129-171 (use_flow_frames=False): RAFT over every frame pair `flow_bs` samples at a time, then the flow
    chain; returns the composed flows as [1, N, 2, Hf, Wf] each, like the reference."""
    if getattr(args, "use_flow_frames", False) and len(orig_imgs) > 2:
        raise NotImplementedError("use_flow_frames is half-implemented in the reference (defect D5)")
    lo_f, lo_b, up = _online_flows(list(orig_imgs), flow_model, args)
    N, F1, _, h, w = lo_f.shape
    Hf, Wf = (8 * h, 8 * w) if up else (h, w)
    flow_fwd = torch.empty((N, 2, Hf, Wf), dtype=torch.float32, device=lo_f.device)
    flow_bwd = torch.empty_like(flow_fwd)
    nv.call("pixpro_flow_prepare", nv.ptr(lo_f), nv.ptr(lo_b), nv.ptr(flow_fwd), nv.ptr(flow_bwd), None, None, None,
            N, F1, h, w, int(up), 0.0, 0.0, nv.stream())
    if getattr(args, "flow_cat_norm", False):
        flow_fwd, flow_bwd = normalize_flow(flow_fwd), normalize_flow(flow_bwd)
    return flow_fwd.unsqueeze(0), flow_bwd.unsqueeze(0)


_FLOW_WS = {}


def _flow_workspace(dev, nbytes):
    """One cached scratch buffer per device (stream-ordered reuse: every call enqueues on the current stream)."""
    if nbytes <= 0:
        return None
    buf = _FLOW_WS.get(dev)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _FLOW_WS[dev] = buf
    return buf


@torch.no_grad()
def apply_optical_flow(data, flow_model, args):
    """The reference's apply_optical_flow (util.py:174-248), fused: permute + upflow8 + concat_flow (both directions) +
    the two consistency masks are ONE launch for nframe=2 and three launches for nframe>2, and the up-sampled flows are
    never materialised.  File branch: data[5] = [target, flow_fwds[N,F-1,2,h,w], flow_bwds[N,F-1,2,h,w]]; online branch:
    data[6] = [size[N,2], num_img[N,1], frame_0[N,3,H,W], ...] and `flow_model` = RAFT.
    Returns [flow_fwd[N,2,Hf,Wf], size, mask_fwd[N,Hf,Wf]], [flow_bwd, size, mask_bwd]."""
    if getattr(args, "use_flow_frames", False) and data[6][1][0].item() > 2:
        raise NotImplementedError("use_flow_frames is half-implemented in the reference (defect D5)")
    size = data[6][0][0]
    up = bool(args.flow_up)
    if getattr(args, "use_flow_file", True):
        _, lo_f, lo_b = data[5]
    else:
        if flow_model is None:
            raise RuntimeError("online flow mode needs a flow model (util.load_flow_model) or --use_flow_file")
        lo_f, lo_b, up = _online_flows(list(data[6][2:]), flow_model, args)
    if getattr(args, "flow_cat_norm", False):
        return _apply_optical_flow_staged(lo_f, lo_b, size, up, args)
    lo_f = nv.dev_f32(lo_f, "flow_fwds")
    lo_b = nv.dev_f32(lo_b, "flow_bwds", lo_f.shape)
    N, F1, two, h, w = lo_f.shape
    if getattr(args, "lazy_flow", False) and up and not getattr(args, "debug", False):
        # nothing is launched here: the loss kernel evaluates flow + mask at its sample points (ops.ppm_loss)
        fwd = LazyFlow(lo_f, lo_b, args.alpha1, args.alpha2)
        bwd = LazyFlow(lo_b, lo_f, args.alpha1, args.alpha2)
        return ([fwd, size, LazyMask(fwd) if fwd.with_mask else None],
                [bwd, size, LazyMask(bwd) if bwd.with_mask else None])
    Hf, Wf = (8 * h, 8 * w) if up else (h, w)
    dev = lo_f.device
    flow_fwd = torch.empty((N, 2, Hf, Wf), dtype=torch.float32, device=dev)
    flow_bwd = torch.empty_like(flow_fwd)
    is_mask = args.alpha1 is not None and args.alpha2 is not None
    mask_fwd = mask_bwd = counts = None
    if is_mask:
        mask_fwd = torch.empty((N, Hf, Wf), dtype=torch.bool, device=dev)
        mask_bwd = torch.empty_like(mask_fwd)
        counts = torch.empty(2 * N, dtype=torch.int32, device=dev)
    # scratch of the certified mask route for nframe > 2 (block bounding boxes of the composed flows); 0 bytes otherwise
    ws_bytes = nv.load().pixpro_flow_prepare_workspace_bytes(N, F1, h, w, int(up)) if is_mask else 0
    ws = _flow_workspace(dev, ws_bytes)
    nv.call("pixpro_flow_prepare_ws", nv.ptr(lo_f), nv.ptr(lo_b), nv.ptr(flow_fwd), nv.ptr(flow_bwd),
            nv.ptr(mask_fwd), nv.ptr(mask_bwd), nv.ptr(counts), N, F1, h, w, int(up),
            float(args.alpha1) if is_mask else 0.0, float(args.alpha2) if is_mask else 0.0,
            nv.ptr(ws), ws_bytes if ws is not None else 0, nv.stream())
    if is_mask:
        mask_fwd = MaskWithCounts.wrap(mask_fwd, counts[:N])
        mask_bwd = MaskWithCounts.wrap(mask_bwd, counts[N:])
        if getattr(args, "debug", False):
            mask_fwd, mask_bwd = [mask_fwd, None], [mask_bwd, None]
    return [flow_fwd, size, mask_fwd], [flow_bwd, size, mask_bwd]


def _apply_optical_flow_staged(lo_f, lo_b, size, up, args):
    """--flow_cat_norm variant (not used by the production recipes): the reference's stage order with
    one HIP launch per stage."""
    ff = lo_f.permute(1, 0, 2, 3, 4).contiguous()
    fb = lo_b.permute(1, 0, 2, 3, 4).contiguous()
    if up:
        num, nb, c, h, w = ff.shape
        ff = upflow8(ff.reshape(-1, c, h, w)).reshape(num, nb, c, 8 * h, 8 * w)
        fb = upflow8(fb.reshape(-1, c, h, w)).reshape(num, nb, c, 8 * h, 8 * w)
    f, b = concat_flow(ff, True), concat_flow(fb, True)
    mf = mb = None
    if args.alpha1 is not None and args.alpha2 is not None:
        mf = forward_backward_consistency(f, b, alpha_1=args.alpha1, alpha_2=args.alpha2, is_norm=True)[2][0]
        mb = forward_backward_consistency(b, f, alpha_1=args.alpha1, alpha_2=args.alpha2, is_norm=True)[2][0]
    return [denormalize_flow(f), size, mf], [denormalize_flow(b), size, mb]
