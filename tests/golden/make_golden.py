"""Generate golden input/output vectors from the reference's own Python (build container only).

Run:  python tests/golden/make_golden.py     (needs /root/reference; never runs on the GPU box)

The reference is imported read-only as an oracle following SURVEY.md §8(c): empty
``torchvision`` stubs (pulled in only by the debug drawing module), ``Tensor.cuda`` made
a no-op, ``get_world_size`` patched to 1.  Only DATA is written: seeded inputs and the
reference's outputs, as small ``.npz`` files next to this script.
"""
import importlib
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from recipes import crop_coords, smooth_flow, seeded_state_dict, tag_seed  # noqa: E402
REF = "/root/reference"


def load_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    P = importlib.import_module("contrast.models.PixPro")
    util = importlib.import_module("contrast.util")
    resnet = importlib.import_module("contrast.resnet")
    lars = importlib.import_module("contrast.lars")
    torch.Tensor.cuda = lambda self, *a, **k: self
    P.get_world_size = lambda: 1
    return P, util, resnet, lars


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def main():
    P, util, resnet, lars = load_reference()
    F = torch.nn.functional
    torch.set_num_threads(4)

    # ---------------- K1: featprop + normalize --------------------------------------
    for tag, (N, C, H, W, p, clamp, layer) in {
        "p2": (2, 256, 7, 7, 2.0, 0.0, 1),
        "p1": (2, 64, 7, 7, 1.0, 0.0, 0),
        "p05": (2, 64, 7, 7, 0.5, 0.1, 1),
        "g14": (1, 128, 14, 14, 2.0, 0.0, 1),
        "g2": (4, 64, 2, 2, 2.0, 0.0, 1),
    }.items():
        g = torch.Generator().manual_seed(tag_seed(tag))
        x = torch.randn(N, C, H, W, generator=g)
        wv = torch.randn(C, C, generator=g) / C ** 0.5
        bv = torch.randn(C, generator=g) * 0.1
        holder = types.SimpleNamespace(pixpro_p=p, pixpro_clamp_value=clamp)
        if layer == 1:
            holder.value_transform = lambda t: F.conv2d(t, wv.view(C, C, 1, 1), bv)
        else:
            holder.value_transform = lambda t: t
        xr = x.clone().requires_grad_(True)
        Y = P.PixPro.featprop(holder, xr)
        pred = F.normalize(Y, dim=1)
        gy = torch.randn(N, C, H, W, generator=g)
        (pred * gy).sum().backward()
        v = holder.value_transform(x)
        save(f"featprop_{tag}", x=x, v=v, wv=wv, bv=bv, p=p, clamp=clamp, layer=layer,
             Y=Y, pred=pred, gpred=gy, dx_total=xr.grad)

    # ---------------- K2/K4: regression_loss in all call forms ------------------------
    rng = random.Random(7)
    for tag, cfg in {
        "noflow7": dict(N=6, C=64, H=7, W=7, Wo=320, Ho=180, mode="none"),
        "zeroflow7": dict(N=6, C=64, H=7, W=7, Wo=320, Ho=180, mode="zero"),
        "flowfull7": dict(N=6, C=64, H=7, W=7, Wo=64, Ho=48, mode="full", sigma=6.0),
        "flowmask7": dict(N=6, C=64, H=7, W=7, Wo=64, Ho=48, mode="full", sigma=6.0, mask=True),
        "floweighth7": dict(N=6, C=64, H=7, W=7, Wo=512, Ho=384, mode="eighth", sigma=3.0, mask=True),
        "flowbare14": dict(N=2, C=32, H=14, W=14, Wo=64, Ho=48, mode="bare", sigma=4.0),
        "flowmask2": dict(N=8, C=32, H=2, W=2, Wo=64, Ho=48, mode="full", sigma=4.0, mask=True),
        "c256": dict(N=2, C=256, H=7, W=7, Wo=64, Ho=48, mode="full", sigma=5.0, mask=True),
    }.items():
        g = torch.Generator().manual_seed(tag_seed(tag) + 1)
        N, C, H, W, Wo, Ho = (cfg[k] for k in ("N", "C", "H", "W", "Wo", "Ho"))
        q = F.normalize(torch.randn(N, C, H, W, generator=g), dim=1).requires_grad_(True)
        k = F.normalize(torch.randn(N, C, H, W, generator=g), dim=1)
        cq, ck = crop_coords(N, Wo, Ho, rng), crop_coords(N, Wo, Ho, rng)
        # overlap-friendly: make half of the k crops small perturbations of the q crops
        ck[: N // 2, :4] = (cq[: N // 2, :4] + 0.03 * torch.randn(N // 2, 4, generator=g))
        mode = cfg["mode"]
        flow = mask = None
        size = (Ho, Wo)
        if mode == "zero":
            flow = torch.zeros(N, 2, Ho, Wo)
        elif mode in ("full", "bare"):
            flow = smooth_flow((N, 2, Ho, Wo), cfg["sigma"], g)
        elif mode == "eighth":
            flow = smooth_flow((N, 2, Ho // 8, Wo // 8), cfg["sigma"], g)
        if cfg.get("mask"):
            mask = torch.rand(N, *flow.shape[-2:], generator=g) > 0.3
        if mode == "none":
            a, b = cq, ck
        elif mode == "bare":
            a, b = [cq, flow], [ck, flow]
        else:
            a = [cq, [flow, torch.tensor([Ho, Wo]) if mode == "eighth" else size, mask]]
            b = [ck, [flow, size, mask]]
        loss, (pn, pm) = P.regression_loss(q, k, a, b, 0.7)
        loss.backward()
        extra = {}
        if flow is not None:
            # also capture add_optical_flow itself on the bin centres of crop q
            xs = torch.arange(0., float(W)).view(1, 1, -1).repeat(1, H, 1)
            ys = torch.arange(0., float(H)).view(1, -1, 1).repeat(1, 1, W)
            bw = ((cq[:, 2] - cq[:, 0]) / W).view(-1, 1, 1)
            bh = ((cq[:, 3] - cq[:, 1]) / H).view(-1, 1, 1)
            gx = ((xs + 0.5) * bw + cq[:, 0].view(-1, 1, 1)) * (Wo - 1)
            gy = ((ys + 0.5) * bh + cq[:, 1].view(-1, 1, 1)) * (Ho - 1)
            ox, oy, mg = P.add_optical_flow(flow, gx, gy, size, mask)
            extra = dict(aof_x_in=gx, aof_y_in=gy, aof_x=ox, aof_y=oy)
            if mg is not None:
                extra["aof_mask"] = mg
        save(f"regloss_{tag}", q=q, k=k, coord_q=cq, coord_k=ck, mode=mode, size=np.array(size),
             flow=flow if flow is not None else np.zeros(0, np.float32),
             mask=mask if mask is not None else np.zeros(0, bool), has_mask=mask is not None,
             pos_ratio=0.7, loss=loss, pos_num=pn, pos_mean=pm, dq=q.grad, **extra)

    # ---------------- end-to-end hot path (both views, both directions) ---------------
    for tag, (N, C, H, W, p, layer) in {"e2e7": (3, 256, 7, 7, 2.0, 1), "e2e14": (1, 64, 14, 14, 2.0, 1),
                                         "e2e2": (8, 32, 2, 2, 2.0, 1)}.items():
        g = torch.Generator().manual_seed(100 + H)
        Wo, Ho = 64, 48
        x1 = torch.randn(N, C, H, W, generator=g).requires_grad_(True)
        x2 = torch.randn(N, C, H, W, generator=g).requires_grad_(True)
        k1 = torch.randn(N, C, H, W, generator=g)
        k2 = torch.randn(N, C, H, W, generator=g)
        wv = (torch.randn(C, C, generator=g) / C ** 0.5).requires_grad_(True)
        bv = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
        c1, c2 = crop_coords(N, Wo, Ho, rng), crop_coords(N, Wo, Ho, rng)
        c2[: N // 2, :4] = c1[: N // 2, :4] + 0.02 * torch.randn(N // 2, 4, generator=g)
        ff = smooth_flow((N, 2, Ho, Wo), 4.0, g)
        fb = -ff
        mf = torch.rand(N, Ho, Wo, generator=g) > 0.2
        mb = torch.rand(N, Ho, Wo, generator=g) > 0.2
        holder = types.SimpleNamespace(pixpro_p=p, pixpro_clamp_value=0.0,
                                       value_transform=lambda t: F.conv2d(t, wv.view(C, C, 1, 1), bv))
        pred1 = F.normalize(P.PixPro.featprop(holder, x1), dim=1)
        pred2 = F.normalize(P.PixPro.featprop(holder, x2), dim=1)
        k1n, k2n = F.normalize(k1, dim=1), F.normalize(k2, dim=1)
        a = [c1, [ff, (Ho, Wo), mf]]
        b = [c2, [fb, (Ho, Wo), mb]]
        l1 = P.regression_loss(pred1, k2n, a, b, 0.7)
        l2 = P.regression_loss(pred2, k1n, b, a, 0.7)
        loss = l1[0] + l2[0]
        loss.backward()
        save(f"hotpath_{tag}", x1=x1, x2=x2, k1=k1, k2=k2, wv=wv, bv=bv, coord1=c1, coord2=c2,
             flow_fwd=ff, flow_bwd=fb, mask_fwd=mf, mask_bwd=mb, size=np.array([Ho, Wo]), p=p,
             loss=loss, pos_num_1=l1[1][0], pos_num_2=l2[1][0], pos_mean_1=l1[1][1], pos_mean_2=l2[1][1],
             pred1=pred1, pred2=pred2, dx1=x1.grad, dx2=x2.grad, dwv=wv.grad, dbv=bv.grad)

    # ---------------- K5-K8: flow preparation -----------------------------------------
    from contrast.flow.utils.utils import upflow8
    g = torch.Generator().manual_seed(11)
    lo = smooth_flow((3, 2, 6, 8), 0.5, g)
    save("upflow8", flow=lo, out=upflow8(lo))

    for Fm1 in (1, 2, 5):
        g = torch.Generator().manual_seed(20 + Fm1)
        flows = smooth_flow((Fm1, 2, 2, 40, 56), 6.0, g)
        save(f"concat_flow_{Fm1}", flows=flows, out=util.concat_flow(flows),
             out_norm=util.concat_flow(flows, is_norm=True))

    g = torch.Generator().manual_seed(31)
    ff = smooth_flow((2, 2, 40, 56), 8.0, g)
    fb = -ff + smooth_flow((2, 2, 40, 56), 1.5, g)
    _, _, (m, cyc) = util.forward_backward_consistency(ff, fb, alpha_1=0.01, alpha_2=0.5)
    save("fbc", flow_fwd=ff, flow_bwd=fb, alpha1=0.01, alpha2=0.5, mask=m, cycle=cyc,
         ratio=util.calc_mask_ratio(m))

    for tag, (Fr, up, h, w) in {"f2up": (2, True, 6, 8), "f6up": (6, True, 6, 8), "f3noup": (3, False, 24, 32)}.items():
        g = torch.Generator().manual_seed(40 + Fr)
        N = 2
        fw = smooth_flow((N, Fr - 1, 2, h, w), 0.6 if up else 4.0, g)
        bw = -fw.flip(1) + smooth_flow((N, Fr - 1, 2, h, w), 0.1 if up else 0.8, g)
        Hf, Wf = (8 * h, 8 * w) if up else (h, w)
        size = torch.tensor([Hf, Wf])
        data = [None, None, None, None, None, [torch.zeros(N), fw, bw],
                [size.repeat(N, 1), torch.full((N, 1), Fr)]]
        args = types.SimpleNamespace(alpha1=0.01, alpha2=0.5, use_flow_frames=False, use_flow_file=True,
                                     flow_up=up, flow_cat_norm=False, debug=False)
        (f, s, mf), (b, _, mb) = util.apply_optical_flow(data, None, args)
        save(f"applyflow_{tag}", flow_fwds=fw, flow_bwds=bw, flow_up=up, alpha1=0.01, alpha2=0.5,
             flow_fwd=f, flow_bwd=b, mask_fwd=mf, mask_bwd=mb, size=s)

    # ---------------- LARS step and EMA -------------------------------------------------
    g = torch.Generator().manual_seed(50)
    shapes = [(16, 8, 3, 3), (16,), (16,), (32, 16, 1, 1), (32,)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    model = torch.nn.Module()
    for i, prm in enumerate(params):
        model.register_parameter(f"p{i}", prm)
    groups = lars.add_weight_decay(model, 1e-5)
    opt = lars.LARS(torch.optim.SGD(groups, lr=0.3, momentum=0.9))
    grads = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(3)]
    p0 = [prm.detach().clone() for prm in params]
    traj = []
    for step in range(3):
        for prm, gr in zip(params, grads[step]):
            prm.grad = gr.clone()
        opt.step()
        traj.append([prm.detach().clone() for prm in params])
    save("lars", **{f"p0_{i}": t for i, t in enumerate(p0)},
         **{f"g{s}_{i}": t for s in range(3) for i, t in enumerate(grads[s])},
         **{f"p{s + 1}_{i}": t for s in range(3) for i, t in enumerate(traj[s])},
         lr=0.3, momentum=0.9, weight_decay=1e-5, n=len(shapes))

    # ---------------- tiny full model step: BASELINE config 1 ---------------------------
    # ResNet-18, 64x64 crops, N=8, identity flow on a 320x180 frame; Proj_Head in_dim=512
    # (reference defect D3: in_dim is hard-coded to 2048).
    torch.manual_seed(0)
    orig_head = P.Proj_Head
    P.Proj_Head = lambda in_dim=512, inner_dim=4096, out_dim=256: orig_head(in_dim, inner_dim, out_dim)
    margs = types.SimpleNamespace(pixpro_p=2.0, pixpro_momentum=0.99, pixpro_pos_ratio=0.7,
                                  pixpro_clamp_value=0.0, pixpro_transform_layer=1,
                                  pixpro_ins_loss_weight=0.0, output_dir="/tmp", num_instances=800,
                                  batch_size=8, epochs=10, start_epoch=1, feature_dim=256,
                                  head_type="early_return")
    model = P.PixPro(resnet.resnet18, margs)
    P.Proj_Head = orig_head
    model.train()
    # state comes from a seeded recipe keyed by parameter name, so the test can rebuild the very
    # same 90 MB state dict without shipping it (and without relying on constructor RNG, defect D8)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(seeded_state_dict(shapes, 61))
    g = torch.Generator().manual_seed(61)
    N, Wo, Ho = 8, 320, 180
    im1 = torch.randn(N, 3, 64, 64, generator=g)
    im2 = torch.randn(N, 3, 64, 64, generator=g)
    c1, c2 = crop_coords(N, Wo, Ho, rng), crop_coords(N, Wo, Ho, rng)
    c2[:4, :4] = c1[:4, :4] + 0.02 * torch.randn(4, 4, generator=g)
    zf = torch.zeros(N, 2, Ho, Wo)
    loss, pos = model(im1, im2, [c1, [zf, (Ho, Wo), None]], [c2, [zf, (Ho, Wo), None]])
    loss.backward()
    keep = ["value_transform.weight", "value_transform.bias", "projector.linear2.weight",
            "projector.linear2.bias", "encoder.conv1.weight", "encoder.layer4.1.conv2.weight"]
    named = dict(model.named_parameters())
    sd1 = model.state_dict()
    save("model_cfg1", im1=im1, im2=im2, coord1=c1, coord2=c2, loss=loss,
         pos_num_1=pos[0][0], pos_num_2=pos[1][0], pos_mean_1=pos[0][1], pos_mean_2=pos[1][1],
         **{"grad:" + k: named[k].grad for k in (keep[0], keep[1], keep[3])},
         **{"gradslice:" + k: named[k].grad.flatten()[::97].clone() for k in keep},
         **{"gradsum:" + k: named[k].grad.double().abs().sum() for k in keep},
         **{"ema:" + k: sd1[k] for k in ["projector_k.linear2.bias", "encoder_k.bn1.weight"]},
         momentum_k=model.k, momentum_K=model.K)
    import json
    with open(os.path.join(HERE, "model_cfg1_shapes.json"), "w") as fh:
        json.dump({k: list(v) for k, v in shapes.items()}, fh)


if __name__ == "__main__":
    main()
