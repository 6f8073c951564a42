"""Model-level parity: BASELINE config 1 (ResNet-18, 64x64 crops, N=8, identity flow on a 320x180 frame)
on the GPU vs the loss / pos_num / gradients captured from the reference's own PixPro on CPU; EMA and
LARS multi-tensor kernels vs golden trajectories."""
import json
import os
import types

import pytest
import torch

from conftest import GOLDEN, load_golden
from gpu_utils import close

pytestmark = pytest.mark.gpu


def cfg1_args(**kw):
    base = dict(pixpro_p=2.0, pixpro_momentum=0.99, pixpro_pos_ratio=0.7, pixpro_clamp_value=0.0,
                pixpro_transform_layer=1, pixpro_ins_loss_weight=0.0, output_dir="/tmp", num_instances=800,
                batch_size=8, epochs=10, start_epoch=1, feature_dim=256, head_type="early_return", sync_bn=True)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_model_cfg1_step(dev):
    from contrast import resnet
    from contrast.models import PixPro
    from recipes import seeded_state_dict
    g = load_golden("model_cfg1")
    with open(os.path.join(GOLDEN, "model_cfg1_shapes.json")) as fh:
        shapes = {k: tuple(v) for k, v in json.load(fh).items()}
    model = PixPro(resnet.resnet18, cfg1_args())
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert mine == shapes                                   # checkpoint-compatible names and shapes
    model.load_state_dict(seeded_state_dict(shapes, 61))
    model = model.to(dev).train()
    Ho, Wo = 180, 320
    zf = torch.zeros(8, 2, Ho, Wo, device=dev)
    c1, c2 = g["coord1"].to(dev), g["coord2"].to(dev)
    loss, pos = model(g["im1"].to(dev), g["im2"].to(dev), [c1, [zf, (Ho, Wo), None]], [c2, [zf, (Ho, Wo), None]])
    loss.backward()
    assert torch.equal(pos[0][0].cpu(), g["pos_num_1"]) and torch.equal(pos[1][0].cpu(), g["pos_num_2"])
    assert abs(loss.item() - g["loss"]) <= 1e-4 * max(1.0, abs(g["loss"]))
    named = dict(model.named_parameters())
    for key in ("value_transform.weight", "value_transform.bias", "projector.linear2.bias"):
        close(named[key].grad, g["grad:" + key], rel=2e-3)
    close(named["projector.linear2.weight"].grad.flatten()[::97], g["gradslice:projector.linear2.weight"], rel=5e-3)
    # encoder gradients come out of PyTorch-ROCm (MIOpen convs + BatchNorm over only N*H*W = 32 values per
    # channel at 64x64 crops): fp32 re-association differences vs the CPU reference are amplified there
    for key in ("encoder.conv1.weight", "encoder.layer4.1.conv2.weight"):
        close(named[key].grad.flatten()[::97], g["gradslice:" + key], rel=3e-2)
    sd = model.state_dict()
    for key in ("projector_k.linear2.bias", "encoder_k.bn1.weight"):
        assert torch.allclose(sd[key].cpu(), g["ema:" + key], rtol=1e-6, atol=1e-7)
    assert model.k == g["momentum_k"] and model.K == g["momentum_K"]


def test_ema_matches_oracle(dev):
    from contrast.multi_tensor import ema_update
    from oracle import pixpro_oracle as O
    gen = torch.Generator().manual_seed(0)
    shapes = [(64, 3, 7, 7), (64,), (1,), (4096, 37), (5,), (129, 3, 3, 3)]
    pq = [torch.randn(*s, generator=gen) for s in shapes]
    pk = [torch.randn(*s, generator=gen) for s in shapes]
    m = O.momentum_coefficient(0.99, 123, 8750)
    want = O.momentum_update(pq, pk, m)
    dq = [t.to(dev) for t in pq]
    dk = [t.to(dev) for t in pk]
    # odd offsets exercise the unaligned path
    ema_update(dq, dk, m)
    for a, b in zip(dk, want):
        assert torch.equal(a.cpu(), b)


def test_multi_tensor_chunk_walk_edge_sizes(dev):
    """The chunked walk of the multi-tensor kernels (16 K-element chunks from an in-kernel prefix table): sizes around the
    chunk and float4 boundaries, many tiny tensors, a tensor of several chunks, and views that start at an odd element
    (scalar path) - EMA bit for bit against the oracle, LARS + SGD against it within the order of the norm sums."""
    from contrast.multi_tensor import ema_update, LarsSgdStep
    from oracle import pixpro_oracle as O
    gen = torch.Generator().manual_seed(3)
    sizes = [1, 2, 3, 4, 5, 255, 16383, 16384, 16385, 32768, 50001, 3 * 16384 + 7] + [7] * 180 + [64] * 120
    pq = [torch.randn(n, generator=gen) for n in sizes]
    pk = [torch.randn(n, generator=gen) for n in sizes]
    flat_q, flat_k = torch.randn(20011, generator=gen), torch.randn(20011, generator=gen)
    dq = [t.to(dev) for t in pq] + [flat_q.to(dev)[1:], flat_q.to(dev)[3:16390]]       # storage offsets 1 and 3: unaligned
    dk = [t.to(dev) for t in pk] + [flat_k.to(dev)[1:].clone(), flat_k.to(dev)[3:16390].clone()]
    base = torch.empty(20011 + 16390, device=dev)
    dk[-2] = base[1:20011].copy_(flat_k.to(dev)[1:])                                   # views into one buffer at odd offsets
    dk[-1] = base[20011 + 3:20011 + 16390].copy_(flat_k.to(dev)[3:16390])
    hq = pq + [flat_q[1:], flat_q[3:16390]]
    hk = pk + [flat_k[1:], flat_k[3:16390]]
    m = O.momentum_coefficient(0.99, 77, 8750)
    want = O.momentum_update(hq, hk, m)
    ema_update(dq, dk, m)
    for i, (a, b) in enumerate(zip(dk, want)):
        assert torch.equal(a.cpu(), b), (i, a.numel())
    # more tensors than one C call takes (2048): the wrapper splits the list
    many_q = [torch.randn(3 + i % 3, generator=gen) for i in range(2100)]
    many_k = [torch.randn(3 + i % 3, generator=gen) for i in range(2100)]
    dmq, dmk = [t.to(dev) for t in many_q], [t.to(dev) for t in many_k]
    ema_update(dmq, dmk, m)
    for a, b in zip(dmk, O.momentum_update(many_q, many_k, m)):
        assert torch.equal(a.cpu(), b)
    # LARS: every third tensor 1-D-style (no decay, no adaptation)
    ps = [torch.nn.Parameter(t.to(dev).clone()) for t in pq]
    wds = [0.0 if i % 3 == 0 else 1e-4 for i in range(len(ps))]
    ads = [i % 3 != 0 for i in range(len(ps))]
    step = LarsSgdStep(ps, wds, ads, momentum=0.9)
    ref_p = [t.clone() for t in pq]
    ref_b = [None] * len(ps)
    for it in range(2):
        gs = [torch.randn(n, generator=gen) for n in sizes]
        for p_, g_ in zip(ps, gs):
            p_.grad = g_.to(dev)
        step.step(0.3)
        for i in range(len(ps)):
            ref_p[i], ref_b[i] = O.lars_sgd_step(ref_p[i], gs[i], ref_b[i], 0.3, 0.9, wds[i], ads[i])
        for i, p_ in enumerate(ps):
            assert torch.allclose(p_.detach().cpu(), ref_p[i], rtol=3e-6, atol=1e-7), (it, i, sizes[i])


@pytest.mark.parametrize("count", [2047, 2048, 2049, 4096, 4100])
def test_multi_tensor_full_table(dev, count):
    """Lists around the 2048-entry chunk table of one C call: a full table (its total is written by the last thread, not
    by an entry of its own) and the wrapper's split just beyond it - EMA bit for bit, LARS + SGD with adaptive tensors on
    both sides of every split, two steps (first_step is shared by the sub-calls of a step)."""
    from contrast.multi_tensor import ema_update, LarsSgdStep
    from oracle import pixpro_oracle as O
    gen = torch.Generator().manual_seed(count)
    sizes = [2 + (i * 7) % 11 for i in range(count)]
    for i in (0, 2046, 2047, 2048, count - 1):           # a few tensors of more than one chunk next to the boundaries
        if i < count:
            sizes[i] = 16384 + 5 + i % 3
    pq = [torch.randn(n, generator=gen) for n in sizes]
    pk = [torch.randn(n, generator=gen) for n in sizes]
    m = O.momentum_coefficient(0.99, 5, 8750)
    dq, dk = [t.to(dev) for t in pq], [t.to(dev) for t in pk]
    ema_update(dq, dk, m)
    for i, (a, b) in enumerate(zip(dk, O.momentum_update(pq, pk, m))):
        assert torch.equal(a.cpu(), b), i
    ps = [torch.nn.Parameter(t.to(dev).clone()) for t in pq]
    wds = [0.0 if i % 5 == 0 else 1e-4 for i in range(count)]
    ads = [i % 5 != 0 for i in range(count)]
    step = LarsSgdStep(ps, wds, ads, momentum=0.9)
    ref_p, ref_b = [t.clone() for t in pq], [None] * count
    for it in range(2):
        gs = [torch.randn(n, generator=gen) for n in sizes]
        for p_, g_ in zip(ps, gs):
            p_.grad = g_.to(dev)
        step.step(0.3)
        for i in range(count):
            ref_p[i], ref_b[i] = O.lars_sgd_step(ref_p[i], gs[i], ref_b[i], 0.3, 0.9, wds[i], ads[i])
        for i, p_ in enumerate(ps):
            assert torch.allclose(p_.detach().cpu(), ref_p[i], rtol=3e-6, atol=1e-7), (it, i, sizes[i])


def test_lars_golden_trajectory(dev):
    from contrast.lars import LARS, add_weight_decay
    g = load_golden("lars")
    n = g["n"]
    model = torch.nn.Module()
    params = []
    for i in range(n):
        p = torch.nn.Parameter(g[f"p0_{i}"].to(dev))
        model.register_parameter(f"p{i}", p)
        params.append(p)
    opt = LARS(torch.optim.SGD(add_weight_decay(model, g["weight_decay"]), lr=g["lr"], momentum=g["momentum"]))
    for s in range(3):
        opt.zero_grad()
        for i, p in enumerate(params):
            p.grad = g[f"g{s}_{i}"].to(dev)
        opt.step()
        for i, p in enumerate(params):
            assert torch.allclose(p.detach().cpu(), g[f"p{s + 1}_{i}"], rtol=2e-6, atol=1e-7), (s, i)


def test_train_step_runs_and_learns(dev):
    """Three synthetic steps of the main_pretrain step (flow prep + fused hot path + LARS + EMA): finite,
    no exceptions, parameters and key encoder move."""
    import main_pretrain
    from contrast import option
    from contrast.lr_scheduler import get_scheduler
    from contrast.synthetic import make_batch
    args = option.parse_option(argv=["--model", "PixPro", "--arch", "resnet18", "--synthetic", "--use_flow",
                                     "--use_flow_file", "--flow_up", "--alpha1", "0.01", "--alpha2", "0.5",
                                     "--n-frames", "3", "--optimizer", "lars", "--base-lr", "1.0",
                                     "--weight-decay", "1e-5", "--amp-opt-level", "O0", "--pixpro-p", "2",
                                     "--pixpro-transform-layer", "1", "--batch-size", "4", "--epochs", "2",
                                     "--warmup-epoch", "0", "--head-type", "early_return"])
    args.num_instances = 64
    model, opt = main_pretrain.build_model(args, dev)
    sched = get_scheduler(opt, 4, args)
    model.train()
    data = make_batch(4, image_size=64, n_frames=3, frame_hw=(96, 128), device=dev, sigma=0.5)
    w0 = model.value_transform.weight.detach().clone()
    k0 = model.projector_k.linear2.weight.detach().clone()
    for _ in range(3):
        loss, pos, info = main_pretrain.train_step(model, opt, sched, data, args)
        assert torch.isfinite(loss).item()
    assert not torch.equal(w0, model.value_transform.weight)
    assert not torch.equal(k0, model.projector_k.linear2.weight)
    assert info["mask_fwd"].dtype == torch.bool


def test_checkpoint_resume_roundtrip(dev, tmp_path):
    """save_checkpoint / load_checkpoint (main_pretrain.py:96-127 of the reference): model, LARS momentum buffers,
    scheduler and the EMA step counter survive; the resumed run reproduces the next step bit for bit."""
    import main_pretrain
    from contrast import option
    from contrast.lr_scheduler import get_scheduler
    from contrast.synthetic import make_batch

    def build():
        args = option.parse_option(argv=["--model", "PixPro", "--arch", "resnet18", "--synthetic", "--use_flow",
                                         "--use_flow_file", "--flow_up", "--alpha1", "0.01", "--alpha2", "0.5",
                                         "--n-frames", "2", "--optimizer", "lars", "--base-lr", "1.0",
                                         "--weight-decay", "1e-5", "--amp-opt-level", "O0", "--pixpro-p", "2",
                                         "--pixpro-transform-layer", "1", "--batch-size", "4", "--epochs", "4",
                                         "--warmup-epoch", "1", "--head-type", "early_return",
                                         "--output-dir", str(tmp_path)])
        args.num_instances = 16
        torch.manual_seed(0)
        model, opt = main_pretrain.build_model(args, dev)
        return args, model, opt, get_scheduler(opt, 4, args)

    data = make_batch(4, image_size=64, n_frames=2, frame_hw=(96, 128), device=dev)
    args, model, opt, sched = build()
    model.train()
    # one whole epoch of the EMA schedule: num_instances / batch = 4 steps (PixPro.py:294-295 counts k from the epoch)
    for _ in range(4):
        main_pretrain.train_step(model, opt, sched, data, args)
    main_pretrain.save_checkpoint(args, 1, model, opt, sched)
    k_at_save, m_at_save, lr_at_save = model.k, model.momentum_coefficient(), opt.param_groups[0]["lr"]
    assert k_at_save == 4
    loss_a, _, _ = main_pretrain.train_step(model, opt, sched, data, args)
    w_a = model.value_transform.weight.detach().clone()

    args2, model2, opt2, sched2 = build()
    args2.resume = str(tmp_path / "current.pth")
    main_pretrain.load_checkpoint(args2, model2, opt2, sched2)
    assert args2.start_epoch == 2
    # the counter load_checkpoint restores is the one the interrupted run had: same momentum for the next EMA update
    assert model2.k == k_at_save and model2.momentum_coefficient() == m_at_save
    assert sched2.last_epoch == 4 and lr_at_save == pytest.approx(opt2.param_groups[0]["lr"], rel=1e-12)
    model2.train()
    loss_b, _, _ = main_pretrain.train_step(model2, opt2, sched2, data, args2)
    # MIOpen's weight-gradient kernels and the LARS norm reduction accumulate with float atomics: equal up to
    # re-association, not bit for bit
    assert abs(loss_a.item() - loss_b.item()) <= 1e-5 * max(1.0, abs(loss_a.item()))
    assert torch.allclose(w_a, model2.value_transform.weight.detach(), rtol=1e-4, atol=1e-6)
    assert model2.k == model.k == 5


def test_public_featprop_and_instance_branch(dev):
    """PixPro.featprop as a public method (autograd through the HIP op) and the optional instance branch."""
    from contrast import resnet
    from contrast.models import PixPro
    from oracle import pixpro_oracle as O
    model = PixPro(resnet.resnet18, cfg1_args(pixpro_ins_loss_weight=0.5)).to(dev).train()
    feat = torch.randn(2, 256, 4, 4, device=dev, requires_grad=True)
    Y = model.featprop(feat)
    Y.square().sum().backward()
    xr = feat.detach().cpu().requires_grad_(True)
    w, b = model.value_transform.weight.detach().cpu(), model.value_transform.bias.detach().cpu()
    Yo = O.featprop(xr, torch.nn.functional.conv2d(xr, w, b), 2.0, 0.0)
    Yo.square().sum().backward()
    close(Y, Yo.detach(), rel=1e-5)
    close(feat.grad, xr.grad, rel=2e-4)
    im = torch.randn(2, 3, 64, 64, device=dev)
    c = torch.tensor([[0.1, 0.1, 0.6, 0.7, 0, 0, 0, 0, 320, 180]] * 2, device=dev)
    loss, pos = model(im, im, c, c)
    loss.backward()
    assert torch.isfinite(loss).item() and model.projector_instance.linear2.weight.grad is not None


def test_train_step_lazy_flow_matches_dense(dev):
    """main_pretrain.train_step with --lazy-flow gives the same loss / pos_num as the dense flow preparation."""
    import main_pretrain
    from contrast import option
    from contrast.lr_scheduler import get_scheduler
    from contrast.synthetic import make_batch
    base = ["--model", "PixPro", "--arch", "resnet18", "--synthetic", "--use_flow", "--use_flow_file", "--flow_up",
            "--alpha1", "0.01", "--alpha2", "0.5", "--n-frames", "2", "--optimizer", "sgd", "--base-lr", "0.1",
            "--amp-opt-level", "O0", "--pixpro-p", "2", "--pixpro-transform-layer", "1", "--batch-size", "4",
            "--epochs", "2", "--warmup-epoch", "0", "--head-type", "early_return"]
    data = make_batch(4, image_size=64, n_frames=2, frame_hw=(96, 128), device=dev, sigma=0.8)
    out = []
    for extra in ([], ["--lazy-flow"]):
        args = option.parse_option(argv=base + extra)
        args.num_instances = 16
        torch.manual_seed(0)
        model, opt = main_pretrain.build_model(args, dev)
        sched = get_scheduler(opt, 4, args)
        model.train()
        loss, pos, info = main_pretrain.train_step(model, opt, sched, data, args)
        out.append((loss.item(), pos[0][0].clone(), pos[1][0].clone()))
    # the hot path itself is bit-identical (test_lazy_flow_equals_dense); two separately built encoders are not
    # (MIOpen fp32 convolutions re-associate from run to run), hence a tolerance on the loss
    assert abs(out[0][0] - out[1][0]) <= 1e-5 * max(1.0, abs(out[0][0]))
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    assert out[0][1].sum() > 0


def test_train_step_fused_bn_and_checkpoint_interchange(dev, tmp_path):
    """main_pretrain with --fused-bn --channels-last: the step runs, the loss falls, and its checkpoint loads into a
    model built WITHOUT the fused modules (same state-dict names), giving the same evaluation-mode features."""
    import main_pretrain
    from contrast import option
    from contrast.lr_scheduler import get_scheduler
    from contrast.synthetic import make_batch
    from contrast.sync_bn import FusedSyncBatchNorm
    base = ["--model", "PixPro", "--arch", "resnet18", "--synthetic", "--use_flow", "--use_flow_file", "--flow_up",
            "--alpha1", "0.01", "--alpha2", "0.5", "--n-frames", "2", "--optimizer", "lars", "--base-lr", "1.0",
            "--weight-decay", "1e-5", "--amp-opt-level", "O0", "--pixpro-p", "2", "--pixpro-transform-layer", "1",
            "--batch-size", "8", "--epochs", "2", "--warmup-epoch", "0", "--head-type", "early_return",
            "--output-dir", str(tmp_path)]
    data = make_batch(8, image_size=64, n_frames=2, frame_hw=(96, 128), device=dev, sigma=0.8, channels_last=True)
    args = option.parse_option(argv=base + ["--fused-bn", "--channels-last"])
    args.num_instances = 32
    torch.manual_seed(0)
    model, opt = main_pretrain.build_model(args, dev)
    assert sum(isinstance(m, FusedSyncBatchNorm) for m in model.modules()) == 2 * (20 + 1)
    sched = get_scheduler(opt, 4, args)
    model.train()
    losses = [main_pretrain.train_step(model, opt, sched, data, args)[0].item() for _ in range(6)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]
    main_pretrain.save_checkpoint(args, 1, model, opt, sched)

    args2 = option.parse_option(argv=base)
    args2.num_instances = 32
    model2, opt2 = main_pretrain.build_model(args2, dev)
    assert list(model2.state_dict().keys()) == list(model.state_dict().keys())
    args2.resume = str(tmp_path / "current.pth")
    main_pretrain.load_checkpoint(args2, model2, opt2, get_scheduler(opt2, 4, args2))
    model.eval(), model2.eval()
    im = data[0]
    with torch.no_grad():
        a, b = model.encoder(im), model2.encoder(im)
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_overlapped_momentum_branch_gives_the_same_step(dev):
    """--overlap-momentum only changes WHERE the key branch runs (a second stream): after three steps the loss, the
    positive counts and the EMA-updated key weights are those of the sequential schedule."""
    import main_pretrain
    from contrast import option
    from contrast.lr_scheduler import get_scheduler
    from contrast.synthetic import make_batch
    base = ["--model", "PixPro", "--arch", "resnet18", "--synthetic", "--use_flow", "--use_flow_file", "--flow_up",
            "--alpha1", "0.01", "--alpha2", "0.5", "--n-frames", "2", "--optimizer", "lars", "--base-lr", "1.0",
            "--weight-decay", "1e-5", "--amp-opt-level", "O0", "--pixpro-p", "2", "--pixpro-transform-layer", "1",
            "--batch-size", "8", "--epochs", "2", "--warmup-epoch", "0", "--head-type", "early_return", "--fused-bn",
            "--channels-last", "--pixpro-ins-loss-weight", "0.5"]
    data = make_batch(8, image_size=64, n_frames=2, frame_hw=(96, 128), device=dev, sigma=0.8, channels_last=True)
    out = []
    for extra in ([], ["--overlap-momentum"]):
        args = option.parse_option(argv=base + extra)
        args.num_instances = 32
        torch.manual_seed(0)
        model, opt = main_pretrain.build_model(args, dev)
        assert model.overlap_momentum == bool(extra)
        sched = get_scheduler(opt, 4, args)
        model.train()
        for _ in range(3):
            loss, pos, _ = main_pretrain.train_step(model, opt, sched, data, args)
        torch.cuda.synchronize()
        out.append((loss.item(), pos[0][0].clone(), model.encoder_k.conv1.weight.detach().clone(),
                    model.encoder_k.bn1.running_mean.detach().clone()))
    assert abs(out[0][0] - out[1][0]) <= 1e-4 * max(1.0, abs(out[0][0]))
    assert torch.equal(out[0][1], out[1][1])
    assert torch.allclose(out[0][2], out[1][2], rtol=1e-4, atol=1e-6)
    assert torch.allclose(out[0][3], out[1][3], rtol=1e-4, atol=1e-6)
