"""FusedSyncBatchNorm (csrc/sync_bn.hip through the C ABI) against torch's own batch norm + add + ReLU in fp32:
forward, all gradients, running statistics, two gloo ranks sharing the GPU against one process on the whole batch,
and the PixPro model with / without the fused modules.  Tolerances: outputs 2e-5 abs (values are O(1)), gradients
1e-4 of the reference's max-abs, statistics 1e-5 relative."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from conftest import PKG, ROOT
from gpu_utils import close

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _reference(x, w, b, res, relu, rm, rv, momentum=0.1, eps=1e-5):
    """torch's batch norm + add + ReLU.  Double inputs give the exact-arithmetic reference (MIOpen's fp32 kernels lose
    digits of the variance when a channel holds a handful of nearly equal values, e.g. N=2 at 1x1)."""
    y = F.batch_norm(x, rm, rv, w, b, True, momentum, eps)
    if res is not None:
        y = y + res
    if relu is True:
        return F.relu(y)
    # a mask instead of True: the ReLU decisions of the implementation under test (an output within rounding of zero may
    # sit on either side; with ~10^2 values per channel one flipped decision moves that channel's sums visibly)
    return y * relu.to(y.dtype) if isinstance(relu, torch.Tensor) else y


@pytest.mark.parametrize("shape", [(8, 64, 14, 14), (4, 256, 7, 7), (3, 2048, 7, 7), (16, 64, 56, 56), (32, 4096, 1, 1),
                                   (5, 36, 9, 11), (2, 1024, 14, 14), (64, 4, 3, 3)])
@pytest.mark.parametrize("relu,with_res", [(False, False), (True, False), (True, True), (False, True)])
def test_forward_backward_match_torch(dev, shape, relu, with_res):
    from contrast.sync_bn import FusedSyncBatchNorm
    g = torch.Generator().manual_seed(sum(shape) + 2 * relu + with_res)
    N, C, H, W = shape
    x = (torch.randn(shape, generator=g) * 1.7 + 0.4).to(dev).contiguous(memory_format=CL)
    res = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=CL) if with_res else None
    dy = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=CL)
    bn = FusedSyncBatchNorm(C).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    rm, rv = bn.running_mean.double(), bn.running_var.double()
    w0, b0 = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    xf = x.clone().requires_grad_(True)
    rf = res.clone().requires_grad_(True) if with_res else None
    got = bn(xf, rf, relu)
    got.backward(dy)
    want = _reference(xr, w0, b0, rr, (got > 0) if relu else False, rm, rv)
    want.backward(dy.double())
    assert got.is_contiguous(memory_format=CL) and got.shape == x.shape
    assert (got - want).abs().max().item() <= 2e-5
    close(xf.grad, xr.grad)
    close(bn.weight.grad, w0.grad, rel=2e-4)
    close(bn.bias.grad, b0.grad, rel=2e-4)
    if with_res:
        close(rf.grad, rr.grad)
    assert torch.allclose(bn.running_mean.double(), rm, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), rv, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1


def test_statistics_survive_a_large_mean(dev):
    """Shifted sums: a channel mean 1000x its standard deviation must not destroy the variance."""
    from contrast.sync_bn import FusedSyncBatchNorm
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(32, 64, 28, 28, generator=g) * 0.05 + 50.0).to(dev).contiguous(memory_format=CL)
    bn = FusedSyncBatchNorm(64).to(dev).train()
    y = bn(x)
    x64 = x.double()
    mu = x64.mean((0, 2, 3), keepdim=True)
    var = x64.var((0, 2, 3), unbiased=False, keepdim=True)
    want = ((x64 - mu) / torch.sqrt(var + 1e-5)).float()
    assert (y - want).abs().max().item() <= 2e-3          # fp32 input rounding at |x| = 50 is 4e-6 / 0.05 std already
    assert torch.allclose(bn.running_var, (0.9 + 0.1 * x64.var((0, 2, 3))).float(), rtol=1e-4)


def test_eval_mode_and_errors(dev):
    from contrast.sync_bn import FusedSyncBatchNorm
    bn = FusedSyncBatchNorm(8).to(dev)
    x = torch.randn(4, 8, 5, 5, device=dev).contiguous(memory_format=CL)
    bn.train()
    bn(x)
    bn.eval()
    want = F.relu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, 1e-5) + x)
    assert torch.allclose(bn(x, x, True), want)
    bn.train()
    with pytest.raises(RuntimeError):
        bn(x.cpu())
    with pytest.raises(RuntimeError):
        FusedSyncBatchNorm(6).to(dev).train()(torch.randn(2, 6, 3, 3, device=dev))   # C not a multiple of 4
    nchw = torch.randn(4, 8, 5, 5, device=dev)
    assert torch.allclose(bn(nchw), bn(nchw.contiguous(memory_format=CL)), atol=1e-6)  # converted, not rejected


def test_convert_shares_parameters_and_names(dev):
    from contrast import resnet
    from contrast.sync_bn import FusedSyncBatchNorm
    plain = resnet.resnet18(head_type="early_return")
    names = list(plain.state_dict().keys())
    fused = FusedSyncBatchNorm.convert(plain)
    assert list(fused.state_dict().keys()) == names
    assert all(not (isinstance(m, nn.BatchNorm2d) and not isinstance(m, FusedSyncBatchNorm)) for m in fused.modules())
    assert sum(isinstance(m, FusedSyncBatchNorm) for m in fused.modules()) == 20


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, path):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from contrast.sync_bn import FusedSyncBatchNorm
    dev = torch.device("cuda:0")
    d = torch.load(path)
    sl = slice(0, 7) if rank == 0 else slice(7, d["x"].shape[0])      # unequal shards: 7 and 5 samples
    bn = FusedSyncBatchNorm(d["x"].shape[1]).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(d["w"]); bn.bias.copy_(d["b"])
    x = d["x"][sl].to(dev).contiguous(memory_format=CL).requires_grad_(True)
    res = d["res"][sl].to(dev).contiguous(memory_format=CL).requires_grad_(True)
    y = bn(x, res, True)
    y.backward(d["dy"][sl].to(dev).contiguous(memory_format=CL))
    torch.save(dict(y=y.detach().cpu(), dx=x.grad.cpu(), dres=res.grad.cpu(), dw=bn.weight.grad.cpu(), db=bn.bias.grad.cpu(),
                    rm=bn.running_mean.cpu(), rv=bn.running_var.cpu()), f"{path}.out{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_whole_batch(dev, tmp_path):
    """The SyncBN contract: statistics and input gradients are those of the global batch - also when the ranks hold
    different numbers of samples; parameter gradients are the local sums (DDP averages them afterwards)."""
    g = torch.Generator().manual_seed(11)
    shape = (12, 128, 9, 9)
    d = dict(x=torch.randn(shape, generator=g) * 2 + 1, res=torch.randn(shape, generator=g), dy=torch.randn(shape, generator=g),
             w=torch.rand(128, generator=g) + 0.5, b=torch.randn(128, generator=g))
    d["x"][7:] += 0.7                                   # the two shards have different statistics (and sizes: 7 + 5)
    path = str(tmp_path / "case.pt")
    torch.save(d, path)
    mp.spawn(_rank, args=(2, _free_port(), path), nprocs=2, join=True)
    outs = [torch.load(f"{path}.out{r}") for r in range(2)]
    x = d["x"].to(dev).contiguous(memory_format=CL).requires_grad_(True)
    res = d["res"].to(dev).contiguous(memory_format=CL).requires_grad_(True)
    w, b = d["w"].to(dev).requires_grad_(True), d["b"].to(dev).requires_grad_(True)
    rm, rv = torch.zeros(128, device=dev), torch.ones(128, device=dev)
    y = torch.cat([o["y"] for o in outs]).to(dev)
    want = _reference(x, w, b, res, y > 0, rm, rv)
    want.backward(d["dy"].to(dev).contiguous(memory_format=CL))
    assert (y - want).abs().max().item() <= 2e-5
    close(torch.cat([o["dx"] for o in outs]).to(dev), x.grad)
    close(torch.cat([o["dres"] for o in outs]).to(dev), res.grad)
    close(outs[0]["dw"] + outs[1]["dw"], w.grad, rel=2e-4)
    close(outs[0]["db"] + outs[1]["db"], b.grad, rel=2e-4)
    for o in outs:
        assert torch.allclose(o["rm"].to(dev), rm, rtol=1e-5, atol=1e-6) and torch.allclose(o["rv"].to(dev), rv, rtol=1e-5, atol=1e-6)


def _rank_model(rank, world, port, path):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from contrast import resnet
    from contrast.models import PixPro
    from contrast.synthetic import make_batch
    dev = torch.device("cuda:0")
    out = {}
    for own_group in (False, True):
        args = types.SimpleNamespace(pixpro_p=2.0, pixpro_momentum=0.99, pixpro_pos_ratio=0.7, pixpro_clamp_value=0.0,
                                     pixpro_transform_layer=1, pixpro_ins_loss_weight=0.0, output_dir="/tmp", num_instances=64,
                                     batch_size=4, epochs=2, start_epoch=1, feature_dim=256, head_type="early_return",
                                     sync_bn=True, fused_bn=True, key_bn_group=own_group, overlap_momentum=own_group)
        torch.manual_seed(0)
        model = PixPro(resnet.__dict__["resnet18"], args).to(dev).to(memory_format=CL).train()
        data = make_batch(4, image_size=64, n_frames=2, frame_hw=(48, 64), device=dev, seed=1234 + rank, flow_up=False)
        size = (48, 64)
        c1 = [data[2], [data[5][1][:, 0], size, None]]
        c2 = [data[3], [data[5][2][:, 0], size, None]]
        loss, _ = model(data[0].contiguous(memory_format=CL), data[1].contiguous(memory_format=CL), c1, c2)
        loss.backward()
        torch.cuda.synchronize()
        bn_k = next(m for m in model.encoder_k.modules() if getattr(m, "fused", False))
        out[own_group] = dict(loss=loss.item(), g=model.value_transform.weight.grad.cpu().clone(),
                              rm=bn_k.running_mean.cpu().clone(), has_group=model.key_bn_group is not None)
    if rank == 0:
        torch.save(out, path)
    dist.barrier()
    dist.destroy_process_group()


def test_key_branch_on_its_own_process_group_gives_the_same_step(dev, tmp_path):
    """--key-bn-group with the key branch on its second stream, two ranks (gloo, sharing the GPU): the SyncBN statistics of
    the key encoder travel over a second process group - loss, gradients and running statistics are those of the plain
    step (the collectives of the two branches no longer share one queue)."""
    path = str(tmp_path / "model.pt")
    mp.spawn(_rank_model, args=(2, _free_port(), path), nprocs=2, join=True)
    r = torch.load(path)
    assert r[True]["has_group"] and not r[False]["has_group"]
    assert abs(r[True]["loss"] - r[False]["loss"]) <= 1e-6 * max(1.0, abs(r[False]["loss"]))
    close(r[True]["g"], r[False]["g"], rel=1e-5)
    assert torch.equal(r[True]["rm"], r[False]["rm"])


def test_model_step_with_fused_bn_matches_stock(dev):
    """ResNet-18 PixPro step (cfg-1 shapes), same weights and inputs, stock BatchNorm + ReLU vs the fused modules."""
    import types
    import importlib
    pixpro_mod = importlib.import_module("contrast.models.PixPro")
    from contrast import resnet
    from contrast.synthetic import make_batch
    out = []
    for fused in (False, True):
        torch.manual_seed(0)
        margs = types.SimpleNamespace(pixpro_p=2.0, pixpro_momentum=0.99, pixpro_pos_ratio=0.7, pixpro_clamp_value=0.0,
                                      pixpro_transform_layer=1, pixpro_ins_loss_weight=0.0, output_dir="/tmp",
                                      num_instances=70000, batch_size=8, epochs=2000, start_epoch=1, feature_dim=256,
                                      head_type="early_return", sync_bn=True, fused_bn=fused)
        model = pixpro_mod.PixPro(resnet.resnet18, margs).to(dev).to(memory_format=CL).train()
        d = make_batch(8, image_size=64, n_frames=2, frame_hw=(180, 320), device=dev, seed=7, with_flow=False)
        im1, im2 = d[0].contiguous(memory_format=CL), d[1].contiguous(memory_format=CL)
        loss, (a, b) = model(im1, im2, d[2], d[3])
        loss.backward()
        out.append((loss.item(), a[0].clone(), model.encoder.conv1.weight.grad.clone(),
                    model.projector.linear2.weight.grad.clone(), model.encoder.bn1.running_var.clone()))
    assert abs(out[0][0] - out[1][0]) <= 1e-5 and torch.equal(out[0][1], out[1][1])
    close(out[1][3], out[0][3], rel=2e-3)
    close(out[1][2], out[0][2], rel=3e-2)       # 53 normalisations deep; re-association noise amplified by BN at 32 values
    assert torch.allclose(out[0][4], out[1][4], rtol=1e-4)


@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False)])
def test_view_groups_equal_two_separate_passes(dev, relu, with_res):
    """view_groups(2): one call over [view 1; view 2] == two calls, bit for bit - outputs, every gradient, and the running
    statistics after the two sequential updates (the reference makes two passes, PixPro.py:376-384)."""
    from contrast.sync_bn import FusedSyncBatchNorm, view_groups
    g = torch.Generator().manual_seed(5 + relu + 2 * with_res)
    N, C, H, W = 6, 64, 9, 7

    def mk():
        return [(torch.randn(N, C, H, W, generator=g) * s + m).to(dev).contiguous(memory_format=CL) for s, m in ((1.5, 0.3), (0.7, -1.0))]
    xs, rs, dys = mk(), mk(), mk()
    out = []
    for batched in (False, True):
        bn = FusedSyncBatchNorm(C).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
        xa = [x.clone().requires_grad_(True) for x in xs]
        ra = [r.clone().requires_grad_(True) for r in rs] if with_res else [None, None]
        if batched:
            x2 = torch.cat(xa)
            r2 = torch.cat(ra) if with_res else None
            with view_groups(2):
                y = bn(x2, r2, relu)
            y.backward(torch.cat(dys))
            ys = list(y.chunk(2))
        else:
            ys = [bn(xa[0], ra[0], relu), bn(xa[1], ra[1], relu)]
            (ys[0] * dys[0]).sum().add((ys[1] * dys[1]).sum()).backward()
        out.append(dict(y=[t.detach().clone() for t in ys], dx=[t.grad.clone() for t in xa],
                        dr=[t.grad.clone() for t in ra] if with_res else [], dw=bn.weight.grad.clone(), db=bn.bias.grad.clone(),
                        rm=bn.running_mean.clone(), rv=bn.running_var.clone(), nb=int(bn.num_batches_tracked)))
    a, b = out
    for k in ("y", "dx", "dr"):
        for u, v in zip(a[k], b[k]):
            assert torch.equal(u, v), k
    assert torch.equal(a["rm"], b["rm"]) and torch.equal(a["rv"], b["rv"]) and a["nb"] == b["nb"] == 2
    assert torch.allclose(a["dw"], b["dw"], rtol=1e-6, atol=1e-7) and torch.allclose(a["db"], b["db"], rtol=1e-6, atol=1e-7)


def _rank_views(rank, world, port, path):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from contrast.sync_bn import FusedSyncBatchNorm, view_groups
    dev = torch.device("cuda:0")
    d = torch.load(path)
    calls = {"all_gather": 0, "all_reduce": 0}
    real_ag, real_ar = dist.all_gather, dist.all_reduce

    def ag(*a, **k):
        calls["all_gather"] += 1
        return real_ag(*a, **k)

    def ar(*a, **k):
        calls["all_reduce"] += 1
        return real_ar(*a, **k)
    dist.all_gather, dist.all_reduce = ag, ar
    sl = slice(0, 4) if rank == 0 else slice(4, 6)                    # unequal shards: 4 and 2 samples per view
    res = {}
    for batched in (False, True):
        calls.update(all_gather=0, all_reduce=0)
        bn = FusedSyncBatchNorm(d["x"].shape[2]).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(d["w"]); bn.bias.copy_(d["b"])
        xs = [d["x"][v][sl].to(dev).contiguous(memory_format=CL).requires_grad_(True) for v in range(2)]
        dys = [d["dy"][v][sl].to(dev).contiguous(memory_format=CL) for v in range(2)]
        if batched:
            with view_groups(2):
                y = bn(torch.cat(xs), None, True)
            y.backward(torch.cat(dys))
            ys = list(y.chunk(2))
        else:
            ys = [bn(xs[0], None, True), bn(xs[1], None, True)]
            (ys[0] * dys[0]).sum().add((ys[1] * dys[1]).sum()).backward()
        res[batched] = dict(y=[t.detach().cpu() for t in ys], dx=[t.grad.cpu() for t in xs], rm=bn.running_mean.cpu(),
                            rv=bn.running_var.cpu(), calls=dict(calls))
    dist.all_gather, dist.all_reduce = real_ag, real_ar
    torch.save(res, f"{path}.out{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_view_groups_halve_the_collectives_over_two_ranks(dev, tmp_path):
    """Two gloo ranks sharing the GPU (unequal shards): batching the two views of a layer sends ONE all_gather and ONE
    all_reduce instead of two each, and changes no value (statistics stay per view and global over the ranks)."""
    g = torch.Generator().manual_seed(21)
    shape = (2, 6, 32, 5, 5)                             # [view, sample, C, H, W]
    d = dict(x=torch.randn(shape, generator=g) * 2 + 1, dy=torch.randn(shape, generator=g), w=torch.rand(32, generator=g) + 0.5,
             b=torch.randn(32, generator=g))
    d["x"][1] -= 1.5                                     # the views have different statistics
    path = str(tmp_path / "views.pt")
    torch.save(d, path)
    mp.spawn(_rank_views, args=(2, _free_port(), path), nprocs=2, join=True)
    for r in range(2):
        o = torch.load(f"{path}.out{r}")
        assert o[False]["calls"] == {"all_gather": 2, "all_reduce": 2}
        assert o[True]["calls"] == {"all_gather": 1, "all_reduce": 1}
        for k in ("y", "dx"):
            for u, v in zip(o[False][k], o[True][k]):
                assert torch.equal(u, v), k
        assert torch.equal(o[False]["rm"], o[True]["rm"]) and torch.equal(o[False]["rv"], o[True]["rv"])


def test_model_step_with_view_batching_matches_two_passes(dev):
    """PixPro step with --bn-view-batch (one pass per encoder over both views) vs the reference's two passes: same loss,
    pos_num, gradients and running statistics up to the convolutions' batch-size dependent summation order."""
    import types
    import importlib
    pixpro_mod = importlib.import_module("contrast.models.PixPro")
    from contrast import resnet
    from contrast.synthetic import make_batch
    out = []
    for vb in (False, True):
        torch.manual_seed(0)
        margs = types.SimpleNamespace(pixpro_p=2.0, pixpro_momentum=0.99, pixpro_pos_ratio=0.7, pixpro_clamp_value=0.0,
                                      pixpro_transform_layer=1, pixpro_ins_loss_weight=0.0, output_dir="/tmp",
                                      num_instances=70000, batch_size=8, epochs=2000, start_epoch=1, feature_dim=256,
                                      head_type="early_return", sync_bn=True, fused_bn=True, bn_view_batch=vb)
        model = pixpro_mod.PixPro(resnet.resnet18, margs).to(dev).to(memory_format=CL).train()
        d = make_batch(8, image_size=64, n_frames=2, frame_hw=(180, 320), device=dev, seed=7, with_flow=False)
        im1, im2 = d[0].contiguous(memory_format=CL), d[1].contiguous(memory_format=CL)
        loss, (a, b) = model(im1, im2, d[2], d[3])
        loss.backward()
        out.append((loss.item(), a[0].clone(), model.encoder.conv1.weight.grad.clone(),
                    model.projector.linear2.weight.grad.clone(), model.encoder.bn1.running_var.clone(),
                    model.encoder_k.layer4[1].bn2.running_mean.clone(), int(model.encoder.bn1.num_batches_tracked)))
    assert abs(out[0][0] - out[1][0]) <= 1e-5 and torch.equal(out[0][1], out[1][1])
    close(out[1][3], out[0][3], rel=2e-3)
    close(out[1][2], out[0][2], rel=3e-2)
    assert torch.allclose(out[0][4], out[1][4], rtol=1e-5) and torch.allclose(out[0][5], out[1][5], rtol=1e-4, atol=1e-6)
    assert out[0][6] == out[1][6] == 2
    with pytest.raises(ValueError, match="needs --fused-bn"):
        pixpro_mod.PixPro(resnet.resnet18, types.SimpleNamespace(**{**vars(margs), "fused_bn": False}))


def test_flush_counters_makes_buffer_walks_current(dev):
    """`num_batches_tracked` is incremented lazily; readers that walk buffers() directly must call
    sync_bn.flush_counters first (attribute access and state_dict flush by themselves)."""
    from contrast.sync_bn import FusedSyncBatchNorm, flush_counters
    net = torch.nn.Sequential(FusedSyncBatchNorm(8), FusedSyncBatchNorm(8)).to(dev).train()
    x = torch.randn(4, 8, 5, 5, device=dev).contiguous(memory_format=CL)
    for _ in range(3):
        net(x)
    stale = [int(b) for n, b in net.named_buffers() if n.endswith("num_batches_tracked")]
    assert stale == [0, 0]                                  # the documented limitation
    flush_counters(net)
    assert [int(b) for n, b in net.named_buffers() if n.endswith("num_batches_tracked")] == [3, 3]
    net(x)
    assert int(net.state_dict()["1.num_batches_tracked"]) == 4


@pytest.mark.parametrize("fold", ["0", "1"])
def test_partial_sum_fold_is_repeatable_and_stream_safe(dev, fold, monkeypatch):
    """The partial sums of bn_stats / bn_backward_reduce are folded either by a small launch (default) or, PIXPRO_BN_FOLD=1,
    by the work-group that draws the last ticket of its channel chunk.  Which work-group that is varies from launch to
    launch; the result must not: 40 forward+backward passes over the same tensors agree bit for bit in both modes, also when
    two streams run different layers at the same time (each stream owns its workspace), and the tickets are left at zero."""
    from contrast import _native as nv
    monkeypatch.setenv("PIXPRO_BN_FOLD", fold)
    from contrast.sync_bn import FusedSyncBatchNorm
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 64, 56, 56), (32, 256, 28, 28), (16, 2048, 7, 7), (3, 12, 5, 7)]
    layers = []
    for shp in shapes:
        bn = FusedSyncBatchNorm(shp[1]).to(dev).train()
        x = torch.randn(*shp, generator=g).to(dev).contiguous(memory_format=CL)
        dy = torch.randn(*shp, generator=g).to(dev).contiguous(memory_format=CL)
        layers.append((bn, x, dy))

    def run(bn, x, dy):
        xf = x.clone().requires_grad_(True)
        y = bn(xf, None, True)
        y.backward(dy)
        return y.detach().clone(), xf.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()

    first = []
    for bn, x, dy in layers:
        first.append(run(bn, x, dy))
        bn.zero_grad()
    side = torch.cuda.Stream(device=dev)
    for it in range(40):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            got_side = [run(*layers[i]) for i in (1, 3)]
        got_main = [run(*layers[i]) for i in (0, 2)]
        torch.cuda.current_stream().wait_stream(side)
        for i, got in zip((1, 3, 0, 2), got_side + got_main):
            for a, b in zip(got, first[i]):
                assert torch.equal(a, b), f"iteration {it}, layer {i}"
            layers[i][0].zero_grad()
    torch.cuda.synchronize()
    for buf in nv._zeroed.values():
        assert int(buf[:4096].view(torch.int32).abs().sum()) == 0          # every ticket back at zero
    assert len(nv._zeroed) >= 2                                           # one workspace per stream


@pytest.mark.parametrize("shape", [(4, 256, 7, 7), (3, 2048, 7, 7), (16, 512, 28, 28), (7, 1024, 5, 3), (2, 256, 1, 1)])
@pytest.mark.parametrize("groups", [1, 2])
def test_relu_bit_record_equals_reading_the_output(dev, shape, groups, monkeypatch):
    """Behind a residual add the ReLU decisions travel to backward as a bit record (pixpro_bn_apply_bits, 1/32 of the tensor)
    instead of being read back from the saved output: every gradient must be BIT-identical to the y-reading kernels, for
    row counts that leave tails in the row loops and with two views per pass."""
    from contrast import sync_bn, _native
    from contrast.sync_bn import FusedSyncBatchNorm, view_groups
    N, C, H, W = shape
    N *= groups
    assert _native.load().pixpro_bn_relu_bits_bytes(N // groups * H * W, C) == N // groups * H * W * C // 8
    assert _native.load().pixpro_bn_relu_bits_bytes(100, 64) == 0          # no record for this shape: callers keep y
    g = torch.Generator().manual_seed(sum(shape) + groups)
    x = (torch.randn(N, C, H, W, generator=g) * 1.3 - 0.2).to(dev).contiguous(memory_format=CL)
    res = torch.randn(N, C, H, W, generator=g).to(dev).contiguous(memory_format=CL)
    dy = torch.randn(N, C, H, W, generator=g).to(dev).contiguous(memory_format=CL)
    out = {}
    real = _native.call
    for bits in (True, False):
        monkeypatch.setattr(sync_bn, "_RELU_BITS", bits)
        calls = []
        monkeypatch.setattr(_native, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
        bn = FusedSyncBatchNorm(C).to(dev).train()
        xf, rf = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        with view_groups(groups):
            y = bn(xf, rf, True)
        y.backward(dy)
        monkeypatch.setattr(_native, "call", real)
        used = any(n.endswith("_bits") for n in calls)
        assert used == bits, calls
        out[bits] = (y.detach(), xf.grad, rf.grad, bn.weight.grad, bn.bias.grad, bn.running_var.clone())
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)
    # and the record is not written when nothing upstream wants a gradient (the key encoder's passes)
    monkeypatch.setattr(sync_bn, "_RELU_BITS", True)
    calls = []
    monkeypatch.setattr(_native, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    with torch.no_grad():
        FusedSyncBatchNorm(C).to(dev).train()(x, res, True)
    assert "pixpro_bn_apply_bits" not in calls and "pixpro_bn_apply" in calls


@pytest.mark.parametrize("groups", [1, 2])
def test_shortcut_link_bitwise_on_deterministic_units(dev, groups, monkeypatch):
    """ResidualLink: the last BatchNorm of a unit hands (dy, ReLU bits) to `fork` / to the shortcut's BatchNorm instead of
    writing the masked gradient for autograd to add.  Three residual units whose 'convolutions' are deterministic
    element-wise scalings (identity shortcut, shortcut with its own BatchNorm, identity again): with and without the link
    every output and gradient is the same bit for bit; a retained graph hands the gradient over again."""
    from contrast import sync_bn
    from contrast.sync_bn import FusedSyncBatchNorm, bn_act, fork, residual_link, view_groups
    C = 256
    g = torch.Generator().manual_seed(21 + groups)
    bns = [FusedSyncBatchNorm(C).to(dev).train() for _ in range(7)]
    ws = [(torch.rand(1, C, 1, 1, generator=g) + 0.5).to(dev).requires_grad_(True) for _ in range(7)]
    with torch.no_grad():
        for m in bns:
            m.weight.copy_(torch.rand(C, generator=g) + 0.5)
            m.bias.copy_(torch.randn(C, generator=g) * 0.3)

    def unit(x, i, down):
        link = residual_link(x, bns[i + 1])
        if down is None:
            x, identity = fork(x, link) if link is not None else (x, x)
        else:
            identity = bn_act(bns[down], x * ws[down], link_in=link)
        out = bn_act(bns[i], x * ws[i], relu=True)
        return bn_act(bns[i + 1], out * ws[i + 1], relu=True, residual=identity, link_out=link)

    def net(x):
        x = unit(x, 0, None)
        x = unit(x, 2, 6)
        return unit(x, 4, None)

    x = torch.randn(3 * groups, C, 5, 7, generator=g).to(dev).contiguous(memory_format=CL)
    dy = torch.randn(3 * groups, C, 5, 7, generator=g).to(dev).contiguous(memory_format=CL)
    out = {}
    for lazy in (True, False):
        monkeypatch.setattr(sync_bn, "_LAZY_SHORTCUT", lazy)
        for t in ws + [p for m in bns for p in m.parameters()]:
            t.grad = None
        xi = x.clone().requires_grad_(True)
        with view_groups(groups):
            y = net(xi)
        y.backward(dy)
        out[lazy] = ([y.detach(), xi.grad] + [t.grad.clone() for t in ws]
                     + [p.grad.clone() for m in bns for p in m.parameters()])
        if lazy:
            xi2 = x.clone().requires_grad_(True)
            with view_groups(groups):
                y2 = net(xi2)
            y2.backward(dy, retain_graph=True)
            g1 = xi2.grad.clone()
            xi2.grad = None
            y2.backward(dy)
            assert torch.equal(g1, xi2.grad)
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("arch,groups", [("resnet50", 1), ("resnet50", 2), ("resnet18", 1)])
def test_shortcut_gradient_over_the_link_in_the_encoders(dev, arch, groups, monkeypatch):
    """The same through the real encoders (contrast/resnet.py): nothing is written for the shortcut (no dresidual pointer),
    identity shortcuts get one masked add each.  MIOpen's split-K convolutions accumulate with atomics - two passes of the
    same encoder differ in the last bits and ~50 BatchNorms amplify that - so this is a plumbing check against the
    run-to-run noise of the plain route itself (a lost or doubled shortcut gradient is an O(1) error); the bit-for-bit
    statement is test_shortcut_link_bitwise_on_deterministic_units."""
    from contrast import resnet, sync_bn, _native
    from contrast.sync_bn import FusedSyncBatchNorm, view_groups
    torch.manual_seed(11)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)      # (asks MIOpen for deterministic solvers where it has them)
    net = FusedSyncBatchNorm.convert(getattr(resnet, arch)(head_type="early_return")).to(dev).to(memory_format=CL).train()
    with torch.no_grad():      # the zero-initialised last gammas would hide the residual branch's BatchNorm gradients
        for m in net.modules():
            if isinstance(m, resnet._Residual):
                (m.bn2 if m.kind == "basic" else m.bn3).weight.fill_(0.25)
    x = torch.randn(4 * groups, 3, 128, 128, device=dev).contiguous(memory_format=CL)
    dy = None
    out = {}
    real = _native.call
    for key, lazy in (("lazy", True), ("plain", False), ("plain2", False)):
        monkeypatch.setattr(sync_bn, "_LAZY_SHORTCUT", lazy)
        calls = []
        monkeypatch.setattr(_native, "call", lambda name, *a: (calls.append((name, a)), real(name, *a))[1])
        net.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        with view_groups(groups):
            y = net(xi)
        if dy is None:
            dy = torch.randn_like(y)
        y.backward(dy)
        monkeypatch.setattr(_native, "call", real)
        out[key] = [y.detach(), xi.grad] + [p.grad for p in net.parameters()]
        adds = [a for n, a in calls if n == "pixpro_bn_masked_add"]
        dres = [a[13] for n, a in calls if n == "pixpro_bn_backward_apply_bits"]
        assert len(dres) > 0
        if lazy:
            assert all(d is None for d in dres)                 # nothing written for the shortcut
            assert 0 < len(adds) <= len(dres)                   # identity shortcuts: one masked add per unit and view
        else:
            assert not adds and all(d is not None for d in dres)

    def rel_err(a, b):
        return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    names = ["y", "dx"] + [n for n, _ in net.named_parameters()]
    worst = []
    for n, l, p1, p2 in zip(names, out["lazy"], out["plain"], out["plain2"]):
        noise, got = rel_err(p2, p1), rel_err(l, p1)
        worst.append((got - 4 * noise, n, got, noise))
        assert got <= 4 * noise + 1e-5, (n, got, noise)
    # the comparison must say something: on the big tensors the noise is small
    assert rel_err(out["plain2"][0], out["plain"][0]) < 1e-2 and rel_err(out["plain2"][1], out["plain"][1]) < 5e-2, sorted(worst)[-3:]


@pytest.mark.parametrize("shape,groups", [((4, 64, 112, 112), 1), ((3, 64, 30, 42), 1), ((2, 32, 17, 9), 2), ((2, 128, 16, 16), 1),
                                          ((1, 64, 2, 3), 1)])
@pytest.mark.parametrize("ties", [False, True])
def test_stem_bn_relu_maxpool_in_one_pass(dev, shape, groups, ties, monkeypatch):
    """bn_relu_pool: BatchNorm + ReLU + 3x3/2/1 max pooling without the full-resolution tensor in between, against the same
    FusedSyncBatchNorm followed by ATen's pooling.  The pooled output is bit-identical; the input gradient and the
    BatchNorm gradients agree to the re-association of the channel sums (the gathered pooled gradient itself is summed in
    ATen's order).  `ties`: inputs on a coarse grid, so that windows hold equal maxima and the first-maximum rule decides."""
    from contrast import sync_bn, _native
    from contrast.sync_bn import FusedSyncBatchNorm, bn_relu_pool, view_groups
    N, C, H, W = shape
    N *= groups
    g = torch.Generator().manual_seed(sum(shape) + groups + 7 * ties)
    x = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.2
    if ties:
        x = torch.round(x * 2) / 2
    x = x.to(dev).contiguous(memory_format=CL)
    pool = nn.MaxPool2d(3, stride=2, padding=1)
    assert _native.load().pixpro_bn_pool_supported(N // groups, H, W, C) == 1
    out = {}
    dy = None
    real = _native.call
    for fused in (True, False):
        monkeypatch.setattr(sync_bn, "_FUSED_POOL", fused)
        calls = []
        monkeypatch.setattr(_native, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
        bn = FusedSyncBatchNorm(C).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, C))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
        xf = x.clone().requires_grad_(True)
        with view_groups(groups):
            y = bn_relu_pool(bn, xf, pool)
        if dy is None:
            dy = torch.randn(y.shape, generator=g).to(dev).contiguous(memory_format=CL)
        y.backward(dy)
        monkeypatch.setattr(_native, "call", real)
        assert ("pixpro_bn_apply_pool" in calls) == fused and ("pixpro_bn_pool_backward_apply" in calls) == fused
        out[fused] = (y.detach(), xf.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone(),
                      int(bn.num_batches_tracked))
    assert out[True][0].shape == (N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1)
    assert torch.equal(out[True][0], out[False][0])
    close(out[True][1], out[False][1], rel=2e-5)
    close(out[True][2], out[False][2], rel=2e-4)
    close(out[True][3], out[False][3], rel=2e-4)
    assert torch.equal(out[True][4], out[False][4]) and torch.equal(out[True][5], out[False][5]) and out[True][6] == out[False][6]
    # the key encoder's passes (no gradient) take the fused forward as well
    monkeypatch.setattr(sync_bn, "_FUSED_POOL", True)
    with torch.no_grad(), view_groups(groups):
        y2 = bn_relu_pool(FusedSyncBatchNorm(C).to(dev).train(), x, pool)
    assert y2.shape == out[True][0].shape


@pytest.mark.parametrize("shape,relu,with_res", [((6, 256, 14, 14), True, True), ((5, 64, 28, 28), True, False),
                                                 ((4, 512, 7, 7), False, False), ((3, 1024, 5, 9), True, True)])
def test_streaming_load_kernels_are_bit_identical(dev, shape, relu, with_res, monkeypatch):
    """bn_stream_policy picks the non-temporal-load instantiations only for tensors of tens of MB, which no other test
    reaches: PIXPRO_BN_STREAM=1 forces them here (0 forces the plain ones).  A load is a load: forward, every gradient and
    the running statistics must be the same bits, through the bit record and the shortcut link as well."""
    from contrast.sync_bn import FusedSyncBatchNorm, bn_act, fork, residual_link
    N, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 1.4 + 0.1).to(dev).contiguous(memory_format=CL)
    dy = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=CL)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PIXPRO_BN_STREAM", mode)
        bn_a, bn_b = FusedSyncBatchNorm(C).to(dev).train(), FusedSyncBatchNorm(C).to(dev).train()
        xf = x.clone().requires_grad_(True)
        if with_res:                       # a unit: fork -> bn_a + relu -> bn_b + shortcut + relu, gradient over the link
            link = residual_link(xf, bn_b)
            xa, identity = fork(xf, link) if link is not None else (xf, xf)
            y = bn_act(bn_b, bn_act(bn_a, xa * 1.5, relu=True), relu=True, residual=identity, link_out=link)
        else:
            y = bn_act(bn_b, bn_act(bn_a, xf * 1.5, relu=relu), relu=relu)
        y.backward(dy)
        out[mode] = [y.detach(), xf.grad, bn_a.weight.grad, bn_a.bias.grad, bn_b.weight.grad, bn_b.bias.grad,
                     bn_a.running_var.clone(), bn_b.running_mean.clone()]
    for a, b in zip(out["1"], out["0"]):
        assert torch.equal(a, b)


def _rank_pool(rank, world, port, path):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from contrast.sync_bn import FusedSyncBatchNorm, bn_relu_pool
    dev = torch.device("cuda:0")
    d = torch.load(path)
    sl = slice(0, 5) if rank == 0 else slice(5, d["x"].shape[0])      # unequal shards: 5 and 3 samples
    bn = FusedSyncBatchNorm(d["x"].shape[1]).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(d["w"]); bn.bias.copy_(d["b"])
    x = d["x"][sl].to(dev).contiguous(memory_format=CL).requires_grad_(True)
    y = bn_relu_pool(bn, x, nn.MaxPool2d(3, stride=2, padding=1))
    y.backward(d["dy"][sl].to(dev).contiguous(memory_format=CL))
    torch.save(dict(y=y.detach().cpu(), dx=x.grad.cpu(), dw=bn.weight.grad.cpu(), db=bn.bias.grad.cpu(),
                    rm=bn.running_mean.cpu(), rv=bn.running_var.cpu()), f"{path}.out{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_fused_stem_over_two_ranks_equals_one_process_on_the_whole_batch(dev, tmp_path):
    """The fused BatchNorm + ReLU + max pooling of the stem under the SyncBN contract: two ranks with unequal shards give
    the pooled output, the input gradient and the running statistics of one process on the whole batch; parameter
    gradients are the local sums."""
    from contrast.sync_bn import FusedSyncBatchNorm, bn_relu_pool
    g = torch.Generator().manual_seed(23)
    shape, C = (8, 64, 20, 18), 64
    d = dict(x=torch.randn(shape, generator=g) * 1.5 + 0.3, dy=torch.randn(8, C, 10, 9, generator=g),
             w=torch.rand(C, generator=g) + 0.5, b=torch.randn(C, generator=g) * 0.3)
    d["x"][5:] += 0.6
    path = str(tmp_path / "pool.pt")
    torch.save(d, path)
    mp.spawn(_rank_pool, args=(2, _free_port(), path), nprocs=2, join=True)
    outs = [torch.load(f"{path}.out{r}") for r in range(2)]
    bn = FusedSyncBatchNorm(C).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(d["w"]); bn.bias.copy_(d["b"])
    x = d["x"].to(dev).contiguous(memory_format=CL).requires_grad_(True)
    y = bn_relu_pool(bn, x, nn.MaxPool2d(3, stride=2, padding=1))
    y.backward(d["dy"].to(dev).contiguous(memory_format=CL))
    got_y = torch.cat([o["y"] for o in outs]).to(dev)
    assert (got_y - y).abs().max().item() <= 2e-5
    # (the statistics are the global ones on both sides; what differs is the order of their sums)
    close(torch.cat([o["dx"] for o in outs]).to(dev), x.grad, rel=2e-3)
    close(outs[0]["dw"] + outs[1]["dw"], bn.weight.grad, rel=1e-3)
    close(outs[0]["db"] + outs[1]["db"], bn.bias.grad, rel=1e-3)
    for o in outs:
        assert torch.allclose(o["rm"].to(dev), bn.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(o["rv"].to(dev), bn.running_var, rtol=1e-5, atol=1e-6)
