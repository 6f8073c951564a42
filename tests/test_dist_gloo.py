"""N>1 path on CPU: two gloo ranks run the DDP pre-training step of main_pretrain with the HIP ops replaced
*by the test* with the CPU oracle (the product has no CPU route of its own).  Checks the data-parallel
contract of SURVEY.md §8(e): samples are sharded across ranks with no collective inside the hot path, DDP
averages the gradients, the ranks stay in lock-step, and bench.py's max-over-ranks timing reduction works."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_ppm_loss(dirs, p, clamp, pos_ratio, normalize_k=True, want_pred=False, generic=False):
    from oracle import pixpro_oracle as O
    total, nums, means = 0, [], []
    for d in dirs:
        a = [d.coord_q, [d.flow, d.size, d.mask]] if d.flow is not None else d.coord_q
        b = [d.coord_k, None] if d.flow is not None else d.coord_k
        loss, _, pn, pm = O.ppm_loss(d.x, d.v, d.k, a, b, p, clamp, pos_ratio, normalize_k)
        total = total + loss
        nums.append(pn)
        means.append(pm)
    return total, torch.stack(nums), torch.stack(means), None


def _oracle_ema(params_q, params_k, m):
    for pq, pk in zip(params_q, params_k):
        pk.data.mul_(m).add_(pq.data, alpha=1. - m)


def _worker(rank, world, port, out):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import main_pretrain
    from contrast import ops, option
    from contrast.lr_scheduler import get_scheduler
    import importlib
    pixpro_mod = importlib.import_module("contrast.models.PixPro")   # the module, not the re-exported class
    from contrast.synthetic import make_batch
    ops.ppm_loss = _oracle_ppm_loss                    # test-side substitution of the HIP launches
    pixpro_mod.ema_update = _oracle_ema
    args = option.parse_option(argv=["--model", "PixPro", "--arch", "resnet18", "--synthetic", "--use_flow",
                                     "--use_flow_file", "--optimizer", "sgd", "--base-lr", "0.1", "--amp-opt-level", "O0",
                                     "--pixpro-p", "2", "--pixpro-transform-layer", "1", "--batch-size", "2",
                                     "--epochs", "2", "--warmup-epoch", "0", "--head-type", "early_return", "--no-sync-bn"])
    args.num_instances = 16
    torch.manual_seed(0)                               # same init on both ranks (DDP would broadcast anyway)
    model, opt = main_pretrain.build_model(args, torch.device("cpu"))
    assert isinstance(model, torch.nn.parallel.DistributedDataParallel)
    sched = get_scheduler(opt, 4, args)
    model.train()
    # each rank draws its own shard (seed 1234 + rank, as bench.py / SURVEY.md §8d)
    data = make_batch(2, image_size=32, n_frames=2, frame_hw=(48, 64), device="cpu", seed=1234 + rank, flow_up=False)
    flow_f, flow_b = data[5][1][:, 0], data[5][2][:, 0]
    size = (48, 64)
    coord1 = [data[2], [flow_f, size, None]]
    coord2 = [data[3], [flow_b, size, None]]
    loss, pos = model(data[0], data[1], coord1, coord2)
    opt.zero_grad()
    loss.backward()
    g = model.module.value_transform.weight.grad.clone()
    opt.step(); sched.step()
    w = model.module.value_transform.weight.detach().clone()
    both = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(both, g)
    wboth = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(wboth, w)
    shard = [torch.zeros_like(data[2]) for _ in range(world)]
    dist.all_gather(shard, data[2])
    # bench.py's timing reduction: max over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        torch.save({"same_grad": torch.equal(both[0], both[1]), "same_w": torch.equal(wboth[0], wboth[1]),
                    "distinct_shards": not torch.equal(shard[0], shard[1]), "tmax": t.item(),
                    "loss_finite": bool(torch.isfinite(loss)), "gnorm": g.norm().item(),
                    "pos_shape": tuple(pos[0][0].shape)}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_ddp_step(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["same_grad"] and r["same_w"], "DDP must leave both ranks with identical averaged gradients / weights"
    assert r["distinct_shards"] and r["tmax"] == 2.0 and r["loss_finite"] and r["gnorm"] > 0
    assert r["pos_shape"] == (2,)


def _worker_groups(rank, world, port, out):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from contrast import resnet
    from contrast.models import PixPro
    from contrast.sync_bn import FusedSyncBatchNorm
    res = {}
    for fused in (False, True):
        args = types.SimpleNamespace(pixpro_p=2.0, pixpro_momentum=0.99, pixpro_pos_ratio=0.7, pixpro_clamp_value=0.0,
                                     pixpro_transform_layer=1, pixpro_ins_loss_weight=0.0, output_dir="/tmp", num_instances=16,
                                     batch_size=2, epochs=2, start_epoch=1, feature_dim=256, head_type="early_return",
                                     sync_bn=True, fused_bn=fused, key_bn_group=True)
        model = PixPro(resnet.__dict__["resnet18"], args)
        kind = FusedSyncBatchNorm if fused else torch.nn.SyncBatchNorm
        key = [m for part in (model.encoder_k, model.projector_k) for m in part.modules() if isinstance(m, kind)]
        onl = [m for part in (model.encoder, model.projector) for m in part.modules() if isinstance(m, kind)]
        g = model.key_bn_group
        t = torch.tensor([1.0 + rank])
        dist.all_reduce(t, group=g)                       # the second group is a working communicator over the same ranks
        res[fused] = dict(n_key=len(key), n_onl=len(onl), key_ok=all(m.process_group is g for m in key),
                          onl_ok=all(m.process_group is None for m in onl), distinct=g is not None and g is not dist.group.WORLD,
                          size=dist.get_world_size(g), sum=t.item())
    if rank == 0:
        torch.save(res, out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_key_branch_syncbn_gets_its_own_process_group(tmp_path):
    """--key-bn-group: the BatchNorm layers of the key (momentum) encoder and projector exchange their statistics over a
    second process group (own RCCL communicator and stream), the online branch's over the default one - plumbing only:
    SyncBN itself needs the GPU (tests/test_gpu_sync_bn.py runs the fused module over explicit groups)."""
    out = str(tmp_path / "groups.pt")
    mp.spawn(_worker_groups, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    for fused in (False, True):
        q = r[fused]
        assert q["n_key"] > 10 and q["n_key"] == q["n_onl"]
        assert q["key_ok"] and q["onl_ok"] and q["distinct"] and q["size"] == 2 and q["sum"] == 3.0
