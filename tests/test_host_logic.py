"""Host-side logic that needs no GPU: flags, argument polymorphism, schedules, synthetic records,
state-dict compatibility, and the 'fail loudly on CPU' contract."""
import json
import math
import os
import random
import types

import pytest
import torch

from conftest import GOLDEN


def test_flags_match_reference_recipe():
    from contrast import option
    # tools/pretrain_bdd100k_job_base.sh:251-281 flag set (values of the nframe-6 recipe)
    argv = ["--data-dir", "/d", "--output-dir", "/o", "--dataset", "bdd100k", "--aug", "BYOL", "--crop", "0.08",
            "--image-size", "224", "224", "--n-frames", "6", "--use_flow", "--use_flow_file", "--flow_up",
            "--alpha1", "0.01", "--alpha2", "0.5", "--flow_root", "/f", "--fwd_name", "a", "--bwd_name", "b",
            "--batch-size", "64", "--arch", "resnet50", "--model", "PixPro", "--head-type", "early_return",
            "--optimizer", "lars", "--base-lr", "1.0", "--weight-decay", "1e-5", "--warmup-epoch", "5",
            "--epochs", "2000", "--amp-opt-level", "O0", "--save-freq", "10", "--auto-resume", "--zip",
            "--cache-mode", "no", "--num-workers", "8", "--local_rank", "3",
            "--pixpro-p", "2", "--pixpro-momentum", "0.99", "--pixpro-pos-ratio", "0.7", "--pixpro-clamp-value", "0.",
            "--pixpro-transform-layer", "1", "--pixpro-ins-loss-weight", "0."]
    a = option.parse_option(argv=argv)
    assert a.image_size == 224 and a.local_rank == 3 and a.debug_epochs == 2001
    assert a.base_learning_rate == 1.0 and a.pixpro_p == 2.0 and a.use_flow and a.flow_up
    with pytest.raises(SystemExit):
        option.parse_option(argv=[])          # --model is required, as in the reference


def test_unpack_polymorphism():
    from contrast.models.PixPro import _unpack
    cq, ck = torch.rand(2, 10), torch.rand(2, 10)
    flow, mask = torch.zeros(2, 2, 4, 4), torch.ones(2, 4, 4, dtype=torch.bool)
    assert _unpack(cq, ck)[2:] == (None, None, None)
    out = _unpack([cq, flow], [ck, flow])
    assert out[2] is flow and out[3] is None and out[4] is None
    out = _unpack([cq, [flow, (4, 4), mask]], [ck, [flow, (4, 4), None]])
    assert out[0] is cq and out[1] is ck and out[3] == (4, 4) and out[4] is mask
    out = _unpack([cq, [flow, torch.tensor([4, 4]), [mask, None]]], [ck, [flow, None, None]])
    assert out[4] is mask
    with pytest.raises(NotImplementedError):
        _unpack((cq, None), (ck, None))


def test_cpu_tensors_are_rejected_not_emulated():
    from contrast import util
    from contrast.models.PixPro import add_optical_flow, regression_loss
    q = torch.randn(2, 8, 2, 2)
    c = torch.rand(2, 10)
    for fn in (lambda: regression_loss(q, q, c, c, 0.7),
               lambda: util.concat_flow(torch.zeros(2, 1, 2, 4, 4)),
               lambda: util.upflow8(torch.zeros(1, 2, 4, 4)),
               lambda: util.forward_backward_consistency(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4)),
               lambda: add_optical_flow(torch.zeros(2, 2, 4, 4), torch.zeros(2, 2, 2), torch.zeros(2, 2, 2), (4, 4))):
        with pytest.raises(RuntimeError, match="GPU only|CUDA"):
            fn()


def test_lr_schedule_matches_reference_formula():
    from contrast.lr_scheduler import get_scheduler
    args = types.SimpleNamespace(lr_scheduler="cosine", epochs=10, warmup_epoch=2, warmup_multiplier=100,
                                 lr_decay_epochs=[4, 8], lr_decay_rate=0.1)
    n_iter = 5
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=4.0)
    sched = get_scheduler(opt, n_iter, args)
    got = []
    for _ in range(50):
        opt.step(); sched.step(); got.append(opt.param_groups[0]["lr"])
    warm = 2 * n_iter
    for step, lr in enumerate(got, start=1):
        if step <= warm:   # GradualWarmupScheduler.get_lr, contrast/lr_scheduler.py:25-30
            want = 4.0 / 100 * ((100 - 1.) * step / warm + 1.)
        else:              # CosineAnnealingLR closed form with eta_min 1e-6, T_max (epochs-warmup)*n_iter
            t = step - warm
            want = 1e-6 + (4.0 - 1e-6) * (1 + math.cos(math.pi * t / ((10 - 2) * n_iter))) / 2
        assert abs(lr - want) < 1e-9, (step, lr, want)
    args.lr_scheduler = "step"
    opt = torch.optim.SGD([p], lr=4.0)
    sched = get_scheduler(opt, n_iter, args)
    lrs = []
    for _ in range(45):
        opt.step(); sched.step(); lrs.append(opt.param_groups[0]["lr"])
    assert abs(lrs[warm + (4 - 2) * n_iter - 2] - 4.0) < 1e-9 and abs(lrs[warm + (4 - 2) * n_iter - 1] - 0.4) < 1e-9
    assert abs(lrs[-1] - 0.04) < 1e-9


def test_synthetic_batch_layout():
    from contrast.synthetic import crop_coords, make_batch
    d = make_batch(3, image_size=16, n_frames=6, frame_hw=(72, 128), device="cpu")
    im1, im2, c1, c2, idx, (tgt, fw, bw), (size, num) = d
    assert im1.shape == (3, 3, 16, 16) and c1.shape == (3, 10) and fw.shape == (3, 5, 2, 9, 16)
    assert torch.equal(bw, -fw.flip(1)) and size.tolist() == [[72, 128]] * 3 and num.tolist() == [[6]] * 3
    c = crop_coords(200, 1280, 720, random.Random(0))
    assert torch.all(c[:, 8] == 1280) and torch.all(c[:, 9] == 720)
    flipped = c[:, 0] > c[:, 2]
    assert 0.3 < flipped.float().mean() < 0.7                           # p=0.5 flips encoded as x0 > x2
    lo, hi = torch.minimum(c[:, 0], c[:, 2]), torch.maximum(c[:, 0], c[:, 2])
    assert torch.allclose(lo * 1279, c[:, 4], atol=1e-3) and torch.allclose(hi * 1279, c[:, 4] + c[:, 6] - 1, atol=1e-3)
    area = c[:, 6] * c[:, 7] / (1280 * 720)
    assert area.min() >= 0.07 and area.max() <= 1.0


def test_calc_frame_ratio_contract():
    from contrast.util import calc_frame_ratio
    mean_f, no_r, per = calc_frame_ratio(torch.tensor([[2], [2], [1], [6]]))
    assert mean_f == pytest.approx(2.75) and no_r == pytest.approx(0.25)
    assert per.shape == (6, 2) and per[1].tolist() == [0.5, 2.0] and per[5].tolist() == [0.25, 1.0]


def test_state_dict_names_match_reference():
    from contrast import resnet
    from contrast.models import PixPro
    with open(os.path.join(GOLDEN, "model_cfg1_shapes.json")) as fh:
        ref = {k: tuple(v) for k, v in json.load(fh).items()}
    args = types.SimpleNamespace(pixpro_p=2.0, pixpro_momentum=0.99, pixpro_pos_ratio=0.7, pixpro_clamp_value=0.0,
                                 pixpro_transform_layer=1, pixpro_ins_loss_weight=0.0, output_dir="/tmp",
                                 num_instances=800, batch_size=8, epochs=10, start_epoch=1, feature_dim=256,
                                 head_type="early_return", sync_bn=False)
    m = PixPro(resnet.resnet18, args)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == ref
    assert m.K == 1000 and m.k == 0
    assert all(not p.requires_grad for p in m.encoder_k.parameters())
    # ResNet-50: 656 state tensors, 167 trainable tensors, 33,023,552 trainable elements
    m50 = PixPro(resnet.resnet50, args)
    assert len(m50.state_dict()) == 656
    tr = [p for p in m50.parameters() if p.requires_grad]
    assert len(tr) == 167 and sum(p.numel() for p in tr) == 33023552
    assert m50.momentum_coefficient() == pytest.approx(0.99)


def test_lars_grouping():
    from contrast.lars import add_weight_decay
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
    g = add_weight_decay(net, 1e-5)
    assert [len(x["params"]) for x in g] == [3, 1] and g[0]["ignore"] and not g[1]["ignore"]
    assert g[0]["weight_decay"] == 0 and g[1]["weight_decay"] == 1e-5


def test_fused_bn_conversion_keeps_names_and_has_no_cpu_route():
    import torch.nn as nn
    from contrast import resnet
    from contrast.sync_bn import FusedSyncBatchNorm, bn_act
    plain = resnet.resnet18(head_type="early_return")
    names = list(plain.state_dict().keys())
    w = plain.bn1.weight
    fused = FusedSyncBatchNorm.convert(plain)
    assert list(fused.state_dict().keys()) == names and fused.bn1.weight is w
    assert isinstance(fused.layer1[0].bn2, FusedSyncBatchNorm) and isinstance(fused.layer2[0].downsample[1], FusedSyncBatchNorm)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fused.train()(torch.randn(2, 3, 32, 32))
    # eval mode needs no statistics kernel and runs anywhere; stock modules go through the stock ops
    fused.eval()
    assert fused(torch.randn(1, 3, 32, 32)).shape == (1, 512, 1, 1)
    bn = nn.BatchNorm2d(4)
    x, r = torch.randn(2, 4, 3, 3), torch.randn(2, 4, 3, 3)
    assert torch.equal(bn_act(bn, x, relu=True, residual=r), torch.relu(bn(x) + r))


def test_product_and_tools_never_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under the package or tools/ may import it, and the product must fail
    loudly without the HIP library instead of falling back."""
    import re
    from conftest import PKG, ROOT
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|import_module\(.oracle", re.M)
    offenders = []
    for base in (PKG, os.path.join(ROOT, "tools")):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if f.endswith(".py") and pat.search(open(os.path.join(dirpath, f)).read()):
                    offenders.append(os.path.join(dirpath, f))
    assert offenders == []
    from contrast import _native
    saved, _native._lib, _native.LIB_PATH = (_native._lib, _native.LIB_PATH), None, "/nonexistent/libpixpro_hip.so"
    try:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            _native.load()
    finally:
        _native._lib, _native.LIB_PATH = saved


def _tiny_pixpro(**over):
    from contrast import resnet
    from contrast.models import PixPro
    a = dict(pixpro_p=2.0, pixpro_momentum=0.99, pixpro_pos_ratio=0.7, pixpro_clamp_value=0.0, pixpro_transform_layer=1,
             pixpro_ins_loss_weight=0.0, output_dir="/tmp", num_instances=64, batch_size=4, epochs=3, start_epoch=1,
             feature_dim=256, head_type="early_return", sync_bn=False)
    a.update(over)
    return PixPro(resnet.resnet18, types.SimpleNamespace(**a))


def test_momentum_schedule_is_clamped_and_resume_counter_matches_constructor():
    """k must not wrap the cosine once it passes K; a resume computes k with the constructor's expression."""
    m = _tiny_pixpro()
    assert m.K == 48 and m.k == 0 and m.momentum_coefficient() == pytest.approx(0.99)
    m.k = m.K
    assert m.momentum_coefficient() == pytest.approx(1.0)
    m.k = 2 * m.K          # a loader with more steps than num_instances / batch: stays at 1, the cosine does not come back
    assert m.momentum_coefficient() == pytest.approx(1.0)
    # PixPro.py:295 with a fractional steps-per-epoch: int(10.5 * 2) = 21, not 2 * int(10.5)
    m2 = _tiny_pixpro(num_instances=42, start_epoch=3)
    assert m2.k == 21 and m2.steps_before_epoch(3) == 21 and m2.K == 31


def test_checkpoints_are_loaded_with_the_restricted_unpickler(tmp_path):
    """a checkpoint is an untrusted pickle.  Tensors + plain containers + the reference's
    argparse.Namespace load; anything else is refused instead of executed."""
    import argparse
    import main_pretrain

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    good = tmp_path / "good.pth"
    torch.save({"model": {"w": torch.ones(2)}, "epoch": 3, "opt": argparse.Namespace(lr=0.1, arch="resnet50")}, good)
    ck = main_pretrain._safe_load(str(good))
    assert ck["epoch"] == 3 and ck["opt"].arch == "resnet50" and torch.equal(ck["model"]["w"], torch.ones(2))
    bad = tmp_path / "bad.pth"
    torch.save({"model": {"w": torch.ones(2)}, "epoch": 3, "opt": Evil()}, bad)
    with pytest.raises(Exception, match="(?i)unsupported|weights_only|not an allowed"):
        main_pretrain._safe_load(str(bad))
    assert not (tmp_path / "pwned").exists()
    model = torch.nn.Linear(2, 2)
    with pytest.raises(Exception):
        main_pretrain.load_pretrained(model, str(bad))


def test_resume_accepts_the_reference_scheduler_layout(tmp_path):
    """the reference saves a GradualWarmupScheduler state dict (keys multiplier, total_epoch, after_scheduler,
    last_epoch ...).  Only the step count matters for the closed-form schedule here."""
    import main_pretrain
    from contrast import option
    from contrast.lr_scheduler import get_scheduler
    args = option.parse_option(argv=["--model", "PixPro", "--arch", "resnet18", "--synthetic", "--optimizer", "sgd",
                                     "--base-lr", "0.1", "--batch-size", "4", "--epochs", "4", "--warmup-epoch", "1"])

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(2, 2)
            self.k = -1

        def steps_before_epoch(self, epoch):
            return 10 * (epoch - 1)

    def fresh():
        m = M()
        o = torch.optim.SGD(m.parameters(), lr=0.5, momentum=0.9)
        return m, o, get_scheduler(o, 10, args)

    m, o, s = fresh()
    for _ in range(17):
        o.step()
        s.step()
    lr17 = o.param_groups[0]["lr"]
    ref_sched = {"multiplier": 100, "total_epoch": 10, "after_scheduler": {"T_max": 30, "eta_min": 1e-6, "last_epoch": 7},
                 "finished": True, "last_epoch": 17, "base_lrs": [0.5], "_step_count": 18}
    path = tmp_path / "ref.pth"
    torch.save({"model": m.state_dict(), "optimizer": o.state_dict(), "scheduler": ref_sched, "epoch": 2}, path)
    m2, o2, s2 = fresh()
    args.resume = str(path)
    main_pretrain.load_checkpoint(args, m2, o2, s2)
    assert args.start_epoch == 3 and m2.k == 20 and s2.last_epoch == 17
    assert o2.param_groups[0]["lr"] == pytest.approx(lr17, rel=1e-12)
    o2.step(); s2.step(); o.step(); s.step()
    assert o2.param_groups[0]["lr"] == pytest.approx(o.param_groups[0]["lr"], rel=1e-12)
    # our own layout still round-trips
    torch.save({"model": m.state_dict(), "optimizer": o.state_dict(), "scheduler": s.state_dict(), "epoch": 2}, path)
    m3, o3, s3 = fresh()
    main_pretrain.load_checkpoint(args, m3, o3, s3)
    assert s3.last_epoch == 18 and o3.param_groups[0]["lr"] == pytest.approx(o.param_groups[0]["lr"], rel=1e-12)


def test_shipped_miopen_db_is_copied_not_written(monkeypatch):
    """MIOpen must never append to the git-tracked files; each process gets a private copy."""
    from contrast import miopen_tuning as mt
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("MIOPEN_CUSTOM_CACHE_DIR", raising=False)
    monkeypatch.setattr(mt, "_run_dir", None)
    before = {f: os.path.getmtime(os.path.join(mt.DB_DIR, f)) for f in os.listdir(mt.DB_DIR)}
    assert mt.use_shipped_db() is True
    run_dir = os.environ["MIOPEN_USER_DB_PATH"]
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", run_dir)          # so monkeypatch restores the environment afterwards
    monkeypatch.setenv("MIOPEN_CUSTOM_CACHE_DIR", run_dir)
    assert run_dir != mt.DB_DIR and os.environ["MIOPEN_CUSTOM_CACHE_DIR"] == run_dir
    assert sorted(os.listdir(run_dir)) == sorted(before)
    assert mt.use_shipped_db() is True                          # idempotent
    st = mt.status()
    assert "find-db" in st
    # a MIOpen with another build string opens its own file next to the copy: reported, not silent
    open(os.path.join(run_dir, "gfx950100.HIP.9_9_9_other.ufdb.txt"), "w").close()
    assert mt.status().startswith("library heuristics in use")
    assert {f: os.path.getmtime(os.path.join(mt.DB_DIR, f)) for f in os.listdir(mt.DB_DIR)} == before
    # the user's own choice wins
    monkeypatch.setattr(mt, "_run_dir", None)
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    assert mt.use_shipped_db() is False and os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"


def _bench_module():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_bench_line_helpers():
    """The pieces of the bench line that are pure host logic: algorithmic work per launch, the fractions, the
    attainable bound and the PMC figures (kept summaries, filtered to kernels the built library still contains)."""
    b = _bench_module()
    assert b._mangled_fragment("k_tl_strip<2, 0, 8>") == b"10k_tl_stripILi2ELi0ELi8EE"
    assert b._mangled_fragment("ppm_nhwc_kernel") == b"15ppm_nhwc_kernel"
    a = types.SimpleNamespace(image_size=224, n_frames=2, batch=256, arch="resnet50", overlap_momentum=True)
    ms = {"flow_prepare": 0.8, "ppm_loss": 0.08, "lars_sgd_step": 0.185, "ema_update": 0.076}
    k = b.hot_kernel_lines(a, ms)
    assert k["ppm_loss"]["algorithmic_flops_per_launch"] == 24 * 256 * 49 * 49 * 256
    assert abs(k["ppm_loss"]["frac"] - 24 * 256 * 49 * 49 * 256 / 0.08e-3 / 157.3e12) < 1e-3
    assert k["flow_prepare"]["algorithmic_bytes_per_launch"] == int((720 * 1280 * 4 / 16 + 4.5 * 720 * 1280 * 4) * 256)
    assert all(0 < v["frac"] < 1 for v in k.values())
    att = k["ppm_loss"].get("attainable")
    if att is not None:                              # present when the library is built and a PMC summary is kept
        assert att["kernel"] in ("ppm_nhwc_kernel", "ppm_fused_kernel") and 0 < att["frac_of_attainable"] < 1
    util = k["ppm_loss"]["mfma_util_pmc_percent"]
    if util:                                         # only kernels the library still launches
        live = b._lib_bytes()
        assert all(b._mangled_fragment(name) in live for name in util)
    a6 = types.SimpleNamespace(image_size=224, n_frames=6, batch=64, arch="resnet50", overlap_momentum=True)
    k6 = b.hot_kernel_lines(a6, {"flow_prepare": 2.5, "ppm_loss": 0.045})
    vi = k6["flow_prepare"].get("valu_issue")
    if vi:
        assert vi["bound"] == "valu_issue" and 0.5 < vi["frac"] < 1.0
    a5 = types.SimpleNamespace(image_size=448, n_frames=2, batch=64, arch="resnet50", overlap_momentum=True)
    k5 = b.hot_kernel_lines(a5, {"ppm_loss": 0.296})
    assert k5["ppm_loss"]["algorithmic_flops_per_launch"] == 24 * 256 * 196 * 196 * 64 and "attainable" not in k5["ppm_loss"]


def test_residual_units_and_stem_keep_the_stock_route_without_the_fused_modules():
    """The traffic-saving routes of contrast/sync_bn.py (ReLU bit record, ResidualLink, fused stem) exist for
    FusedSyncBatchNorm on the GPU only: with stock BatchNorm on the CPU `residual_link` hands out no link, `bn_relu_pool`
    is pool(relu(bn(x))), and the encoder's forward / backward are those of the plain ops."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from contrast import resnet
    from contrast.sync_bn import bn_relu_pool, residual_link
    torch.manual_seed(0)
    bn, pool = nn.BatchNorm2d(8).train(), nn.MaxPool2d(3, stride=2, padding=1)
    x = torch.randn(2, 8, 9, 11, requires_grad=True)
    assert residual_link(x, bn) is None
    want = pool(F.relu(nn.BatchNorm2d(8).train()(x)))
    assert torch.equal(bn_relu_pool(bn, x, pool), want)
    net = resnet.resnet18(head_type="early_return").train()
    y = net(torch.randn(2, 3, 64, 64))
    y.sum().backward()
    assert y.shape == (2, 512, 2, 2) and all(p.grad is not None for p in net.parameters())


def test_lars_zero_grad_releases_the_gradients():
    """LARS.zero_grad() sets gradients to None (torch's default since 2.0; contrast/lars.py says why) and still honours
    set_to_none=False."""
    import torch
    from contrast.lars import LARS
    p = torch.nn.Parameter(torch.ones(3, 3))
    opt = LARS([{"params": [p], "weight_decay": 0.0, "ignore": False}], lr=0.1, momentum=0.9)
    p.grad = torch.ones_like(p)
    opt.zero_grad()
    assert p.grad is None
    p.grad = torch.ones_like(p)
    opt.zero_grad(set_to_none=False)
    assert p.grad is not None and float(p.grad.abs().sum()) == 0.0
