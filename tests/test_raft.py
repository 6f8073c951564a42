"""RAFT inference (contrast/flow): oracle and host model against golden vectors captured from the reference's
contrast.flow package (tests/golden/make_golden_raft.py); HIP correlation lookup and GPU inference against them."""
import argparse
import importlib

import pytest
import torch

from conftest import load_golden

CORR = ["l3r2", "l4r3"]


def _model(tag, g):
    from contrast.flow import RAFT
    from recipes import raft_state_dict, tag_seed
    model = RAFT(argparse.Namespace(small=(tag == "small"))).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    return model, shapes, raft_state_dict(shapes, tag_seed("raft" + tag))


@pytest.mark.parametrize("tag", CORR)
def test_oracle_lookup_matches_reference(tag):
    from oracle import raft_oracle as R
    g = load_golden(f"raft_corr_{tag}")
    pyr = R.corr_pyramid(g["fmap1"], g["fmap2"], int(g["levels"]))
    assert torch.equal(R.corr_lookup(pyr, g["coords"], int(g["radius"])), g["out"])


@pytest.mark.parametrize("tag", ["small", "basic"])
def test_state_dict_names_and_host_inference_match_reference(tag, monkeypatch):
    """Same parameter names / shapes as the reference's RAFT (its checkpoints load), and - with the lookup substituted by
    the oracle FROM THE TEST - the same low-resolution flow on the host."""
    from oracle import raft_oracle as R
    from recipes import raft_images, tag_seed
    import numpy as np
    z = np.load(f"{importlib.import_module('conftest').GOLDEN}/raft_{tag}_e2e.npz")
    model, shapes, sd = _model(tag, z)
    assert sorted(shapes) == [str(n) for n in z["names"]]
    assert [str(shapes[k]) for k in sorted(shapes)] == [str(s) for s in z["shapes"]]
    model.load_state_dict(sd)
    hw = [int(v) for v in z["hw"]]
    im1, im2 = raft_images(1, hw[0], hw[1], tag_seed("raftimg" + tag))
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            model(im1, im2, iters=1, test_mode=True)
        monkeypatch.setattr(importlib.import_module("contrast.flow.raft"), "CorrBlock", R.CorrBlockCPU)
        monkeypatch.setattr(importlib.import_module("contrast.flow.raft"), "upflow8", R.upflow8_cpu)
        lo, up = model(im1, im2, iters=12, test_mode=True)
    want = torch.from_numpy(z["flow_lo"])
    assert (lo - want).abs().max().item() <= 1e-5
    assert (up - torch.from_numpy(z["flow_up"]).float()).abs().max().item() <= 2e-2     # stored as float16


def test_checkpoint_with_dataparallel_prefix_loads(tmp_path):
    """Public RAFT checkpoints are saved from nn.DataParallel: `module.`-prefixed keys (main_pretrain.py:48-51)."""
    from contrast.flow import RAFT
    from contrast.util import load_flow_model
    src = RAFT(argparse.Namespace(small=True))
    path = tmp_path / "raft-small.pth"
    torch.save({"module." + k: v for k, v in src.state_dict().items()}, path)
    args = argparse.Namespace(flow_model=str(path))
    model = load_flow_model(args, device="cpu")
    assert args.small is True and not model.training and all(not p.requires_grad for p in model.parameters())
    assert torch.equal(model.fnet.conv1.weight, src.fnet.conv1.weight)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CORR)
def test_hip_lookup_bit_exact(dev, tag):
    from contrast.flow import corr
    from oracle import raft_oracle as R
    g = load_golden(f"raft_corr_{tag}")
    pyr = R.corr_pyramid(g["fmap1"], g["fmap2"], int(g["levels"]))
    out = corr.lookup([p.to(dev) for p in pyr], g["coords"].to(dev), int(g["radius"]))
    assert torch.equal(out.cpu(), g["out"])
    # the device-built pyramid (hipBLASLt GEMM) differs from the host one only by GEMM re-association
    blk = corr.CorrBlock(g["fmap1"].to(dev), g["fmap2"].to(dev), int(g["levels"]), int(g["radius"]))
    assert (blk(g["coords"].to(dev)).cpu() - g["out"]).abs().max().item() <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["small", "basic"])
def test_gpu_inference_matches_reference(dev, tag):
    """fp32 MIOpen convolutions vs the reference's CPU run, 12 refinement iterations: low-res flow within 2e-3 of a
    1/8-resolution pixel (measured ~1e-4), i.e. < 0.02 px at full resolution."""
    import numpy as np
    from recipes import raft_images, tag_seed
    z = np.load(f"{importlib.import_module('conftest').GOLDEN}/raft_{tag}_e2e.npz")
    model, shapes, sd = _model(tag, z)
    model.load_state_dict(sd)
    model = model.to(dev)
    hw = [int(v) for v in z["hw"]]
    im1, im2 = raft_images(1, hw[0], hw[1], tag_seed("raftimg" + tag))
    with torch.no_grad():
        lo, up = model(im1.to(dev), im2.to(dev), iters=12, test_mode=True)
    err = (lo.cpu() - torch.from_numpy(z["flow_lo"])).abs().max().item()
    assert err <= 2e-3, err
    assert (up.cpu() - torch.from_numpy(z["flow_up"]).float()).abs().max().item() <= 4e-2


@pytest.mark.gpu
def test_online_flow_mode_equals_file_mode_on_the_same_flows(dev):
    """apply_optical_flow with a flow model (reference util.py:200-203) = RAFT per frame pair, then the same fused
    preparation as the file mode: feeding RAFT's own low-res flows through the file branch must give identical tensors."""
    import types
    from contrast import util
    from recipes import raft_images, tag_seed
    model, shapes, sd = _model("small", None)
    model.load_state_dict(sd)
    model = model.to(dev)
    N, F, H, W = 3, 3, 128, 160      # 1/8 grid 16x20: the coarsest pyramid level must keep >= 2 pixels (W-1 = 0 otherwise)
    frames = []
    for t in range(F):
        a, _ = raft_images(N, H, W, tag_seed(f"online{t}"))
        frames.append(a.to(dev))
    size = torch.tensor([[H, W]] * N)
    data = [None] * 5 + [None, [size, torch.full((N, 1), F)] + frames]
    args = types.SimpleNamespace(use_flow_file=False, use_flow_frames=False, flow_cat_norm=False, flow_up=True, alpha1=0.01,
                                 alpha2=0.5, debug=False, flow_bs=2, verbose=False)
    (ff, _, mf), (fb, _, mb) = util.apply_optical_flow(data, model, args)
    assert ff.shape == (N, 2, H, W) and mf.shape == (N, H, W)
    with torch.no_grad():     # same chunks of flow_bs samples as the online branch (MIOpen may pick per-batch-size kernels)
        lo_f = torch.cat([torch.stack([model(frames[t][s], frames[t + 1][s], test_mode=True)[0] for t in range(F - 1)], 1)
                          for s in (slice(0, 2), slice(2, 3))])
        lo_b = torch.cat([torch.stack([model(frames[t + 1][s], frames[t][s], test_mode=True)[0]
                                       for t in reversed(range(F - 1))], 1) for s in (slice(0, 2), slice(2, 3))])
    args.use_flow_file = True
    data2 = [None] * 5 + [[None, lo_f, lo_b], [size, torch.full((N, 1), F)]]
    (gf, _, gmf), (gb, _, gmb) = util.apply_optical_flow(data2, None, args)
    mf_, mb_ = util.mem_reduce_calc_optical_flow(frames, model, args)       # public helper of the reference API
    assert mf_.shape == (1, N, 2, H, W) and torch.allclose(mf_[0], ff, rtol=1e-4, atol=1e-3)
    assert torch.allclose(mb_[0], fb, rtol=1e-4, atol=1e-3)
    # two evaluations of the same network are equal up to MIOpen's run-to-run re-association (~1e-5 relative)
    assert torch.allclose(ff, gf, rtol=1e-4, atol=1e-3) and torch.allclose(fb, gb, rtol=1e-4, atol=1e-3)
    assert (mf != gmf).float().mean().item() < 1e-3


@pytest.mark.gpu
def test_train_step_online_flow_mode(dev):
    """main_pretrain with --use_flow and no --use_flow_file: RAFT-small (random weights, synthetic frames) feeds the same
    hot path; dense and lazy preparation agree."""
    import main_pretrain
    from contrast import option
    from contrast.lr_scheduler import get_scheduler
    from contrast.synthetic import make_batch
    base = ["--model", "PixPro", "--arch", "resnet18", "--synthetic", "--use_flow", "--flow_up", "--n-frames", "3", "--optimizer", "sgd", "--base-lr", "0.1", "--amp-opt-level", "O0",
            "--pixpro-p", "2", "--pixpro-transform-layer", "1", "--batch-size", "4", "--epochs", "2", "--warmup-epoch", "0",
            "--head-type", "early_return", "--flow_bs", "2"]
    from recipes import raft_state_dict
    data = make_batch(4, image_size=64, n_frames=3, frame_hw=(128, 160), device=dev, with_frames=True)
    assert len(data[6]) == 2 + 3 and data[6][2].shape == (4, 3, 128, 160)
    out = []
    for extra in ([], ["--lazy-flow"]):
        args = option.parse_option(argv=base + extra)
        args.num_instances = 16
        torch.manual_seed(0)
        model, opt = main_pretrain.build_model(args, dev)
        flow_model = main_pretrain.build_flow_model(args, dev)
        assert flow_model is not None and flow_model.update_block.mask is None
        # a random network with a damped flow head: flows of a few pixels instead of noise
        flow_model.load_state_dict({k: v.to(dev) for k, v in raft_state_dict(
            {k: tuple(v.shape) for k, v in flow_model.state_dict().items()}, 5).items()})
        model.train()
        loss, pos, info = main_pretrain.train_step(model, opt, get_scheduler(opt, 4, args), data, args, flow_model)
        out.append((loss.item(), pos[0][0].clone()))
    assert out[0][0] == out[0][0] and abs(out[0][0] - out[1][0]) <= 1e-5 * max(1.0, abs(out[0][0]))
    assert torch.equal(out[0][1], out[1][1]) and out[0][1].sum() > 0
