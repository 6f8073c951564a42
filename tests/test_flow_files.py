"""On-disk flow format (contrast/flow_files.py) against vectors captured from the reference's loader functions
(tests/golden/make_golden_flowfiles.py) and against its path rules (contrast/data/__init__.py:15-34)."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "flowfiles.npz"))


@pytest.fixture()
def files(gold, tmp_path):
    paths = {}
    for n_video in (2, 3, 6, 9):
        fp, bp = tmp_path / f"f{n_video}.pth", tmp_path / f"b{n_video}.pth"
        torch.save(torch.from_numpy(gold[f"file_fwd_{n_video}"]), fp)
        torch.save(torch.from_numpy(gold[f"file_bwd_{n_video}"]), bp)
        paths[n_video] = (str(fp), str(bp))
    return paths


def test_windows_and_slices_match_reference(gold, files):
    from contrast import flow_files as ff
    for k, (n_video, n_frames, local_i, s, n, bs, bn) in enumerate(gold["cases"].tolist()):
        assert ff.clip_window(local_i, n_video, n_frames) == (s, n, bs, bn)
        assert ff.calc_bwd_idx(s, n, n_video - 1) == (bs, bn)
        fp, bp = files[n_video]
        f, b = ff.load_flows([fp, s, n], [bp, -1, -1])
        assert np.array_equal(f.numpy(), gold[f"out_fwd_{k}"]) and np.array_equal(b.numpy(), gold[f"out_bwd_{k}"])
        part, num = ff.load_flow(fp, s, n)
        assert num == n_video - 1 and torch.equal(part, f)


def test_backward_stack_is_time_reversed(gold):
    """Property of the format (dataset.py:357-361): the backward window of a clip covers the same frame pairs as the
    forward window, in reverse order."""
    from contrast import flow_files as ff
    for n_video in (2, 5, 9):
        for n_frames in (2, 3, 6):
            nf = min(n_frames, n_video)
            for i in range(n_video - nf + 1):
                s, n, bs, bn = ff.clip_window(i, n_video, n_frames)
                fwd_pairs = [(t, t + 1) for t in range(s, n)]
                bwd_pairs = [(n_video - 1 - t, n_video - 2 - t) for t in range(bs, bn)]
                assert [(b, a) for a, b in reversed(fwd_pairs)] == bwd_pairs
    with pytest.raises(IndexError):
        ff.clip_window(4, 6, 3)


def test_path_rules():
    from contrast import flow_files as ff
    a = types.SimpleNamespace(use_flow_file=True, flow_root="", data_dir="/data/bdd100k/images", fwd_name="", bwd_name="")
    assert ff.flow_roots(a, "train") == ("/data/bdd100k/flow/pth/train/forward", "/data/bdd100k/flow/pth/train/backward")
    assert a.flow_root == "/data/bdd100k/flow/pth/train" and (a.fwd_name, a.bwd_name) == ("forward", "backward")
    b = types.SimpleNamespace(use_flow_file=True, flow_root="/f", data_dir="/d", fwd_name="fw", bwd_name="bw")
    assert ff.flow_roots(b, "val") == ("/f/val/fw", "/f/val/bw")
    assert ff.flow_roots(types.SimpleNamespace(use_flow_file=False)) == ("", "")


def test_errors(files, tmp_path):
    from contrast import flow_files as ff
    with pytest.raises(NotImplementedError):
        ff.load_flow(str(tmp_path / "x.npy"), 0, 1)
    with pytest.raises(FileNotFoundError):
        ff.video_flow_paths(str(tmp_path), str(tmp_path), "/videos/clip0/frame0.jpg")
    torch.save({"not": "a tensor"}, tmp_path / "bad.pth")
    with pytest.raises(ValueError):
        ff.load_flow(str(tmp_path / "bad.pth"), 0, 1)
    os.makedirs(tmp_path / "fw"), os.makedirs(tmp_path / "bw")
    torch.save(torch.zeros(1, 2, 2, 2), tmp_path / "fw" / "clip0.pth")
    torch.save(torch.zeros(1, 2, 2, 2), tmp_path / "bw" / "clip0.pth")
    assert ff.video_flow_paths(str(tmp_path / "fw"), str(tmp_path / "bw"), "/videos/clip0/frame0.jpg") == \
        (str(tmp_path / "fw" / "clip0.pth"), str(tmp_path / "bw" / "clip0.pth"))


def test_clip_batch_collates(gold, files):
    from contrast import flow_files as ff
    fp, bp = files[9]
    batch = ff.FlowClipBatch(3, 2, 3, 4, pin=False)
    want_f, want_b = [], []
    for i in (0, 3, 6):
        s, n, bs, bn = ff.clip_window(i, 9, 3)
        batch.add([fp, s, n], [bp, bs, bn])
        want_f.append(torch.from_numpy(gold["file_fwd_9"][s:n]))
        want_b.append(torch.from_numpy(gold["file_bwd_9"][bs:bn]))
    f, b = batch.to("cpu")
    assert torch.equal(f, torch.stack(want_f)) and torch.equal(b, torch.stack(want_b))
    with pytest.raises(IndexError):
        batch.add([fp, 0, 2], [bp, 0, 2])
    short = ff.FlowClipBatch(2, 2, 3, 4, pin=False)
    with pytest.raises(ValueError):
        short.add([files[2][0], 0, 1], [files[2][1], 0, 1])     # a 2-frame video in a 3-frame batch
    with pytest.raises(ValueError):
        short.to("cpu")


@pytest.mark.gpu
def test_files_to_flow_prepare(tmp_path):
    """files -> FlowClipBatch -> util.apply_optical_flow on the device == the oracle on the same windows."""
    import argparse
    from contrast import flow_files as ff, util
    import oracle.pixpro_oracle as O
    g = torch.Generator().manual_seed(5)
    n_video, F = 7, 3
    fwd = torch.nn.functional.avg_pool2d(torch.randn(n_video - 1, 2, 6, 8, generator=g), 3, 1, 1)
    bwd = -fwd.flip(0)
    torch.save(fwd, tmp_path / "v.pth"), torch.save(bwd, tmp_path / "w.pth")
    batch = ff.FlowClipBatch(4, F - 1, 6, 8)
    for i in (0, 1, 3, 4):
        s, n, bs, bn = ff.clip_window(i, n_video, F)
        batch.add([str(tmp_path / "v.pth"), s, n], [str(tmp_path / "w.pth"), bs, bn])
    lo_f, lo_b = batch.to("cuda")
    args = argparse.Namespace(use_flow_file=True, use_flow_frames=False, flow_cat_norm=False, flow_up=True,
                              alpha1=0.01, alpha2=0.5, debug=False)
    size = torch.tensor([[48, 64]] * 4)
    data = [None] * 5 + [[None, lo_f, lo_b], [size, torch.tensor([[F]] * 4)]]
    (ff_, _, mf), (fb_, _, mb) = util.apply_optical_flow(data, None, args)
    (wf, _, wmf), (wb, _, wmb) = O.apply_optical_flow_file(lo_f.cpu(), lo_b.cpu(), size[0], True, 0.01, 0.5)
    assert torch.equal(ff_.cpu(), wf) and torch.equal(fb_.cpu(), wb)
    assert torch.equal(mf.cpu().bool(), wmf) and torch.equal(mb.cpu().bool(), wmb)
