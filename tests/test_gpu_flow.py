"""HIP flow preparation vs the reference's golden vectors (bit-exact targets) and the oracle."""
import types

import pytest
import torch

from conftest import load_golden
from gpu_utils import to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["", "_big", "_edge128", "_edge136", "_edge136w"])
def test_upflow8_bitwise(dev, tag):
    """Both of ATen's CPU bilinear kernels (switch at output_H + output_W = 128) and the threshold itself."""
    from contrast import util
    g = load_golden("upflow8" + tag)
    out = util.upflow8(g["flow"].to(dev)).cpu()
    assert torch.equal(out, g["out"])


@pytest.mark.parametrize("n", [1, 2, 5])
def test_concat_flow_bitwise(dev, n):
    from contrast import util
    g = load_golden(f"concat_flow_{n}")
    assert torch.equal(util.concat_flow(g["flows"].to(dev)).cpu(), g["out"])
    assert torch.equal(util.concat_flow(g["flows"].to(dev), is_norm=True).cpu(), g["out_norm"])


def test_fbc_exact(dev):
    from contrast import util
    g = load_golden("fbc")
    _, _, (m, cyc) = util.forward_backward_consistency(g["flow_fwd"].to(dev), g["flow_bwd"].to(dev),
                                                       alpha_1=g["alpha1"], alpha_2=g["alpha2"])
    assert torch.equal(m.cpu(), g["mask"])
    assert torch.equal(cyc.cpu(), g["cycle"])
    r = util.calc_mask_ratio(m).cpu()
    assert torch.allclose(r, g["ratio"], atol=1e-6, rtol=0)


def _args(up, **kw):
    base = dict(alpha1=0.01, alpha2=0.5, use_flow_frames=False, use_flow_file=True, flow_up=up,
                flow_cat_norm=False, debug=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


def _data(fw, bw, size, nfr, dev):
    N = fw.shape[0]
    return [None, None, None, None, None, [torch.zeros(N, device=dev), fw.to(dev), bw.to(dev)],
            [size.to(dev).repeat(N, 1), torch.full((N, 1), nfr, device=dev)]]


@pytest.mark.parametrize("tag", ["f2up", "f6up", "f3noup", "f2up_big", "f6up_big", "f2up_bigmotion"])
def test_apply_optical_flow_golden(dev, tag):
    from contrast import util
    g = load_golden(f"applyflow_{tag}")
    fw, bw = g["flow_fwds"], g["flow_bwds"]
    (f, s, mf), (b, _, mb) = util.apply_optical_flow(_data(fw, bw, g["size"], fw.shape[1] + 1, dev), None,
                                                     _args(bool(g["flow_up"])))
    assert torch.equal(f.cpu(), g["flow_fwd"]) and torch.equal(b.cpu(), g["flow_bwd"])
    assert torch.equal(mf.cpu(), g["mask_fwd"]) and torch.equal(mb.cpu(), g["mask_bwd"])
    assert torch.equal(s.cpu(), g["size"])
    # counts carried by the fused launch == dense recount
    r1 = util.calc_mask_ratio(mf).cpu()
    r2 = (~g["mask_fwd"]).float().mean((-1, -2))
    assert torch.allclose(r1, r2, atol=1e-6, rtol=0)


@pytest.mark.parametrize("nfr,hw,sigma", [(2, (90, 160), 0.5), (6, (45, 80), 0.5), (3, (23, 37), 0.5),
                                          (2, (45, 80), 12.0), (2, (17, 29), 3.0), (2, (2, 2), 0.3),
                                          (4, (30, 40), 8.0), (8, (12, 20), 0.4)])
def test_apply_optical_flow_vs_oracle_full_size(dev, nfr, hw, sigma):
    """BDD-sized (720x1280 for nframe 2) and ragged sizes against the oracle; flows bit-exact,
    masks exact."""
    from contrast import util
    from oracle import pixpro_oracle as O
    from recipes import smooth_flow
    gen = torch.Generator().manual_seed(5 + nfr)
    N = 2
    # sigma 12 low-res px = ~100 full-res px displacements: taps leave the kernel's LDS window (global path)
    fw = smooth_flow((N, nfr - 1, 2, *hw), sigma, gen)
    bw = -fw.flip(1) + smooth_flow((N, nfr - 1, 2, *hw), 0.1 * sigma, gen)
    size = torch.tensor([8 * hw[0], 8 * hw[1]])
    (f, _, mf), (b, _, mb) = util.apply_optical_flow(_data(fw, bw, size, nfr, dev), None, _args(True))
    (fo, _, mfo), (bo, _, mbo) = O.apply_optical_flow_file(fw, bw, size, True, 0.01, 0.5)
    assert torch.equal(f.cpu(), fo) and torch.equal(b.cpu(), bo)
    assert torch.equal(mf.cpu(), mfo) and torch.equal(mb.cpu(), mbo)
    assert 0 < mfo.float().mean() < 1 or hw == (2, 2)


def test_apply_optical_flow_variants(dev):
    """alpha None -> no masks; flow_cat_norm staged variant vs oracle."""
    from contrast import util
    from oracle import pixpro_oracle as O
    from recipes import smooth_flow
    gen = torch.Generator().manual_seed(9)
    fw = smooth_flow((2, 2, 2, 12, 16), 0.5, gen)
    bw = -fw.flip(1)
    size = torch.tensor([96, 128])
    (f, _, mf), _ = util.apply_optical_flow(_data(fw, bw, size, 3, dev), None, _args(True, alpha1=None, alpha2=None))
    assert mf is None
    (fo, _, _), _ = O.apply_optical_flow_file(fw, bw, size, True, None, None)
    assert torch.equal(f.cpu(), fo)
    (f, _, mf), (b, _, mb) = util.apply_optical_flow(_data(fw, bw, size, 3, dev), None, _args(True, flow_cat_norm=True))
    (fo, _, mfo), (bo, _, mbo) = O.apply_optical_flow_file(fw, bw, size, True, 0.01, 0.5, flow_cat_norm=True)
    assert torch.allclose(f.cpu(), fo, atol=1e-4) and torch.allclose(b.cpu(), bo, atol=1e-4)
    assert (mf.cpu() != mfo).float().mean() < 1e-3


def test_flow_properties_full_size(dev):
    """Size-independent properties at BDD100k size: chain of zeros is zero; fbc(f,-f) for constant f is
    valid exactly where x+f stays inside; single-flow chain is the identity."""
    from contrast import util
    z = torch.zeros(3, 1, 2, 720, 1280, device=dev)
    assert util.concat_flow(z).abs().max().item() == 0
    f = torch.full((1, 2, 720, 1280), 2.5, device=dev)
    m = util.forward_backward_consistency(f, -f)[2][0]
    inside = torch.zeros(720, 1280, dtype=torch.bool)
    inside[:717, :1277] = True
    assert torch.equal(m[0].cpu(), inside)
    one = torch.randn(1, 2, 2, 64, 64, device=dev)
    assert torch.equal(util.concat_flow(one), one[0])


@pytest.mark.parametrize("nfr,N", [(6, 4), (2, 5)])
def test_apply_optical_flow_bdd_size_batch_vs_oracle(dev, nfr, N):
    """the dense pipeline at the production frame size (720x1280) with a real batch (sample index in
    the grid) against the oracle bit for bit - nframe=6 is the reference's production recipe
    (tools/pretrain_bdd100k_job_2000ep_nframe6_gpu16.sh:26-51)."""
    from contrast import util
    from oracle import pixpro_oracle as O
    from recipes import smooth_flow
    gen = torch.Generator().manual_seed(40 + nfr)
    hw = (90, 160)
    fw = smooth_flow((N, nfr - 1, 2, *hw), 0.5, gen)
    fw[1] *= 6.0                                  # one sample with large displacements (taps beyond the LDS windows)
    bw = -fw.flip(1) + smooth_flow((N, nfr - 1, 2, *hw), 0.05, gen)
    size = torch.tensor([720, 1280])
    (f, _, mf), (b, _, mb) = util.apply_optical_flow(_data(fw, bw, size, nfr, dev), None, _args(True))
    with O.aten_samplers():                       # the ATen kernels the reference calls; bit-identical to the restatement
        (fo, _, mfo), (bo, _, mbo) = O.apply_optical_flow_file(fw, bw, size, True, 0.01, 0.5)
    assert torch.equal(f.cpu(), fo) and torch.equal(b.cpu(), bo)
    assert torch.equal(mf.cpu(), mfo) and torch.equal(mb.cpu(), mbo)
    for n in range(N):
        assert 0 < mfo[n].float().mean() < 1
    r = util.calc_mask_ratio(mf).cpu()
    assert torch.allclose(r, (~mfo).float().mean((-1, -2)), atol=1e-6, rtol=0)


@pytest.mark.parametrize("nfr", [2, 4])
@pytest.mark.parametrize("case", ["nan", "inf", "huge", "alpha2_tiny", "alpha2_zero", "alpha1_zero", "alpha1_negative",
                                  "coherent_40px", "coherent_18px", "motion_boundary", "zero"])
def test_flow_prepare_adversarial_inputs_vs_oracle(dev, case, nfr):
    """The certified kernels (nframe = 2: bound from the low-res cells; nframe > 2: bound from the block boxes of the
    composed flows) decide most mask bits from a bound; whatever the inputs do to that bound (NaN / Inf cells, flows far
    beyond the window, tolerances that certify nothing, a motion boundary inside a tile) the bits must stay the
    oracle's.  Several tiles per map (45x80 -> 360x640)."""
    from contrast import util
    from oracle import pixpro_oracle as O
    from recipes import smooth_flow
    gen = torch.Generator().manual_seed(77)
    N, hw, F1 = 2, (45, 80), nfr - 1
    fw = smooth_flow((N, F1, 2, *hw), 0.5 / F1, gen)
    bw = -fw.flip(1) + smooth_flow((N, F1, 2, *hw), 0.05 / F1, gen)
    a1, a2 = 0.01, 0.5
    if case == "nan":
        fw[0, 0, 0, 10, 20] = float("nan"); bw[1, 0, 1, 30, 50] = float("nan")
    elif case == "inf":
        fw[0, 0, 1, 5, 5] = float("inf"); bw[1, 0, 0, 40, 70] = -float("inf")
    elif case == "huge":
        fw[0] += 1e6; bw[1, 0, :, 20:30] -= 3e4
    elif case == "alpha2_tiny":
        a2 = 1e-6
    elif case == "alpha2_zero":
        a2 = 0.0
    elif case == "alpha1_zero":
        a1 = 0.0
    elif case == "alpha1_negative":
        a1 = -0.05
    elif case == "coherent_40px":           # beyond what the cell window reaches: every tile goes to the staged kernel
        fw[:, :, 0] += 5.0; bw[:, :, 0] -= 5.0
    elif case == "coherent_18px":           # inside the window, far outside the 8-pixel halo of the staged kernel
        fw[:, :, 1] += 2.25; bw[:, :, 1] -= 2.25
    elif case == "motion_boundary":         # a step of 3 low-res pixels (24 px) through the middle of the tiles
        fw[:, :, 0, :, 37:] += 3.0; bw[:, :, 0, :, 40:] -= 3.0
    elif case == "zero":
        fw.zero_(); bw.zero_()
    size = torch.tensor([8 * hw[0], 8 * hw[1]])
    (f, _, mf), (b, _, mb) = util.apply_optical_flow(_data(fw, bw, size, nfr, dev), None, _args(True, alpha1=a1, alpha2=a2))
    with O.aten_samplers():
        (fo, _, mfo), (bo, _, mbo) = O.apply_optical_flow_file(fw, bw, size, True, a1, a2)
    assert torch.equal(torch.nan_to_num(f.cpu(), nan=123.0), torch.nan_to_num(fo, nan=123.0))
    assert torch.equal(torch.nan_to_num(b.cpu(), nan=123.0), torch.nan_to_num(bo, nan=123.0))
    assert torch.equal(mf.cpu(), mfo) and torch.equal(mb.cpu(), mbo)
    r = util.calc_mask_ratio(mf).cpu()
    assert torch.allclose(r, (~mfo).float().mean((-1, -2)), atol=1e-6, rtol=0)


def test_flow_prepare_without_workspace_equals_certified_route(dev):
    """pixpro_flow_prepare (no workspace: every tile through the staged mask kernel) and pixpro_flow_prepare_ws (block
    boxes + certified masks) are two routes to the same bits, nframe = 4 at 360x640."""
    from contrast import util, _native as nv
    from recipes import smooth_flow
    gen = torch.Generator().manual_seed(5)
    N, F1, hw = 3, 3, (45, 80)
    fw = smooth_flow((N, F1, 2, *hw), 0.3, gen)
    bw = -fw.flip(1) + smooth_flow((N, F1, 2, *hw), 0.03, gen)
    size = torch.tensor([8 * hw[0], 8 * hw[1]])
    (f, _, mf), (b, _, mb) = util.apply_optical_flow(_data(fw, bw, size, F1 + 1, dev), None, _args(True))
    lo_f, lo_b = fw.to(dev).contiguous(), bw.to(dev).contiguous()
    f2 = torch.empty_like(f); b2 = torch.empty_like(b)
    mf2 = torch.empty(mf.shape, dtype=torch.bool, device=dev); mb2 = torch.empty_like(mf2)
    cnt = torch.empty(2 * N, dtype=torch.int32, device=dev)
    assert nv.load().pixpro_flow_prepare_workspace_bytes(N, F1, hw[0], hw[1], 1) == 2 * N * hw[0] * hw[1] * 16
    assert nv.load().pixpro_flow_prepare_workspace_bytes(N, 1, hw[0], hw[1], 1) == 0
    nv.call("pixpro_flow_prepare", nv.ptr(lo_f), nv.ptr(lo_b), nv.ptr(f2), nv.ptr(b2), nv.ptr(mf2), nv.ptr(mb2),
            nv.ptr(cnt), N, F1, hw[0], hw[1], 1, 0.01, 0.5, nv.stream())
    assert torch.equal(f, f2) and torch.equal(b, b2)
    assert torch.equal(mf.as_subclass(torch.Tensor), mf2) and torch.equal(mb.as_subclass(torch.Tensor), mb2)
    assert torch.equal(cnt[:N].cpu(), (~mf2).sum((-1, -2)).int().cpu())
    # a workspace that is too small is refused, not ignored
    small = torch.empty(64, dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError):
        nv.call("pixpro_flow_prepare_ws", nv.ptr(lo_f), nv.ptr(lo_b), nv.ptr(f2), nv.ptr(b2), nv.ptr(mf2), nv.ptr(mb2),
                nv.ptr(cnt), N, F1, hw[0], hw[1], 1, 0.01, 0.5, nv.ptr(small), 64, nv.stream())


def test_flow_prepare_unaligned_outputs_and_workspace(dev):
    """The nframe = 2 kernel stores two pixels per lane (8-B flow stores, 2-B mask stores).  Output views at odd offsets take
    the one-thread-per-pixel kernel instead - the same bits - and a misaligned workspace is refused (advisor r2)."""
    from contrast import util, _native as nv
    from recipes import smooth_flow
    gen = torch.Generator().manual_seed(9)
    N, hw = 2, (45, 80)
    fw = smooth_flow((N, 1, 2, *hw), 0.3, gen)
    bw = -fw.flip(1) + smooth_flow((N, 1, 2, *hw), 0.03, gen)
    size = torch.tensor([8 * hw[0], 8 * hw[1]])
    (f, _, mf), (b, _, mb) = util.apply_optical_flow(_data(fw, bw, size, 2, dev), None, _args(True))
    lo_f, lo_b = fw.to(dev).contiguous(), bw.to(dev).contiguous()
    nf, nm = f.numel(), mf.numel()
    fbuf = torch.empty(2 * nf + 8, dtype=torch.float32, device=dev)
    mbuf = torch.empty(2 * nm + 8, dtype=torch.uint8, device=dev)
    f2, b2 = fbuf[1:1 + nf].view(f.shape), fbuf[nf + 3:2 * nf + 3].view(f.shape)          # 4-B aligned only
    mf2, mb2 = mbuf[1:1 + nm].view(mf.shape), mbuf[nm + 3:2 * nm + 3].view(mf.shape)     # odd addresses
    cnt = torch.empty(2 * N, dtype=torch.int32, device=dev)
    assert nv.ptr(f2).value % 8 == 4 and nv.ptr(mf2).value % 2 == 1
    nv.call("pixpro_flow_prepare", nv.ptr(lo_f), nv.ptr(lo_b), nv.ptr(f2), nv.ptr(b2), nv.ptr(mf2), nv.ptr(mb2),
            nv.ptr(cnt), N, 1, hw[0], hw[1], 1, 0.01, 0.5, nv.stream())
    assert torch.equal(f, f2) and torch.equal(b, b2)
    assert torch.equal(mf.as_subclass(torch.Tensor).view(torch.uint8), mf2)
    assert torch.equal(mb.as_subclass(torch.Tensor).view(torch.uint8), mb2)
    assert torch.equal(cnt[:N].cpu(), (mf2 == 0).sum((-1, -2)).int().cpu())
    ws = torch.empty(2 * N * hw[0] * hw[1] * 16 * 3 + 64, dtype=torch.uint8, device=dev)
    lo3 = torch.cat([lo_f, lo_f, lo_f], 1).contiguous()
    nbytes = nv.load().pixpro_flow_prepare_workspace_bytes(N, 3, hw[0], hw[1], 1)
    with pytest.raises(RuntimeError):
        nv.call("pixpro_flow_prepare_ws", nv.ptr(lo3), nv.ptr(lo3), nv.ptr(f), nv.ptr(b), nv.ptr(mf2), nv.ptr(mb2),
                nv.ptr(cnt), N, 3, hw[0], hw[1], 1, 0.01, 0.5, nv.ptr(ws[4:]), nbytes, nv.stream())


@pytest.mark.parametrize("nfr,hw", [(2, (37, 53)), (3, (37, 53)), (2, (11, 29)), (4, (23, 9))])
def test_flow_prepare_ragged_sizes_vs_oracle(dev, nfr, hw):
    """Frame sizes that are no multiple of the 64 x 80 (64 x 40) tiles in either direction: partial tiles, partial
    cell windows and the run / block certificates at the frame border, bit for bit against the oracle."""
    from contrast import util
    from oracle import pixpro_oracle as O
    from recipes import smooth_flow
    gen = torch.Generator().manual_seed(1000 + nfr + hw[0])
    N, F1 = 3, nfr - 1
    fw = smooth_flow((N, F1, 2, *hw), 0.6 / F1, gen)
    fw[1] *= 4.0
    bw = -fw.flip(1) + smooth_flow((N, F1, 2, *hw), 0.05 / F1, gen)
    size = torch.tensor([8 * hw[0], 8 * hw[1]])
    (f, _, mf), (b, _, mb) = util.apply_optical_flow(_data(fw, bw, size, nfr, dev), None, _args(True))
    with O.aten_samplers():
        (fo, _, mfo), (bo, _, mbo) = O.apply_optical_flow_file(fw, bw, size, True, 0.01, 0.5)
    assert torch.equal(f.cpu(), fo) and torch.equal(b.cpu(), bo)
    assert torch.equal(mf.cpu(), mfo) and torch.equal(mb.cpu(), mbo)
    r = util.calc_mask_ratio(mf).cpu()
    assert torch.allclose(r, (~mfo).float().mean((-1, -2)), atol=1e-6, rtol=0)
