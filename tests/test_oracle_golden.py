"""Pin the CPU oracle (oracle/pixpro_oracle.py) against vectors captured from the reference.

The oracle restates ATen's samplers explicitly; wherever the reference's result is a hard
decision (positive-pair counts, validity masks) we require exact equality, and for the
coordinate pipeline we require bit equality of the fp32 values."""
import pytest
import torch

from conftest import load_golden
from oracle import pixpro_oracle as O

REGLOSS = ["noflow7", "zeroflow7", "flowfull7", "flowmask7", "floweighth7", "flowbare14", "flowmask2", "c256"]


def coord_args(g):
    mode = g["mode"]
    cq, ck = g["coord_q"], g["coord_k"]
    size = tuple(int(s) for s in g["size"])
    mask = g["mask"] if g["has_mask"] else None
    if mode == "none":
        return cq, ck
    if mode == "bare":
        return [cq, g["flow"]], [ck, g["flow"]]
    return [cq, [g["flow"], size, mask]], [ck, [g["flow"], size, mask]]


@pytest.mark.parametrize("tag", ["p2", "p1", "p05", "g14", "g2"])
def test_featprop(tag):
    g = load_golden(f"featprop_{tag}")
    Y = O.featprop(g["x"], g["v"], g["p"], g["clamp"])
    assert torch.allclose(Y, g["Y"], rtol=0, atol=2e-6 * max(1.0, g["Y"].abs().max().item()))
    pred = O.l2_normalize(Y)
    assert torch.allclose(pred, g["pred"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("tag", REGLOSS)
def test_regression_loss(tag):
    g = load_golden(f"regloss_{tag}")
    a, b = coord_args(g)
    q = g["q"].clone().requires_grad_(True)
    loss, (pn, pm) = O.regression_loss(q, g["k"], a, b, g["pos_ratio"])
    loss.backward()
    assert torch.equal(pn, g["pos_num"])
    assert torch.equal(pm, g["pos_mean"])
    assert abs(loss.item() - g["loss"]) <= 1e-6
    assert torch.allclose(q.grad, g["dq"], rtol=0, atol=1e-7)
    assert g["pos_num"].sum() > 0


@pytest.mark.parametrize("tag", [t for t in REGLOSS if t not in ("noflow7",)])
def test_add_optical_flow_bitwise(tag):
    g = load_golden(f"regloss_{tag}")
    size = tuple(int(s) for s in g["size"])
    mask = g["mask"] if g["has_mask"] else None
    ox, oy, mg = O.add_optical_flow(g["flow"], g["aof_x_in"], g["aof_y_in"], size, mask)
    assert torch.equal(ox, g["aof_x"]) and torch.equal(oy, g["aof_y"])
    if mask is not None:
        assert torch.equal(mg, g["aof_mask"])


def test_zero_flow_equals_no_flow():
    a, b = load_golden("regloss_noflow7"), load_golden("regloss_zeroflow7")
    assert a["pos_num"].sum() > 0  # property 1 of SURVEY.md §4 holds per fixture pair below
    g = b
    size = tuple(int(s) for s in g["size"])
    l0, (n0, _) = O.regression_loss(g["q"], g["k"], g["coord_q"], g["coord_k"], 0.7)
    l1, (n1, _) = O.regression_loss(g["q"], g["k"], [g["coord_q"], [g["flow"], size, None]],
                                    [g["coord_k"], [g["flow"], size, None]], 0.7)
    assert torch.equal(n0, n1) and l0.item() == l1.item()


def test_mask_properties():
    g = load_golden("regloss_flowmask7")
    size = tuple(int(s) for s in g["size"])
    ones = torch.ones_like(g["mask"])
    base = O.regression_loss(g["q"], g["k"], [g["coord_q"], [g["flow"], size, None]], [g["coord_k"], None], 0.7)
    full = O.regression_loss(g["q"], g["k"], [g["coord_q"], [g["flow"], size, ones]], [g["coord_k"], None], 0.7)
    none = O.regression_loss(g["q"], g["k"], [g["coord_q"], [g["flow"], size, ~ones]], [g["coord_k"], None], 0.7)
    assert torch.equal(base[1][0], full[1][0]) and base[0].item() == full[0].item()
    assert none[1][0].sum() == 0 and none[0].item() == 0.0


@pytest.mark.parametrize("tag", ["e2e7", "e2e14", "e2e2"])
def test_hotpath_end_to_end(tag):
    g = load_golden(f"hotpath_{tag}")
    F = torch.nn.functional
    C = g["x1"].shape[1]
    x1 = g["x1"].clone().requires_grad_(True)
    x2 = g["x2"].clone().requires_grad_(True)
    wv = g["wv"].clone().requires_grad_(True)
    bv = g["bv"].clone().requires_grad_(True)
    size = tuple(int(s) for s in g["size"])
    a = [g["coord1"], [g["flow_fwd"], size, g["mask_fwd"]]]
    b = [g["coord2"], [g["flow_bwd"], size, g["mask_bwd"]]]
    v1 = F.conv2d(x1, wv.view(C, C, 1, 1), bv)
    v2 = F.conv2d(x2, wv.view(C, C, 1, 1), bv)
    l1, p1, n1, m1 = O.ppm_loss(x1, v1, g["k2"], a, b, g["p"], 0.0, 0.7, normalize_k=True)
    l2, p2, n2, m2 = O.ppm_loss(x2, v2, g["k1"], b, a, g["p"], 0.0, 0.7, normalize_k=True)
    (l1 + l2).backward()
    assert torch.equal(n1, g["pos_num_1"]) and torch.equal(n2, g["pos_num_2"])
    assert abs((l1 + l2).item() - g["loss"]) <= 1e-6
    assert torch.allclose(p1, g["pred1"], atol=2e-6, rtol=0)
    for mine, ref in ((x1.grad, g["dx1"]), (x2.grad, g["dx2"]), (wv.grad, g["dwv"]), (bv.grad, g["dbv"])):
        assert (mine - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-9

    # analytic backward (what the HIP kernel implements) vs autograd
    N, _, H, W = g["x1"].shape
    M = O.positive_mask(g["coord1"], g["coord2"], H, W, size, 0.7, g["flow_fwd"], g["mask_fwd"]).float()
    xd, vd = g["x1"].reshape(N, C, -1), v1.detach().reshape(N, C, -1)
    kd = O.l2_normalize(g["k2"]).reshape(N, C, -1)
    xa = xd.clone().requires_grad_(True)
    va = vd.clone().requires_grad_(True)
    la, _, _, _ = O.ppm_loss(xa.reshape(N, C, H, W), va.reshape(N, C, H, W), g["k2"], a, b, g["p"], 0.0, 0.7, True)
    la.backward()
    loss_c, pred_c, dx_c, dv_c = O.ppm_loss_grad_analytic(xd, vd, kd, M, g["p"], 0.0)
    assert abs(loss_c.item() - la.item()) <= 1e-6
    assert (dx_c - xa.grad).abs().max() <= 1e-4 * xa.grad.abs().max() + 1e-9
    assert (dv_c - va.grad).abs().max() <= 1e-4 * va.grad.abs().max() + 1e-9


@pytest.mark.parametrize("p,clamp", [(1.0, 0.0), (0.5, 0.1), (2.0, 0.2), (3.0, 0.0)])
def test_analytic_backward_general_p(p, clamp):
    gen = torch.Generator().manual_seed(3)
    N, C, P = 2, 32, 9
    x = torch.randn(N, C, P, generator=gen, dtype=torch.float64)
    v = torch.randn(N, C, P, generator=gen, dtype=torch.float64)
    k = O.l2_normalize(torch.randn(N, C, P, generator=gen, dtype=torch.float64))
    M = (torch.rand(N, P, P, generator=gen) > 0.5).double()
    xa, va = x.clone().requires_grad_(True), v.clone().requires_grad_(True)
    pred = O.l2_normalize(O.featprop(xa.reshape(N, C, 3, 3), va.reshape(N, C, 3, 3), p, clamp)).reshape(N, C, P)
    logit = torch.bmm(pred.transpose(1, 2), k)
    loss = -2 * ((logit * M).sum((1, 2)) / (M.sum((1, 2)) + 1e-6)).mean()
    loss.backward()
    _, _, dx, dv = O.ppm_loss_grad_analytic(x, v, k, M, p, clamp)
    assert torch.allclose(dx, xa.grad, atol=1e-10) and torch.allclose(dv, va.grad, atol=1e-10)


@pytest.mark.parametrize("tag", ["", "_big", "_edge128", "_edge136", "_edge136w"])
def test_upflow8_bitwise(tag):
    """'' = 48x64 (ATen's small-output kernel); _big = 72x96, _edge136 = 72x64, _edge136w = 56x80 (the other kernel);
    _edge128 = 64x64 sits exactly on the threshold output_H + output_W <= 128."""
    g = load_golden("upflow8" + tag)
    assert torch.equal(O.upflow8(g["flow"]), g["out"])


@pytest.mark.parametrize("n", [1, 2, 5])
def test_concat_flow_bitwise(n):
    g = load_golden(f"concat_flow_{n}")
    assert torch.equal(O.concat_flow(g["flows"]), g["out"])
    assert torch.equal(O.concat_flow(g["flows"], is_norm=True), g["out_norm"])


def test_fbc():
    g = load_golden("fbc")
    m, cyc = O.forward_backward_consistency(g["flow_fwd"], g["flow_bwd"], g["alpha1"], g["alpha2"])
    assert torch.equal(m, g["mask"]) and torch.equal(cyc, g["cycle"])
    assert torch.equal(O.calc_mask_ratio(m), g["ratio"])
    assert 0.02 < g["ratio"].mean() < 0.98  # the fixture exercises both outcomes


def test_fbc_properties():
    f = torch.full((1, 2, 16, 24), 1.5)
    m, _ = O.forward_backward_consistency(f, -f)
    inside = torch.zeros(16, 24, dtype=torch.bool)
    inside[:14, :22] = True  # x+1.5 < 23, y+1.5 < 15
    assert torch.equal(m[0], inside)
    z = torch.zeros(3, 1, 2, 8, 8)
    assert torch.equal(O.concat_flow(z), torch.zeros(1, 2, 8, 8))


@pytest.mark.parametrize("tag", ["f2up", "f6up", "f3noup", "f2up_big", "f6up_big", "f2up_bigmotion"])
def test_apply_optical_flow_file(tag):
    g = load_golden(f"applyflow_{tag}")
    (f, s, mf), (b, _, mb) = O.apply_optical_flow_file(g["flow_fwds"], g["flow_bwds"], g["size"], bool(g["flow_up"]),
                                                       g["alpha1"], g["alpha2"])
    assert torch.equal(f, g["flow_fwd"]) and torch.equal(b, g["flow_bwd"])
    assert torch.equal(mf, g["mask_fwd"]) and torch.equal(mb, g["mask_bwd"])
    assert 0.0 < mf.float().mean() < 1.0


def test_lars_trajectory():
    g = load_golden("lars")
    n = g["n"]
    p = [g[f"p0_{i}"] for i in range(n)]
    buf = [None] * n
    for s in range(3):
        for i in range(n):
            adapt = p[i].ndim > 1
            wd = g["weight_decay"] if adapt else 0.0
            p[i], buf[i] = O.lars_sgd_step(p[i], g[f"g{s}_{i}"], buf[i], g["lr"], g["momentum"], wd, adapt)
            assert torch.allclose(p[i], g[f"p{s + 1}_{i}"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("hw", [(2, 2), (5, 7), (8, 8), (7, 9), (9, 7), (9, 8), (8, 9), (2, 14), (2, 15), (3, 21), (13, 5),
                                (17, 29), (45, 80), (90, 160)])
def test_aten_samplers_agree_on_upflow8_at_every_size(hw):
    """The restated up-sampling against torch's own CPU kernels on both sides of the output_H + output_W = 128 switch
    (h + w = 16), incl. the production size 90x160 -> 720x1280."""
    lo = torch.randn(2, 2, *hw, generator=torch.Generator().manual_seed(hw[0] * 100 + hw[1]))
    with O.aten_samplers():
        ref = O.upflow8(lo)
    assert torch.equal(O.upflow8(lo), ref)


def test_aten_samplers_agree_production_size_chain():
    """nframe = 3 at 720x1280 (chain + masks): explicit restatement == ATen kernels, bit for bit."""
    from recipes import smooth_flow
    gen = torch.Generator().manual_seed(3)
    fw = smooth_flow((1, 2, 2, 90, 160), 0.5, gen)
    bw = -fw.flip(1) + smooth_flow((1, 2, 2, 90, 160), 0.05, gen)
    size = torch.tensor([720, 1280])
    ref = O.apply_optical_flow_file(fw, bw, size, True, 0.01, 0.5)
    with O.aten_samplers():
        fast = O.apply_optical_flow_file(fw, bw, size, True, 0.01, 0.5)
    for a, b in zip(ref, fast):
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


def test_aten_samplers_agree():
    """The timing variant of the oracle (ATen samplers, used only by bench.py's cpu_baseline) is bit-identical
    to the explicit restatement."""
    g = load_golden("applyflow_f6up")
    ref = O.apply_optical_flow_file(g["flow_fwds"], g["flow_bwds"], g["size"], True, 0.01, 0.5)
    with O.aten_samplers():
        fast = O.apply_optical_flow_file(g["flow_fwds"], g["flow_bwds"], g["size"], True, 0.01, 0.5)
    for a, b in zip(ref, fast):
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    g = load_golden("regloss_flowmask7")
    size = tuple(int(s) for s in g["size"])
    a = [g["coord_q"], [g["flow"], size, g["mask"]]]
    b = [g["coord_k"], None]
    l0, (n0, _) = O.regression_loss(g["q"], g["k"], a, b, 0.7)
    with O.aten_samplers():
        l1, (n1, _) = O.regression_loss(g["q"], g["k"], a, b, 0.7)
    assert torch.equal(n0, n1) and l0.item() == l1.item()
