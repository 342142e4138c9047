"""Every second-derivative HIP kernel (`eqf_*_bwd2`, equiformer_amd/csrc/second.hip) against the double backward of the
fp64 restatement of the same operator (tests/second_order_ref.py, itself pinned against the oracle modules by
tests/test_second_order_restatements.py).

For an operator y = f(x; theta):  L = <b, d<a, f(x; theta)> / d x>  is a scalar whose gradient wrt x, theta and a is
what `loss.backward()` asks of the operator when forces were taken with create_graph=True
[ref: nets/graph_attention_transformer_md17.py:318-325].  The HIP path gets there through
`torch.autograd.grad(..., create_graph=True)` on the ops.* autograd Functions; the reference value is the same
expression on the restatement in fp64 on the CPU.  Tolerance 2e-5 of the largest entry (fp32 kernels, sums over rows)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import second_order_ref as so  # noqa: E402

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _second_order(fn, inputs, diff, seed):
    """inputs: list of tensors; diff: indices of the inputs the FIRST derivative is taken with respect to.
    Returns (first derivatives, gradients of L = sum_k <b_k, d<a, f>/d x_k> wrt every input and wrt a)."""
    outs = fn(*inputs)
    outs = list(outs) if isinstance(outs, (tuple, list)) else [outs]
    g = torch.Generator().manual_seed(seed)
    a = [torch.randn(o.shape, generator=g, dtype=torch.float64).to(o.dtype).to(o.device).requires_grad_(True) for o in outs]
    first = torch.autograd.grad(outs, [inputs[i] for i in diff], a, create_graph=True)
    b = [torch.randn(f.shape, generator=g, dtype=torch.float64).to(f.dtype).to(f.device) for f in first]
    L = sum((bb * ff).sum() for bb, ff in zip(b, first))
    wrt = [t for t in inputs if t.requires_grad] + a
    second = torch.autograd.grad(L, wrt, allow_unused=True)
    return first, second


def _compare(fn_hip, fn_ref, inputs64, diff, seed, params=()):
    """inputs64: fp64 CPU tensors (requires_grad set by the caller)."""
    dev = _dev()
    in_ref = [t.clone().requires_grad_(t.requires_grad) for t in inputs64]
    in_hip = [t.detach().float().to(dev).requires_grad_(t.requires_grad) for t in inputs64]
    f_ref, s_ref = _second_order(fn_ref, in_ref, diff, seed)
    f_hip, s_hip = _second_order(fn_hip, in_hip, diff, seed)
    for k, (x, r) in enumerate(zip(f_hip, f_ref)):
        assert _rel(x, r) < TOL, ("first", k, _rel(x, r))
    for k, (x, r) in enumerate(zip(s_hip, s_ref)):
        if r is None or float(r.abs().max()) == 0.0:
            assert x is None or float(x.abs().max()) < 1e-6, ("second", k)
            continue
        assert x is not None, ("second", k)
        assert _rel(x, r) < TOL, ("second", k, _rel(x, r))


def _rows(n, d, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, d, generator=g, dtype=torch.float64) * scale).requires_grad_(True)


def test_scaled_silu_bwd2():
    from equiformer_amd import ops, so3
    x = _rows(37, 96, 1, 2.0)
    _compare(lambda t: ops.scaled_silu(t, so3.C_SILU), lambda t: so.scaled_silu(t, so3.C_SILU), [x], [0], 2)


@pytest.mark.parametrize("irr", ["24x0e+16x1e+8x2e", "384x0e+192x1e+96x2e", "64x0e+32x1e+32x2e+16x3e"])
def test_gate_bwd2(irr):
    from equiformer_amd import ops, so3
    from equiformer_amd.layout import RowLayout
    from oracle import e3
    from oracle import nets as onets
    scalars, gates, gated = onets.irreps2gate(e3.Irreps(irr))
    lay_gated = RowLayout(gated)
    Din = scalars.dim + gates.dim + gated.dim
    x = _rows(29, Din, 3, 1.5)
    S = scalars.dim
    _compare(lambda t: ops.gate(t, S, lay_gated, so3.C_SILU, so3.C_SIGMOID),
             lambda t: so.gate(t, S, lay_gated, so3.C_SILU, so3.C_SIGMOID), [x], [0], 4)


@pytest.mark.parametrize("C", [64, 32])
def test_lnsilu_bwd2(C):
    from equiformer_amd import ops
    g = torch.Generator().manual_seed(5)
    x = _rows(301, C, 6, 1.3)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    _compare(lambda t, ga, be: ops.ln_silu(t, ga, be, 1e-5), lambda t, ga, be: so.ln_silu(t, ga, be, 1e-5),
             [x, gamma, beta], [0], 7)


@pytest.mark.parametrize("C,G", [(64, 7), (32, 3)])
def test_lnsilu_group_bwd2(C, G):
    """the radial bank's LayerNorm + SiLU: G feature vectors side by side in a row, each with its own gamma / beta
    (eqf_lnsilu_group_bwd2, round 5) against the per-group restatement"""
    from equiformer_amd import ops
    g = torch.Generator().manual_seed(15)
    x = _rows(203, C * G, 16, 1.3)
    gamma = (torch.rand(C * G, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = torch.randn(C * G, generator=g, dtype=torch.float64).requires_grad_(True)

    def ref(t, ga, be):
        return torch.cat([so.ln_silu(t[:, k * C:(k + 1) * C], ga[k * C:(k + 1) * C], be[k * C:(k + 1) * C], 1e-5)
                          for k in range(G)], dim=1)
    _compare(lambda t, ga, be: ops.ln_silu(t, ga, be, 1e-5, groups=G), ref, [x, gamma, beta], [0], 17)


@pytest.mark.parametrize("wide", [True, False])
def test_grouped_linear_second_order(wide):
    """G nn.Linear layers on the column blocks of one input in one launch (the radial bank's second and third layers):
    differentiable grouped pieces under create_graph (ops._GroupedDgrad / _GroupedWgrad, round 5)"""
    from equiformer_amd import ops
    G, K = 5, 64
    Ns = [64] * G if wide else [96, 64, 128, 64, 32]
    g = torch.Generator().manual_seed(25)
    x = _rows(157, G * K, 26, 0.7)
    Ws = [(torch.randn(n, K, generator=g, dtype=torch.float64) * 0.2).requires_grad_(True) for n in Ns]
    bs = [torch.randn(n, generator=g, dtype=torch.float64).requires_grad_(True) for n in Ns]

    def hip(t, *p):
        return ops.grouped_linear(t, K, list(p[:G]), list(p[G:]), wide)

    def ref(t, *p):
        outs = [t[:, k * K:(k + 1) * K] @ p[k].t() + p[G + k] for k in range(G)]
        return torch.cat(outs, dim=1) if wide else tuple(outs)
    _compare(hip, ref, [x] + Ws + bs, [0], 27)


@pytest.mark.parametrize("irr", ["128x0e+64x1e+32x2e", "128x0e+64x1e+64x2e+32x3e", "32x0e+16x1e"])
def test_layernorm_bwd2(irr):
    from equiformer_amd import ops
    from equiformer_amd.layout import RowLayout
    lay = RowLayout(irr)
    g = torch.Generator().manual_seed(8)
    nw = sum(m for m, _ in lay.segs)
    nb = sum(m for m, l in lay.segs if l == 0)
    x = _rows(43, lay.dim, 9)
    w = (torch.rand(nw, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    b = torch.randn(nb, generator=g, dtype=torch.float64).requires_grad_(True)
    _compare(lambda t, ww, bb: ops.layer_norm(t, ww, bb, lay, 1e-5), lambda t, ww, bb: so.layer_norm(t, ww, bb, lay, 1e-5),
             [x, w, b], [0], 10)


@pytest.mark.parametrize("H,Kh", [(4, 32), (8, 32), (4, 16)])
def test_alpha_bwd2(H, Kh):
    from equiformer_amd import ops, so3
    g = torch.Generator().manual_seed(11)
    a = _rows(211, H * Kh, 12, 1.5)
    ad = torch.randn(H * Kh, generator=g, dtype=torch.float64).requires_grad_(True)
    c = so3.C_SMOOTH_LEAKY_RELU_02
    _compare(lambda t, d: ops.alpha_logits(t, d, H, Kh, c), lambda t, d: so.alpha_logits(t, d, H, Kh, c), [a, ad], [0], 13)


@pytest.mark.parametrize("head_irr,H", [("32x0e+16x1e+8x2e", 4), ("32x0e+16x1e+16x2e+8x3e", 4), ("32x0e+16x1e", 8)])
def test_attn_aggregate_bwd2(head_irr, H):
    from equiformer_amd import ops
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.irreps import Irreps
    from equiformer_amd.layout import RowLayout
    from equiformer_amd.synthetic import qm9_like_batch
    dev = _dev()
    d = qm9_like_batch(5, 14, side=4.0, seed=4)
    graph = EdgeGraph.from_radius(d["pos"].to(dev), d["batch"].to(dev), 5.0)
    lay = RowLayout(" + ".join("%dx%de" % (mul * H, ir.l) for mul, ir in Irreps(head_irr)))
    E = graph.E
    g = torch.Generator().manual_seed(14)
    logit = (torch.randn(E, H, generator=g, dtype=torch.float64) * 2.0).requires_grad_(True)
    value = torch.randn(E, lay.dim, generator=g, dtype=torch.float64).requires_grad_(True)
    from types import SimpleNamespace
    gcpu = SimpleNamespace(dst=graph.dst.cpu(), N=graph.N)
    _compare(lambda lo, va: ops.attn_aggregate(lo, va, graph, H, lay), lambda lo, va: so.attn_aggregate(lo, va, gcpu, H, lay),
             [logit, value], [0, 1], 15)


def test_attn_aggregate_bwd2_with_dropout_is_consistent():
    """With alpha_drop > 0 there is no restatement to compare with (the mask is a hash of the seed); the second-order
    kernel must agree with finite differences of the first-order backward under the SAME seed."""
    from equiformer_amd import ops
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.layout import RowLayout
    from equiformer_amd.synthetic import qm9_like_batch
    dev = _dev()
    d = qm9_like_batch(3, 10, side=4.0, seed=5)
    graph = EdgeGraph.from_radius(d["pos"].to(dev), d["batch"].to(dev), 5.0)
    H, lay = 4, RowLayout("128x0e+64x1e")
    E = graph.E
    g = torch.Generator().manual_seed(16)
    logit = torch.randn(E, H, generator=g).to(dev).requires_grad_(True)
    value = torch.randn(E, lay.dim, generator=g).to(dev).requires_grad_(True)
    a = torch.randn(graph.N, lay.dim, generator=g).to(dev)
    bl = torch.randn(E, H, generator=g).to(dev)
    bv = torch.randn(E, lay.dim, generator=g).to(dev)

    def first(lo, va):
        out = ops.attn_aggregate(lo, va, graph, H, lay, 0.25, 1234)
        return torch.autograd.grad(out, [lo, va], a, create_graph=True)

    gl, gv = first(logit, value)
    L = (bl * gl).sum() + (bv * gv).sum()
    s_lo, s_va = torch.autograd.grad(L, [logit, value])
    # directional finite differences of L along random directions
    for which, s in ((0, s_lo), (1, s_va)):
        v = torch.randn(s.shape, generator=g).to(dev)
        eps = 1e-2
        vals = []
        for sgn in (1.0, -1.0):
            lo = (logit + sgn * eps * v).detach().requires_grad_(True) if which == 0 else logit.detach().requires_grad_(True)
            va = (value + sgn * eps * v).detach().requires_grad_(True) if which == 1 else value.detach().requires_grad_(True)
            g1, g2 = first(lo, va)
            vals.append(((bl * g1).sum() + (bv * g2).sum()).item())
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = (s * v).sum().item()
        assert abs(fd - an) < 2e-2 * max(1.0, abs(an)), (which, fd, an)


def test_rbf_expnorm_bwd2():
    from equiformer_amd import ops
    from equiformer_amd.nets.layers import ExpNormalSmearing
    rbf = ExpNormalSmearing(cutoff_lower=0.0, cutoff_upper=5.0, num_rbf=32, trainable=False)
    g = torch.Generator().manual_seed(17)
    length = (torch.rand(257, generator=g, dtype=torch.float64) * 5.5 + 0.3).requires_grad_(True)
    means, betas = rbf.means.double(), rbf.betas.double()
    dev = _dev()
    _compare(lambda t: ops.rbf_expnorm(t, means.float().to(dev), betas.float().to(dev), rbf.alpha, 5.0),
             lambda t: so.rbf_expnorm(t, means, betas, rbf.alpha, 5.0), [length], [0], 18)


def test_rbf_gaussian_bwd2():
    from equiformer_amd import ops
    g = torch.Generator().manual_seed(19)
    R = 128
    length = (torch.rand(193, generator=g, dtype=torch.float64) * 4.5 + 0.4).requires_grad_(True)
    mean = torch.rand(1, R, generator=g, dtype=torch.float64).requires_grad_(True)
    std = (torch.rand(1, R, generator=g, dtype=torch.float64) * 0.9 + 0.1).requires_grad_(True)
    weight = torch.tensor([[1.1]], dtype=torch.float64).requires_grad_(True)
    bias = torch.tensor([[0.05]], dtype=torch.float64).requires_grad_(True)

    def ref(le, mu, sd, w, b):  # [ref: nets/gaussian_rbf.py:6-40]
        x = w * (le / 5.0).unsqueeze(-1) + b
        s = sd.abs() + 1e-5
        return torch.exp(-0.5 * ((x - mu) / s) ** 2) / ((2 * 3.14159) ** 0.5 * s)

    _compare(lambda le, mu, sd, w, b: ops.rbf_gaussian(le, mu, sd, w, b, 5.0), ref, [length, mean, std, weight, bias], [0], 20)


def test_rbf_bessel_fwd_bwd_bwd2():
    """Spherical Bessel basis (ocpmodels RadialBasis, restated in oracle/nets.py) : values, first and second derivatives
    wrt the edge length, the trainable frequencies and the output cotangent."""
    from equiformer_amd import ops
    from oracle import nets as onets
    ref = onets.RadialBasis(16, cutoff=5.0, rbf={"name": "spherical_bessel"}).double()
    g = torch.Generator().manual_seed(23)
    length = (torch.rand(211, generator=g, dtype=torch.float64) * 5.4 + 0.5).requires_grad_(True)  # some beyond the cutoff
    freq = (ref.rbf.frequencies.detach().double() * (1.0 + 0.01 * torch.randn(16, generator=g, dtype=torch.float64))).requires_grad_(True)

    def fref(le, fr):
        x = le / 5.0
        env = 1 + ref.a * x ** 5 + ref.b * x ** 6 + ref.c * x ** 7
        env = torch.where(x < 1, env, torch.zeros_like(x))
        return env[:, None] * (ref.rbf.norm_const / x[:, None] * torch.sin(fr * x[:, None]))

    with torch.no_grad():  # the closed form above IS the module
        assert (fref(length, ref.rbf.frequencies.double()) - ref(length)).abs().max() < 1e-12
    _compare(lambda le, fr: ops.rbf_bessel(le, fr, 5.0), fref, [length, freq], [0], 24)
    # first-order gradient wrt the frequencies (a parameter: no second derivative through it, that raises by design)
    dev = _dev()
    le, fr = length.detach().float().to(dev).requires_grad_(True), freq.detach().float().to(dev).requires_grad_(True)
    go = torch.randn(211, 16, generator=g, dtype=torch.float64)
    g_hip = torch.autograd.grad(ops.rbf_bessel(le, fr, 5.0), [le, fr], go.float().to(dev))
    g_ref = torch.autograd.grad(fref(length, freq), [length, freq], go)
    for x, r in zip(g_hip, g_ref):
        assert _rel(x, r) < TOL


@pytest.mark.parametrize("lmax", [1, 2, 3])
def test_edge_geometry_bwd2(lmax):
    from types import SimpleNamespace
    from equiformer_amd import ops
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.synthetic import qm9_like_batch
    dev = _dev()
    d = qm9_like_batch(4, 9, side=4.0, seed=6)
    graph = EdgeGraph.from_radius(d["pos"].to(dev), d["batch"].to(dev), 5.0)
    gcpu = SimpleNamespace(src=graph.src.cpu(), dst=graph.dst.cpu())
    pos = d["pos"].double().requires_grad_(True)

    def hip(p):
        _, length, sh = ops.edge_geometry(p, None, graph, lmax)
        return length, sh

    _compare(hip, lambda p: so.edge_geometry(p, None, gcpu, lmax), [pos], [0], 21)
