"""The differentiable tensor-op restatements that the second-order (create_graph) path of the HIP operators uses
(tests/second_order_ref.py) are plain torch code in the channel-fastest layout, so they can be pinned on the CPU:
values, first derivatives and second derivatives (vjp of the vjp) against the oracle modules (e3nn layout)."""
from types import SimpleNamespace

import torch

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import second_order_ref as so  # noqa: E402
from equiformer_amd import so3
from equiformer_amd.layout import RowLayout
from oracle import e3
from oracle import nets as onets

torch.set_default_dtype(torch.float32)


def _pair(layout, n, seed):
    """random rows in e3nn layout (fp64, requires grad) and the same rows in CF layout"""
    g = torch.Generator().manual_seed(seed)
    xe = torch.randn(n, layout.dim, generator=g, dtype=torch.float64)
    xc = xe[:, layout.perm_from_e3nn()]
    return xe.requires_grad_(True), xc.clone().requires_grad_(True)


def _check_second_order(fe, xe, fc, xc, layout_in, layout_out, seed):
    """f(x), <a, f(x)> gradient and the gradient of <b, d<a,f>/dx> agree between the oracle (e3nn layout) and the
    restatement (CF layout)."""
    g = torch.Generator().manual_seed(seed)
    ye, yc = fe(xe), fc(xc)
    pin, pout = layout_in.perm_from_e3nn(), layout_out.perm_from_e3nn()
    assert (ye[:, pout] - yc).abs().max() < 1e-12 * max(1.0, float(ye.detach().abs().max()))
    ae = torch.randn(ye.shape, generator=g, dtype=torch.float64)
    be = torch.randn(xe.shape, generator=g, dtype=torch.float64)
    (ge,) = torch.autograd.grad((ae * ye).sum(), xe, create_graph=True)
    (gc,) = torch.autograd.grad((ae[:, pout] * yc).sum(), xc, create_graph=True)
    assert (ge[:, pin] - gc).abs().max() < 1e-11 * max(1.0, float(ge.detach().abs().max()))
    (he,) = torch.autograd.grad((be * ge).sum(), xe)
    (hc,) = torch.autograd.grad((be[:, pin] * gc).sum(), xc)
    assert (he[:, pin] - hc).abs().max() < 1e-10 * max(1.0, float(he.abs().max()))


def test_layer_norm_restatement():
    irr = "32x0e+16x1e+8x2e+8x3e"
    lay = RowLayout(irr)
    ref = onets.EquivariantLayerNormV2(irr).double()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        ref.affine_weight.copy_(torch.rand(ref.affine_weight.shape, generator=g, dtype=torch.float64) + 0.5)
        ref.affine_bias.copy_(torch.randn(ref.affine_bias.shape, generator=g, dtype=torch.float64))
    xe, xc = _pair(lay, 7, 1)
    _check_second_order(lambda x: ref(x), xe,
                        lambda x: so.layer_norm(x, ref.affine_weight.detach(), ref.affine_bias.detach(), lay, ref.eps),
                        xc, lay, lay, 2)


def test_gate_restatement():
    irr_out = e3.Irreps("24x0e+16x1e+8x2e")
    scalars, gates, gated = onets.irreps2gate(irr_out)
    ref = onets.Gate(scalars, gates, gated).double()
    lay_in, lay_out, lay_gated = RowLayout(ref.irreps_in), RowLayout((scalars + gated).simplify()), RowLayout(gated)
    xe, xc = _pair(lay_in, 9, 3)
    _check_second_order(lambda x: ref(x), xe,
                        lambda x: so.gate(x, scalars.dim, lay_gated, so3.C_SILU, so3.C_SIGMOID), xc, lay_in, lay_out, 4)


def test_scalar_restatements():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(11, 64, generator=g, dtype=torch.float64)
    act = onets.ScaledAct(torch.nn.functional.silu)
    assert (so.scaled_silu(x, so3.C_SILU) - act(x)).abs().max() < 1e-7  # the constants are fp32 Monte-Carlo values
    ln = torch.nn.LayerNorm(64).double()
    with torch.no_grad():
        ln.weight.copy_(torch.rand(64, generator=g, dtype=torch.float64) + 0.5)
        ln.bias.copy_(torch.randn(64, generator=g, dtype=torch.float64))
    want = torch.nn.functional.silu(ln(x))
    assert (so.ln_silu(x, ln.weight, ln.bias, ln.eps) - want).abs().max() < 1e-12
    # attention logits: sum_k c SmoothLeakyReLU(a[e,h,k]) alpha_dot[h,k]
    H, Kh = 4, 16
    a = torch.randn(13, H * Kh, generator=g, dtype=torch.float64)
    adot = torch.randn(1, H, Kh, generator=g, dtype=torch.float64)
    slr = onets.ScaledAct(onets.SmoothLeakyReLU(0.2))
    want = torch.einsum("bik,aik->bi", slr(a.view(-1, H, Kh)), adot)
    got = so.alpha_logits(a, adot, H, Kh, so3.C_SMOOTH_LEAKY_RELU_02)
    assert (got - want).abs().max() < 1e-6
    # exp-normal radial basis
    rbf = onets.ExpNormalSmearing(0.0, 5.0, 32).double()
    d = torch.rand(17, generator=g, dtype=torch.float64) * 6.0
    got = so.rbf_expnorm(d, rbf.means.double(), rbf.betas.double(), rbf.alpha, 5.0)
    assert (got - rbf(d)).abs().max() < 1e-12


def test_attention_aggregate_restatement():
    """segment softmax + weighted aggregation per head vs the oracle's softmax / vec2heads / scatter / heads2vec."""
    g = torch.Generator().manual_seed(6)
    H = 4
    irreps_head = e3.Irreps("8x0e+4x1e+4x2e")
    lay = RowLayout(e3.Irreps("32x0e+16x1e+16x2e"))
    N, E = 6, 23
    dst = torch.sort(torch.randint(0, N, (E,), generator=g)).values
    src = torch.randint(0, N, (E,), generator=g)
    graph = SimpleNamespace(N=N, E=E, src=src.int(), dst=dst.int())
    logit = torch.randn(E, H, generator=g, dtype=torch.float64, requires_grad=True)
    ve, vc = _pair(lay, E, 7)
    # oracle: value rows in e3nn layout of the simplified head irreps, split per head like Vec2AttnHeads
    alpha = onets.segment_softmax(logit, dst, N)                       # [E, H]
    vh = onets.vec2heads(ve, irreps_head, H)                           # [E, H, dim_head]
    out_h = onets.scatter_sum(vh * alpha.unsqueeze(-1), dst, N)        # [N, H, dim_head]
    want = onets.heads2vec(out_h, irreps_head)                         # [N, D] e3nn layout of 32x0e+16x1e+16x2e
    got = so.attn_aggregate(logit, vc, graph, H, lay)
    p = lay.perm_from_e3nn()
    assert (want[:, p] - got).abs().max() < 1e-12
    a = torch.randn(want.shape, generator=g, dtype=torch.float64)
    gw = torch.autograd.grad((a * want).sum(), [logit, ve], create_graph=True)
    gg = torch.autograd.grad((a[:, p] * got).sum(), [logit, vc], create_graph=True)
    assert (gw[0] - gg[0]).abs().max() < 1e-11 and (gw[1][:, p] - gg[1]).abs().max() < 1e-11
    b = torch.randn(E, H, generator=g, dtype=torch.float64)
    hw = torch.autograd.grad((b * gw[0]).sum(), [logit, ve])
    hg = torch.autograd.grad((b * gg[0]).sum(), [logit, vc])
    assert (hw[0] - hg[0]).abs().max() < 1e-10 and (hw[1][:, p] - hg[1]).abs().max() < 1e-10


def test_edge_geometry_restatement():
    g = torch.Generator().manual_seed(8)
    pos = torch.randn(9, 3, generator=g, dtype=torch.float64, requires_grad=True)
    src = torch.randint(0, 9, (20,), generator=g)
    dst = (src + 1 + torch.randint(0, 8, (20,), generator=g)) % 9
    graph = SimpleNamespace(src=src.int(), dst=dst.int())
    off = torch.randn(20, 3, generator=g, dtype=torch.float64)
    for lmax in (1, 2, 3):
        length, sh = so.edge_geometry(pos, off, graph, lmax)
        vec = pos[src] - pos[dst] + off
        want = e3.spherical_harmonics(lmax, vec, normalize=True, normalization="component")
        assert (length - vec.norm(dim=1)).abs().max() < 1e-13 and (sh - want).abs().max() < 1e-12
        a = torch.randn(sh.shape, generator=g, dtype=torch.float64)
        (g1,) = torch.autograd.grad((a * sh).sum(), pos, create_graph=True)
        (g2,) = torch.autograd.grad((a * want).sum(), pos, create_graph=True)
        assert (g1 - g2).abs().max() < 1e-11
