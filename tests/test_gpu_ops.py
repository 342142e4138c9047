"""Per-operator parity of the HIP kernels (through the C ABI) against the CPU oracle / plain torch fp64 math.
Tolerances are fp32-roundoff class: 1e-5 relative to the output scale unless stated."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import e3 as oe3
from oracle import nets as onets


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _grads(out, inputs, seed=0):
    g = torch.Generator().manual_seed(seed)
    go = torch.randn(out.shape, generator=g, dtype=torch.float64)
    return go, torch.autograd.grad(out, inputs, go.to(out.dtype).to(out.device), allow_unused=True)


def _cf(x, layout):
    return x[:, layout.perm_from_e3nn().to(x.device)].contiguous()


def _e3(x, layout):
    return x[:, layout.perm_to_e3nn().to(x.device)].contiguous()


def test_library_loads_on_gpu(hip_lib):
    from equiformer_amd import lib
    assert "gfx950" in lib.version()


# ------------------------------------------------------------------------------------------------- GEMMs
# tolerance factor per matrix mode: "fp32" = the exact-fp32 MFMA kernels (csrc/gemm.hip), "split" = the default arithmetic
# (csrc/gemmx.hip: fp32 operands as 2 + 3 bf16 planes on the bf16 matrix cores; measured 2-5e-6 of the result scale)
MODE_TOL = {"fp32": 1.0, "split": 4.0}


@pytest.mark.parametrize("mode", sorted(MODE_TOL))
@pytest.mark.parametrize("M,N,K", [(1000, 128, 128), (777, 64, 96), (513, 32, 32), (2000, 224, 352), (130, 960, 64),
                                   (64, 1, 512), (300, 5, 7)])
def test_dense_linear(M, N, K, mode):
    from equiformer_amd import ops
    tol = MODE_TOL[mode]
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g, dtype=torch.float64)
    W = torch.randn(N, K, generator=g, dtype=torch.float64) / math.sqrt(K)
    b = torch.randn(N, generator=g, dtype=torch.float64)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    ref = xr @ Wr.T + br
    go, gref = _grads(ref, [xr, Wr, br])
    xg, Wg, bg = (t.float().to(dev).requires_grad_(True) for t in (x, W, b))
    with ops.matrix_mode(mode):
        out = ops.dense_linear(xg, Wg, bg)
        assert _rel(out, ref) < 2e-6 * tol
        gout = torch.autograd.grad(out, [xg, Wg, bg], go.float().to(dev))
    for a, r in zip(gout, gref):
        assert _rel(a, r) < 5e-6 * tol


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric B catches row/col swaps in the MFMA C mapping."""
    from equiformer_amd import ops
    dev = _dev()
    n = 128
    B = (torch.arange(n * n, dtype=torch.float32).view(n, n) % 97) * 0.5 + torch.arange(n).float()[:, None]
    out = ops.dense_linear(torch.eye(n, device=dev), B.T.contiguous().to(dev), None)
    assert torch.equal(out.cpu(), B)


@pytest.mark.parametrize("irr_in,irr_out", [("128x0e+64x1e+32x2e", "128x0e+64x1e+32x2e"),
                                             ("128x0e+64x1e+32x2e", "672x0e+192x1e+96x2e"),
                                             ("224x0e+384x1e+352x2e", "128x0e"),
                                             ("128x0e+64x1e+32x2e", "512x0e"),
                                             ("128x0e+64x1e+64x2e+32x3e", "128x0e+64x1e+64x2e+32x3e")])
@pytest.mark.parametrize("mode", sorted(MODE_TOL))
def test_irreps_linear(irr_in, irr_out, mode):
    from equiformer_amd import ops
    from equiformer_amd.nets.layers import LinearRS
    tol = MODE_TOL[mode]
    dev = _dev()
    torch.manual_seed(1)
    ref = onets.LinearRS(oe3.Irreps(irr_in), oe3.Irreps(irr_out)).double()
    for b in ref.bias:
        b.data.normal_()
    mod = LinearRS(irr_in, irr_out)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(dev)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(301, mod.layout_in.dim, generator=g, dtype=torch.float64).requires_grad_(True)
    yr = ref(x)
    go, gref = _grads(yr, [x] + list(ref.parameters()))
    xg = _cf(x.detach().float().to(dev), mod.layout_in).requires_grad_(True)
    with ops.matrix_mode(mode):
        y = mod(xg)
        assert _rel(_e3(y, mod.layout_out), yr) < 3e-6 * tol
        gout = torch.autograd.grad(y, [xg] + list(mod.parameters()), _cf(go.float().to(dev), mod.layout_out))
    assert _rel(_e3(gout[0], mod.layout_in), gref[0]) < 5e-6 * tol
    names = [n for n, _ in mod.named_parameters()]
    rnames = [n for n, _ in ref.named_parameters()]
    assert names == rnames
    for a, r in zip(gout[1:], gref[1:]):
        assert _rel(a, r) < 1e-5 * tol


# ------------------------------------------------------------------------------------------------- row ops
@pytest.mark.parametrize("irr", ["128x0e+64x1e+32x2e", "512x0e", "128x0e+64x1e+64x2e+32x3e", "256x0e+128x1e"])
def test_layer_norm(irr):
    from equiformer_amd.nets.layers import EquivariantLayerNormV2
    dev = _dev()
    ref = onets.EquivariantLayerNormV2(oe3.Irreps(irr)).double()
    g = torch.Generator().manual_seed(3)
    ref.affine_weight.data = torch.randn(ref.affine_weight.shape, generator=g, dtype=torch.float64)
    ref.affine_bias.data = torch.randn(ref.affine_bias.shape, generator=g, dtype=torch.float64)
    mod = EquivariantLayerNormV2(irr)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(dev)
    x = (torch.randn(257, mod.layout.dim, generator=g, dtype=torch.float64) * 2 + 0.3).requires_grad_(True)
    yr = ref(x)
    go, gref = _grads(yr, [x, ref.affine_weight, ref.affine_bias])
    xg = _cf(x.detach().float().to(dev), mod.layout).requires_grad_(True)
    y = mod(xg)
    assert _rel(_e3(y, mod.layout), yr) < 5e-6
    gout = torch.autograd.grad(y, [xg, mod.affine_weight, mod.affine_bias], _cf(go.float().to(dev), mod.layout))
    assert _rel(_e3(gout[0], mod.layout), gref[0]) < 2e-5
    assert _rel(gout[1], gref[1]) < 2e-5 and _rel(gout[2], gref[2]) < 2e-5


@pytest.mark.parametrize("irr", ["128x0e+64x1e+32x2e", "384x0e+192x1e+96x2e", "384x0e+192x1e+192x2e+96x3e"])
def test_gate(irr):
    from equiformer_amd.layout import RowLayout
    from equiformer_amd.nets.layers import make_gate
    dev = _dev()
    ref = onets.make_gate(oe3.Irreps(irr))
    mod = make_gate(irr)
    lin_layout, lout = RowLayout(mod.irreps_in), RowLayout(mod.irreps_out)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(203, lin_layout.dim, generator=g, dtype=torch.float64).requires_grad_(True)
    yr = ref(x)
    go, gref = _grads(yr, [x])
    xg = _cf(x.detach().float().to(dev), lin_layout).requires_grad_(True)
    y = mod(xg)
    assert _rel(_e3(y, lout), yr) < 3e-6
    (gx,) = torch.autograd.grad(y, [xg], _cf(go.float().to(dev), lout))
    assert _rel(_e3(gx, lin_layout), gref[0]) < 5e-6


def test_scaled_silu_and_lnsilu():
    from equiformer_amd import ops, so3
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1001, 64, generator=g, dtype=torch.float64).requires_grad_(True)
    yr = torch.nn.functional.silu(x) * so3.C_SILU
    go, gref = _grads(yr, [x])
    xg = x.detach().float().to(dev).requires_grad_(True)
    y = ops.scaled_silu(xg, so3.C_SILU)
    assert _rel(y, yr) < 2e-6
    assert _rel(torch.autograd.grad(y, [xg], go.float().to(dev))[0], gref[0]) < 5e-6
    gam = torch.randn(64, generator=g, dtype=torch.float64).requires_grad_(True)
    bet = torch.randn(64, generator=g, dtype=torch.float64).requires_grad_(True)
    yr = torch.nn.functional.silu(torch.nn.functional.layer_norm(x, (64,), gam, bet, 1e-5))
    go, gref = _grads(yr, [x, gam, bet])
    gg, bg = gam.detach().float().to(dev).requires_grad_(True), bet.detach().float().to(dev).requires_grad_(True)
    y = ops.ln_silu(xg, gg, bg, 1e-5)
    assert _rel(y, yr) < 5e-6
    for a, r in zip(torch.autograd.grad(y, [xg, gg, bg], go.float().to(dev)), gref):
        assert _rel(a, r) < 2e-5


@pytest.mark.parametrize("mode", sorted(MODE_TOL))
def test_radial_profile(mode):
    from equiformer_amd.nets.layers import RadialProfile
    dev = _dev()
    torch.manual_seed(6)
    ref = onets.RadialProfile([128, 64, 64, 960]).double()
    mod = RadialProfile([128, 64, 64, 960])
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(dev)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(515, 128, generator=g, dtype=torch.float64).requires_grad_(True)
    yr = ref(x)
    go, gref = _grads(yr, [x] + list(ref.parameters()))
    xg = x.detach().float().to(dev).requires_grad_(True)
    from equiformer_amd import ops
    with ops.matrix_mode(mode):
        y = mod(xg)
        gout = torch.autograd.grad(y, [xg] + list(mod.parameters()), go.float().to(dev))
    assert _rel(y, yr) < 5e-6 * MODE_TOL[mode]
    for a, r in zip(gout, gref):
        assert _rel(a, r) < 3e-5 * MODE_TOL[mode]


def test_embedding():
    from equiformer_amd.nets.layers import NodeEmbeddingNetwork
    dev = _dev()
    torch.manual_seed(8)
    irr = "128x0e+64x1e+32x2e"
    ref = onets.NodeEmbeddingNetwork(oe3.Irreps(irr), 5).double()
    ref.atom_type_lin.bias[0].data.normal_()
    mod = NodeEmbeddingNetwork(irr, 5)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(dev)
    z = torch.tensor([0, 4, 2, 2, 1, 3, 0, 0, 4])
    yr, _, _ = ref(z)
    go, gref = _grads(yr, list(ref.parameters()))
    y, _, _ = mod(z.to(dev))
    lay = mod.atom_type_lin.layout_out
    assert _rel(_e3(y, lay), yr) < 1e-6
    gout = torch.autograd.grad(y, list(mod.parameters()), _cf(go.float().to(dev), lay))
    for a, r in zip(gout, gref):
        assert _rel(a, r) < 1e-5


# ------------------------------------------------------------------------------------------------- graph + geometry
def _mol_batch(B=5, Na=13, side=4.5, seed=0):
    from equiformer_amd.synthetic import qm9_like_batch
    return qm9_like_batch(B, Na, side=side, seed=seed)


def test_radius_graph_matches_oracle():
    from equiformer_amd.graph import EdgeGraph
    dev = _dev()
    d = _mol_batch(7, 18, 5.5, 1)
    g = EdgeGraph.from_radius(d["pos"].to(dev), d["batch"].to(dev), 5.0)
    src, dst = onets.radius_graph(d["pos"], 5.0, d["batch"])
    assert g.E == src.numel()
    assert torch.equal(g.src.cpu().long(), src) and torch.equal(g.dst.cpu().long(), dst)
    rp = g.row_ptr.cpu().long()
    assert torch.equal(rp[1:] - rp[:-1], torch.bincount(dst, minlength=g.N))
    # by-source view
    perm = g.src_perm.cpu().long()
    assert torch.equal(torch.sort(perm).values, torch.arange(g.E))
    sp = g.src_ptr.cpu().long()
    for n in [0, 3, g.N - 1]:
        assert (src[perm[sp[n]:sp[n + 1]]] == n).all()
    # max_num_neighbors truncation keeps the first sources in index order
    g2 = EdgeGraph.from_radius(d["pos"].to(dev), d["batch"].to(dev), 5.0, max_num_neighbors=4)
    s2, d2 = onets.radius_graph(d["pos"], 5.0, d["batch"], max_num_neighbors=4)
    assert torch.equal(g2.src.cpu().long(), s2) and torch.equal(g2.dst.cpu().long(), d2)
    # CSR bookkeeping kernels (segment offsets, exclusive scan, by-source permutation) are bit-exact against the
    # stable sort / bincount they replace -- also on the truncated (asymmetric) graph and on ragged molecule sizes
    for gg, ss in ((g, src), (g2, s2)):
        assert torch.equal(gg.src_perm.cpu().long(), torch.argsort(ss, stable=True))
        cnt = torch.bincount(ss, minlength=gg.N)
        assert torch.equal(gg.src_ptr.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]))
        assert torch.equal(gg.mol_ptr.cpu().long(), torch.arange(0, 7 * 18 + 1, 18))


def test_csr_bookkeeping_ragged():
    from equiformer_amd.graph import EdgeGraph
    dev = _dev()
    g0 = torch.Generator().manual_seed(5)
    sizes = [1, 7, 300, 2, 33, 1, 64]  # single-atom molecules have no edges; one molecule larger than a workgroup
    batch = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(sizes)])
    pos = torch.cat([torch.rand(n, 3, generator=g0) * (2.0 + n ** (1 / 3.0) * 1.5) for n in sizes])
    g = EdgeGraph.from_radius(pos.to(dev), batch.to(dev), 3.0)
    src, dst = onets.radius_graph(pos, 3.0, batch)
    assert g.E == src.numel() and g.E > 0
    assert torch.equal(g.src.cpu().long(), src) and torch.equal(g.dst.cpu().long(), dst)
    assert torch.equal(g.src_perm.cpu().long(), torch.argsort(src, stable=True))
    N = sum(sizes)
    z = torch.zeros(1, dtype=torch.long)
    assert torch.equal(g.src_ptr.cpu().long(), torch.cat([z, torch.bincount(src, minlength=N).cumsum(0)]))
    assert torch.equal(g.row_ptr.cpu().long(), torch.cat([z, torch.bincount(dst, minlength=N).cumsum(0)]))
    assert torch.equal(g.mol_ptr.cpu().long(), torch.cat([z, torch.tensor(sizes).cumsum(0)]))
    # empty trailing molecule ids (num_graphs larger than the last id + 1)
    g3 = EdgeGraph.from_radius(pos.to(dev), batch.to(dev), 3.0, num_graphs=len(sizes) + 2)
    assert torch.equal(g3.src_perm.cpu(), g.src_perm.cpu()) and g3.mol_ptr.cpu().tolist()[-3:] == [N, N, N]


@pytest.mark.parametrize("lmax", [1, 2, 3])
def test_edge_geometry_and_rbf(lmax):
    from equiformer_amd import ops
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.nets.layers import ExpNormalSmearing, GaussianRadialBasisLayer
    dev = _dev()
    d = _mol_batch(4, 12, 4.0, 2)
    g = EdgeGraph.from_radius(d["pos"].to(dev), d["batch"].to(dev), 5.0)
    src, dst = g.src.cpu().long(), g.dst.cpu().long()
    pos = d["pos"].double().requires_grad_(True)
    vec = pos[src] - pos[dst]
    sh_r = oe3.spherical_harmonics(lmax, vec)
    len_r = vec.norm(dim=1)
    torch.manual_seed(9)
    rg = onets.GaussianRadialBasisLayer(32, 5.0).double()
    re_ = onets.ExpNormalSmearing(0.0, 5.0, 32).double()
    out_r = (sh_r * 0.7).sum(1) + (rg(len_r) * 0.3).sum(1) + (re_(len_r) * 1.1).sum(1)
    go, gref = _grads(out_r, [pos] + list(rg.parameters()))
    pg = d["pos"].to(dev).requires_grad_(True)
    _, length, sh = ops.edge_geometry(pg, None, g, lmax)
    mg = GaussianRadialBasisLayer(32, 5.0)
    mg.load_state_dict({k: v.float() for k, v in rg.state_dict().items()})
    mg = mg.to(dev)
    me = ExpNormalSmearing(0.0, 5.0, 32).to(dev)
    assert _rel(sh, sh_r) < 2e-6 and _rel(length, len_r) < 1e-6
    assert _rel(mg(length), rg(len_r)) < 1e-5 and _rel(me(length), re_(len_r)) < 1e-5
    out = (sh * 0.7).sum(1) + (mg(length) * 0.3).sum(1) + (me(length) * 1.1).sum(1)
    gout = torch.autograd.grad(out, [pg] + list(mg.parameters()), go.float().to(dev))
    assert _rel(gout[0], gref[0]) < 5e-5
    for a, r in zip(gout[1:], gref[1:]):
        assert _rel(a, r) < 5e-5


def test_gather_and_segment_sum():
    from equiformer_amd import ops
    from equiformer_amd.graph import EdgeGraph
    dev = _dev()
    d = _mol_batch(6, 11, 4.0, 3)
    g = EdgeGraph.from_radius(d["pos"].to(dev), d["batch"].to(dev), 5.0)
    src, dst = g.src.cpu().long(), g.dst.cpu().long()
    gen = torch.Generator().manual_seed(10)
    a = torch.randn(g.N, 480, generator=gen, dtype=torch.float64).requires_grad_(True)
    b = torch.randn(g.N, 480, generator=gen, dtype=torch.float64).requires_grad_(True)
    ref = a[src] + b[dst]
    pooled_r = onets.scatter_sum(ref, dst, g.N) * 0.25
    go, gref = _grads(pooled_r, [a, b])
    ag, bg = (t.detach().float().to(dev).requires_grad_(True) for t in (a, b))
    msg = ops.gather_add(ag, bg, g)
    assert _rel(msg, ref) < 1e-6
    pooled = ops.segment_sum(msg, g.row_ptr, g.dst, g.N, 0.25)
    assert _rel(pooled, pooled_r) < 2e-6
    for x, r in zip(torch.autograd.grad(pooled, [ag, bg], go.float().to(dev)), gref):
        assert _rel(x, r) < 3e-6
    # node -> molecule pooling with a single column
    x = torch.randn(g.N, 1, generator=gen, dtype=torch.float64)
    pr = onets.scatter_sum(x, d["batch"], g.num_graphs)
    p = ops.segment_sum(x.float().to(dev), g.mol_ptr, g.batch, g.num_graphs, 1.0)
    assert _rel(p, pr) < 2e-6


# ------------------------------------------------------------------------------------------------- DTP
def _dtp_setup(irr, sh_irr, E=257, seed=11):
    from equiformer_amd.layout import DtpTable
    torch.manual_seed(seed)
    ref = onets.DepthwiseTensorProduct(irr, sh_irr, irr, bias=False).double()
    table = DtpTable(irr, sh_irr, irr)
    lmax = len(oe3.Irreps(sh_irr)) - 1
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(E, table.layout_in.dim, generator=g, dtype=torch.float64).requires_grad_(True)
    vec = torch.randn(E, 3, generator=g, dtype=torch.float64)
    sh = oe3.spherical_harmonics(lmax, vec).requires_grad_(True)
    w = torch.randn(E, table.weight_numel, generator=g, dtype=torch.float64).requires_grad_(True)
    return ref, table, x, sh, w


@pytest.mark.parametrize("irr,sh_irr", [("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e"),
                                        ("128x0e+64x1e+64x2e+32x3e", "1x0e+1x1e+1x2e+1x3e"),
                                        ("256x0e+128x1e", "1x0e+1x1e"),
                                        ("16x0e+8x1e+8x2e", "1x0e+1x1e+1x2e")])
def test_dtp_unfused(irr, sh_irr):
    from equiformer_amd import ops
    from equiformer_amd.layout import RowLayout
    dev = _dev()
    ref, table, x, sh, w = _dtp_setup(irr, sh_irr)
    assert table.weight_numel == ref.tp.weight_numel
    assert repr(table.irreps_out) == repr(ref.irreps_out.simplify())
    yr = ref(x, sh, w)
    go, gref = _grads(yr, [x, sh, w])
    lay_mid = RowLayout(table.irreps_out)
    xg = _cf(x.detach().float().to(dev), table.layout_in).requires_grad_(True)
    shg = sh.detach().float().to(dev).requires_grad_(True)
    wg = w.detach().float().to(dev).requires_grad_(True)
    M = ops.dtp_coupling(shg, table)
    y = ops.dtp(xg, M, wg, table)
    assert _rel(_e3(y, lay_mid), yr) < 5e-6
    gout = torch.autograd.grad(y, [xg, shg, wg], _cf(go.float().to(dev), lay_mid))
    assert _rel(_e3(gout[0], table.layout_in), gref[0]) < 1e-5
    assert _rel(gout[1], gref[1]) < 2e-5
    assert _rel(gout[2], gref[2]) < 1e-5


@pytest.mark.parametrize("irr,sh_irr,use_act", [("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", True),
                                                ("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", False),
                                                ("128x0e+64x1e+64x2e+32x3e", "1x0e+1x1e+1x2e+1x3e", True),
                                                ("256x0e+128x1e", "1x0e+1x1e", True)])
def test_separable_fctp_fused_and_unfused(irr, sh_irr, use_act):
    """SeparableFCTP (radial MLP -> DTP -> linear [-> gate]) against the oracle, fused MFMA path and un-fused path;
    use_act=False exercises the shared-internal-weight form (sep_value)."""
    from equiformer_amd import ops
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.layout import RowLayout
    from equiformer_amd.nets.layers import EdgeContext, SeparableFCTP
    dev = _dev()
    torch.manual_seed(12)
    fc = [16, 64, 64] if use_act else None
    ref = onets.SeparableFCTP(irr, sh_irr, irr, fc, use_activation=use_act, internal_weights=not use_act).double()
    mod = SeparableFCTP(irr, sh_irr, irr, fc, use_activation=use_act, internal_weights=not use_act)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(dev)
    lmax = len(oe3.Irreps(sh_irr)) - 1
    E = 301
    g = torch.Generator().manual_seed(13)
    lay_in, lay_out = RowLayout(irr), RowLayout(irr)
    x = torch.randn(E, lay_in.dim, generator=g, dtype=torch.float64).requires_grad_(True)
    sh = oe3.spherical_harmonics(lmax, torch.randn(E, 3, generator=g, dtype=torch.float64)).requires_grad_(True)
    es = torch.rand(E, 16, generator=g, dtype=torch.float64).requires_grad_(True)
    yr = ref(x, sh, es if use_act else None)
    go, gref = _grads(yr, [x, sh] + ([es] if use_act else []) + list(ref.parameters()))
    for fused in (True, "legacy", False):
        xg = _cf(x.detach().float().to(dev), lay_in).requires_grad_(True)
        shg = sh.detach().float().to(dev).requires_grad_(True)
        esg = es.detach().float().to(dev).requires_grad_(True)
        ectx = EdgeContext(None, shg, esg)
        y = mod(xg, ectx, use_fused=fused)
        assert _rel(_e3(y, lay_out), yr) < 1e-5, fused
        ins = [xg, shg] + ([esg] if use_act else []) + list(mod.parameters())
        gout = torch.autograd.grad(y, ins, _cf(go.float().to(dev), lay_out))
        assert _rel(_e3(gout[0], lay_in), gref[0]) < 3e-5, fused
        for i, (a, r) in enumerate(zip(gout[1:], gref[1:])):
            assert _rel(a, r) < 5e-5, (fused, i)


@pytest.mark.parametrize("irr,sh_irr,n2,E", [("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", 128, 333),
                                             ("128x0e+64x1e+64x2e+32x3e", "1x0e+1x1e+1x2e+1x3e", 128, 130),
                                             ("256x0e+128x1e", "1x0e+1x1e", 256, 77),
                                             ("64x0e+32x1e+32x2e", "1x0e+1x1e+1x2e", 64, 1)])
def test_sfc_two_consumers_matches_unfused(irr, sh_irr, n2, E):
    """eqf_sfc_* with the second scalar consumer (attention logits) against the un-fused composition
    DTP -> {per-degree linear, scalar linear}: outputs and every gradient (x, sh, w, weights, bias)."""
    from equiformer_amd import ops
    from equiformer_amd.layout import DtpTable, RowLayout
    dev = _dev()
    table = DtpTable(irr, sh_irr, irr)
    lay_out = RowLayout(irr)
    lay_mid = table.layout_out
    lmax = len(oe3.Irreps(sh_irr)) - 1
    g = torch.Generator().manual_seed(5)
    x = torch.randn(E, table.layout_in.dim, generator=g).to(dev).requires_grad_(True)
    sh = oe3.spherical_harmonics(lmax, torch.randn(E, 3, generator=g, dtype=torch.float64)).float().to(dev).requires_grad_(True)
    w = torch.randn(E, table.weight_numel, generator=g).to(dev).requires_grad_(True)
    spec = ops.SfcSpec(table, lay_out, n2=n2)
    assert spec.supported
    weight = (torch.randn(spec.weight_numel, generator=g) / 16).to(dev).requires_grad_(True)
    weight2 = (torch.randn(spec.weight2_numel, generator=g) / 16).to(dev).requires_grad_(True)
    n1_0 = lay_out.mul_of(0)
    bias = torch.randn(n1_0, generator=g).to(dev).requires_grad_(True)
    bias2 = torch.randn(n2, generator=g).to(dev).requires_grad_(True)
    M = ops.dtp_coupling(sh, table)
    o1, o2 = ops.sep_fctp(x, M, w, weight, bias, spec, weight2=weight2, bias2=bias2)
    # reference composition on the GPU from the un-fused primitives
    mid = ops.dtp(x, ops.dtp_coupling(sh, table), w, table)
    lin_spec = ops.LinearSpec(lay_mid, lay_out)
    r1 = ops.irreps_linear(mid, weight, bias, lin_spec)
    K0 = spec.degs[0][1]
    r2 = mid[:, :K0] @ weight2.view(K0, n2) + bias2
    assert _rel(o1, r1) < 1e-5 and _rel(o2, r2) < 1e-5
    g1 = torch.randn(o1.shape, generator=g).to(dev)
    g2 = torch.randn(o2.shape, generator=g).to(dev)
    ins = [x, sh, w, bias, bias2, weight, weight2]
    ga = torch.autograd.grad([o1, o2], ins, [g1, g2])
    gb = torch.autograd.grad([r1, r2], ins, [g1, g2])
    for i, (a, b) in enumerate(zip(ga, gb)):
        assert _rel(a, b) < 3e-5, i


# ------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("head_irr,H", [("32x0e+16x1e+8x2e", 4), ("32x0e+16x1e+16x2e+8x3e", 4), ("32x0e+16x1e", 8)])
def test_alpha_and_attention_aggregate(head_irr, H):
    from equiformer_amd import ops, so3
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.layout import RowLayout
    dev = _dev()
    d = _mol_batch(5, 14, 4.0, 4)
    g = EdgeGraph.from_radius(d["pos"].to(dev), d["batch"].to(dev), 5.0)
    dst = g.dst.cpu().long()
    oh = oe3.Irreps(head_irr)
    heads_all = onets.sort_irreps_even_first(oh * H)[0].simplify()
    lay = RowLayout(repr(heads_all))
    gen = torch.Generator().manual_seed(14)
    a = torch.randn(g.E, H * 32, generator=gen, dtype=torch.float64).requires_grad_(True)
    adot = torch.randn(1, H, 32, generator=gen, dtype=torch.float64).requires_grad_(True)
    v = torch.randn(g.E, lay.dim, generator=gen, dtype=torch.float64).requires_grad_(True)
    act = onets.SmoothLeakyReLU(0.2)
    al = act(a.view(g.E, H, 32)) * oe3.normalize2mom_const(act)
    logit_r = torch.einsum("bik,aik->bi", al, adot)
    alpha_r = onets.segment_softmax(logit_r, dst, g.N).unsqueeze(-1)
    vh = onets.vec2heads(v, oh, H)
    out_r = onets.heads2vec(onets.scatter_sum(vh * alpha_r, dst, g.N), oh)
    go, gref = _grads(out_r, [a, adot, v])
    ag = a.detach().float().to(dev).requires_grad_(True)
    dg = adot.detach().float().to(dev).requires_grad_(True)
    vg = _cf(v.detach().float().to(dev), lay).requires_grad_(True)
    logit = ops.alpha_logits(ag, dg, H, 32, so3.C_SMOOTH_LEAKY_RELU_02)
    assert _rel(logit, logit_r) < 5e-6
    out = ops.attn_aggregate(logit, vg, g, H, lay, 0.0, 0)
    assert _rel(_e3(out, lay), out_r) < 1e-5
    gout = torch.autograd.grad(out, [ag, dg, vg], _cf(go.float().to(dev), lay))
    assert _rel(gout[0], gref[0]) < 3e-5
    assert _rel(gout[1], gref[1]) < 3e-5
    assert _rel(_e3(gout[2], lay), gref[2]) < 1e-5


def test_attention_aggregate_long_rows():
    """Destination rows longer than a wavefront (one dense cluster of 160 atoms: ~159 edges per row): the forward keeps
    the first 64 logits of a row in registers and re-reads the rest, the value rows are requested one iteration ahead."""
    from equiformer_amd import ops
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.layout import RowLayout
    dev = _dev()
    gen = torch.Generator().manual_seed(3)
    pos = torch.rand(160, 3, generator=gen) * 2.0
    batch = torch.zeros(160, dtype=torch.long)
    g = EdgeGraph.from_radius(pos.to(dev), batch.to(dev), 5.0)
    assert g.E == 160 * 159
    H, oh = 4, oe3.Irreps("32x0e+16x1e+8x2e")
    lay = RowLayout(repr(onets.sort_irreps_even_first(oh * H)[0].simplify()))
    dst = g.dst.cpu().long()
    logit_r = 3.0 * torch.randn(g.E, H, generator=gen, dtype=torch.float64)
    v = torch.randn(g.E, lay.dim, generator=gen, dtype=torch.float64)
    alpha_r = onets.segment_softmax(logit_r, dst, g.N).unsqueeze(-1)
    out_r = onets.heads2vec(onets.scatter_sum(onets.vec2heads(v, oh, H) * alpha_r, dst, g.N), oh)
    out = ops.attn_aggregate(logit_r.float().to(dev), _cf(v.float().to(dev), lay), g, H, lay, 0.0, 0)
    assert _rel(_e3(out, lay), out_r) < 1e-5


def test_attention_dropout_statistics():
    from equiformer_amd import ops
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.layout import RowLayout
    dev = _dev()
    d = _mol_batch(64, 18, 5.0, 5)
    g = EdgeGraph.from_radius(d["pos"].to(dev), d["batch"].to(dev), 5.0)
    lay = RowLayout("128x0e+64x1e+32x2e")
    logit = torch.zeros(g.E, 4, device=dev)
    v = torch.ones(g.E, lay.dim, device=dev, requires_grad=True)
    out0 = ops.attn_aggregate(logit, v, g, 4, lay, 0.0, 0)
    out = ops.attn_aggregate(logit, v, g, 4, lay, 0.2, 1234)
    assert abs(out0.mean().item() - 1.0) < 1e-5          # uniform attention over ones
    assert abs(out.mean().item() - 1.0) < 0.02            # inverted dropout keeps the mean
    assert (out - out0).abs().max() > 0.05                # and actually drops something
    out_again = ops.attn_aggregate(logit, v, g, 4, lay, 0.2, 1234)
    assert torch.equal(out, out_again)                    # counter-based mask: same seed, same mask
    (gv,) = torch.autograd.grad(out.sum(), [v])           # backward regenerates the same mask
    col = gv[:, 0]
    kept = (col > 0).float().mean().item()
    assert abs(kept - 0.8) < 0.02


@pytest.mark.parametrize("max_nbr", [1000, 9])
def test_radius_graph_pbc_matches_oracle(max_nbr):
    """HIP periodic neighbour search vs the CPU restatement of ocpmodels' radius_graph_pbc (oracle/pbc.py): identical
    edge sets (neighbour, centre, image), dst-sorted rows, Cartesian offsets = image . cell; with truncation the same
    nearest-k survivors."""
    from equiformer_amd.graph import EdgeGraph
    from oracle import pbc
    dev = _dev()
    g0 = torch.Generator().manual_seed(21)
    cell = torch.tensor([[[6.0, 0.0, 0.0], [1.5, 5.5, 0.0], [0.8, -1.1, 7.0]],
                         [[4.0, 0.0, 0.0], [0.0, 9.0, 0.0], [0.0, 0.0, 12.0]],
                         [[11.0, 0.0, 0.0], [0.0, 11.0, 0.0], [0.0, 0.0, 25.0]]])
    natoms = [7, 5, 30]
    frac = torch.rand(sum(natoms), 3, generator=g0)
    pos = torch.cat([frac[:7] @ cell[0], frac[7:12] @ cell[1], frac[12:] @ cell[2]])
    batch = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(natoms)])
    r = 5.0
    ei, off, nb = pbc.radius_graph_pbc(pos, cell, natoms, r, max_nbr)
    g, offsets, cell_off = EdgeGraph.from_radius_pbc(pos.to(dev), cell.to(dev), batch.to(dev), r,
                                                     max_num_neighbors=max_nbr)
    assert g.E == ei.shape[1]
    got = torch.cat([g.src.cpu().long()[:, None], g.dst.cpu().long()[:, None], cell_off.cpu().long()], dim=1)
    want = torch.cat([ei[0][:, None], ei[1][:, None], off], dim=1)
    assert torch.equal(got, want)  # same order as well: by centre, neighbour, image
    cpe = torch.repeat_interleave(cell, nb, dim=0)
    cart = torch.bmm(off.float().view(-1, 1, 3), cpe).view(-1, 3)
    assert (offsets.cpu() - cart).abs().max() < 1e-5
    rp = g.row_ptr.cpu().long()
    assert torch.equal(rp[1:] - rp[:-1], torch.bincount(ei[1], minlength=sum(natoms)))
    perm = g.src_perm.cpu().long()
    assert torch.equal(perm, torch.argsort(ei[0], stable=True))
