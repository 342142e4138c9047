"""Known-answer tests that pin the oracle's e3nn-0.4.4 conventions (KAT-1..KAT-9).
The reference ships no golden vectors, so these analytic identities are what anchors parity."""
import math

import pytest
import torch

from oracle import e3, nets

SL = [slice(0, 1), slice(1, 4), slice(4, 9), slice(9, 16)]


def test_kat1_w3j_values():
    assert abs(e3.wigner_3j(1, 1, 1)[0, 1, 2].item() - 1 / math.sqrt(6)) < 1e-12
    for l in range(4):
        eye = torch.eye(2 * l + 1, dtype=torch.float64) / math.sqrt(2 * l + 1)
        assert (e3.wigner_3j(l, l, 0)[:, :, 0] - eye).abs().max() < 1e-12
        assert (e3.wigner_3j(0, l, l)[0] - eye).abs().max() < 1e-12
    for t in [(1, 1, 1), (2, 1, 2), (3, 2, 2), (1, 2, 3)]:
        assert abs(e3.wigner_3j(*t).norm().item() - 1.0) < 1e-12


@pytest.mark.parametrize("abc,rho", [((1, 1, 0), math.sqrt(3)), ((1, 1, 2), 0.489898), ((1, 2, 1), 0.816497),
                                     ((1, 2, 3), 0.428571), ((1, 3, 2), 0.6), ((2, 2, 0), math.sqrt(5)),
                                     ((2, 2, 2), 0.534522), ((2, 3, 1), 1.0), ((2, 3, 3), 0.436436),
                                     ((3, 3, 0), math.sqrt(7)), ((3, 3, 2), 0.611010)])
def test_kat2_sh_w3j_consistency(abc, rho):
    a, b, c = abc
    g = torch.Generator().manual_seed(1)
    Y = e3.spherical_harmonics(3, torch.randn(7, 3, generator=g, dtype=torch.float64))
    r = torch.einsum("ijk,zi,zj->zk", e3.wigner_3j(a, b, c), Y[:, SL[a]], Y[:, SL[b]])
    assert (r - rho * Y[:, SL[c]]).abs().max() < 2e-6


def test_kat3_sh_norm():
    g = torch.Generator().manual_seed(2)
    Y = e3.spherical_harmonics(3, torch.randn(9, 3, generator=g, dtype=torch.float64))
    for l in range(4):
        assert (Y[:, SL[l]].pow(2).sum(-1) - (2 * l + 1)).abs().max() < 1e-12


def test_w3j_rotation_invariance():
    """C is invariant under D^l1 x D^l2 x D^l3 with D^l obtained from the SH themselves (Y(Rx) = D Y(x))."""
    g = torch.Generator().manual_seed(3)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    x = torch.randn(64, 3, generator=g, dtype=torch.float64)
    Y, YR = e3.spherical_harmonics(3, x), e3.spherical_harmonics(3, x @ q.T)
    D = [torch.linalg.lstsq(Y[:, SL[l]], YR[:, SL[l]]).solution.T for l in range(4)]
    for t in [(1, 1, 1), (1, 2, 2), (2, 2, 2), (2, 1, 3), (3, 3, 2)]:
        C = e3.wigner_3j(*t)
        Cr = torch.einsum("ia,jb,kc,abc->ijk", D[t[0]], D[t[1]], D[t[2]], C)
        assert (C - Cr).abs().max() < 1e-9


def test_normalize2mom_constants():
    assert abs(e3.normalize2mom_const(torch.nn.functional.silu) - 1.6791767924) < 1e-8
    assert abs(e3.normalize2mom_const(torch.sigmoid) - 1.8467055342) < 1e-8
    assert abs(e3.normalize2mom_const(nets.SmoothLeakyReLU(0.2)) - 1.5313204756) < 1e-8


def test_kat4_parameter_counts():
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(nets.graph_attention_transformer_nonlinear_l2("5x0e", 5.0)) == 3531715
    assert n(nets.graph_attention_transformer_nonlinear_exp_l2_md17("64x0e", 5.0, num_basis=32)) == 3496001
    assert n(nets.graph_attention_transformer_nonlinear_exp_l3_md17("64x0e", 5.0, num_basis=32)) == 5500865
    assert n(nets.oc20_l1_256_nonlinear()) == 9123331


def test_kat5_dtp_structure():
    d = nets.DepthwiseTensorProduct("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", bias=False)
    assert len(d.tp.instructions) == 15 and d.tp.weight_numel == 960
    assert repr(d.irreps_out.simplify()) == "224x0e+384x1e+352x2e"
    paths = [(d.irreps_in1[i][1].l, d.irreps_in2[j][1].l, d.irreps_out[k][1].l) for i, j, k, *_ in d.tp.instructions]
    assert paths == [(0, 0, 0), (0, 1, 1), (0, 2, 2), (1, 0, 1), (1, 1, 0), (1, 1, 1), (1, 1, 2), (1, 2, 1), (1, 2, 2),
                     (2, 0, 2), (2, 1, 1), (2, 1, 2), (2, 2, 0), (2, 2, 1), (2, 2, 2)]
    d3 = nets.DepthwiseTensorProduct("128x0e+64x1e+64x2e+32x3e", "1x0e+1x1e+1x2e+1x3e", "128x0e+64x1e+64x2e+32x3e",
                                     bias=False)
    assert len(d3.tp.instructions) == 34 and d3.tp.weight_numel == 2112
    assert repr(d3.irreps_out.simplify()) == "288x0e+576x1e+672x2e+576x3e"


def test_linear_rs_is_plain_matmul():
    torch.manual_seed(0)
    lin = nets.LinearRS(e3.Irreps("8x0e+4x1e"), e3.Irreps("6x0e+2x1e"), bias=False).double()
    x = torch.randn(5, 20, dtype=torch.float64)
    w = lin.tp.weight
    W0, W1 = w[:48].view(8, 6), w[48:].view(4, 2)
    ref = torch.cat([x[:, :8] @ W0, torch.einsum("zui,uw->zwi", x[:, 8:].view(5, 4, 3), W1).reshape(5, 6)], 1)
    assert (lin(x) - ref).abs().max() < 1e-12


def _small_qm9(**kw):
    cfg = dict(irreps_in="5x0e", irreps_node_embedding="32x0e+32x1e+32x2e", num_layers=2, irreps_sh="1x0e+1x1e+1x2e",
               max_radius=5.0, number_of_basis=16, fc_neurons=[64, 64], irreps_feature="64x0e",
               irreps_head="8x0e+8x1e+8x2e", num_heads=4, nonlinear_message=True,
               irreps_mlp_mid="64x0e+32x1e+32x2e", alpha_drop=0.0)
    cfg.update(kw)
    return cfg


def _rot(gen):
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=gen, dtype=torch.float64))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def test_kat6_kat8_equivariance_and_permutation():
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(4)
    m = nets.GraphAttentionTransformer(**_small_qm9()).double().eval()
    B, Na = 3, 9
    pos = torch.rand(B * Na, 3, generator=g, dtype=torch.float64) * 4.0
    batch = torch.arange(B).repeat_interleave(Na)
    z = torch.tensor([1, 6, 7, 8, 9])[torch.randint(0, 5, (B * Na,), generator=g)]
    y = m(None, pos, batch, z)
    R = _rot(g)
    assert (y - m(None, pos @ R.T + 0.7, batch, z)).abs().max() < 1e-10
    p = torch.cat([torch.randperm(Na, generator=g), torch.arange(Na, B * Na)])
    assert (y - m(None, pos[p], batch[p], z[p])).abs().max() < 1e-10


def test_oc20_heads_are_equivariant_and_drop_path_is_per_graph():
    """OC20 auxiliary / attention heads of the oracle [ref: graph_attention_transformer_oc20.py:182-208, :352-381]: the
    energy is rotation invariant, the per-node auxiliary output rotates as a vector; GraphDropPath [ref: nets/drop.py:
    45-61] zeroes whole graphs and rescales the kept ones by 1/keep."""
    g = torch.Generator().manual_seed(6)
    cfg = dict(irreps_node_embedding="64x0e+32x1e", num_layers=2, irreps_sh="1x0e+1x1e", max_radius=5.0,
               number_of_basis=16, fc_neurons=[16, 16], irreps_feature="64x0e+32x1e", irreps_head="16x0e+8x1e",
               num_heads=4, nonlinear_message=True, irreps_mlp_mid="64x0e+32x1e", alpha_drop=0.0)
    N = 14
    pos = torch.rand(N, 3, generator=g, dtype=torch.float64) * 4.0
    batch = torch.tensor([0] * 7 + [1] * 7)
    z = torch.randint(1, 80, (N,), generator=g)
    tags = torch.randint(0, 3, (N,), generator=g)
    R = _rot(g)
    for heads in (dict(use_auxiliary_task=True), dict(use_attention_head=True, use_auxiliary_task=True)):
        torch.manual_seed(0)
        m = nets.GraphAttentionTransformerOC20(**cfg, **heads).double().eval()
        e, a = m(z, tags, pos, batch)
        e2, a2 = m(z, tags, pos @ R.T, batch)
        assert e.shape == (2, 1) and a.shape == (N, 3) and a.abs().max() > 1e-3
        assert (e - e2).abs().max() < 1e-10 and (a2 - a @ R.T).abs().max() < 1e-10
    torch.manual_seed(0)
    m = nets.GraphAttentionTransformerOC20(**cfg, drop_path_rate=0.5).double()
    blk = m.blocks[0].train()
    x = torch.randn(N, 5, generator=g, dtype=torch.float64)
    torch.manual_seed(3)
    y = blk._drop_path(x, batch)
    torch.manual_seed(3)
    keep = torch.floor(0.5 + torch.rand((2, 1), dtype=torch.float64))
    assert torch.equal(y, x * (keep / 0.5)[batch])
    assert torch.equal(blk.eval()._drop_path(x, batch), x)


def test_other_families_o3_and_permutation_invariances():
    """Oracle restatements added in round 2: dot-product attention [ref: nets/dp_attention_transformer.py], E(3) irreps
    (the `_e3` factories) and DeNS [ref: nets/equiformer_md17_dens.py] -- energies invariant under rotations, translations
    and atom permutations; the E(3) model also under INVERSION; forces / denoising vectors rotate (and flip under
    inversion); the DeNS force encoding rotates with its input."""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(8)
    B, Na = 2, 8
    pos = torch.rand(B * Na, 3, generator=g, dtype=torch.float64) * 3.5
    batch = torch.arange(B).repeat_interleave(Na)
    z_q = torch.tensor([1, 6, 7, 8, 9])[torch.randint(0, 5, (B * Na,), generator=g)]
    z_m = torch.randint(1, 9, (B * Na,), generator=g)
    R = _rot(g)
    perm = torch.cat([torch.randperm(Na, generator=g), torch.arange(Na, B * Na)])
    base = dict(num_layers=2, max_radius=5.0, number_of_basis=16, fc_neurons=[16, 16], irreps_feature="32x0e", num_heads=2)
    so3 = dict(irreps_node_embedding="16x0e+8x1e+4x2e", irreps_sh="1x0e+1x1e+1x2e", irreps_head="8x0e+4x1e+2x2e",
               irreps_mlp_mid="16x0e+8x1e+4x2e")
    e3 = dict(irreps_node_embedding="16x0e+4x0o+4x1e+4x1o+2x2e+2x2o", irreps_sh="1x0e+1x1o+1x2e",
              irreps_head="8x0e+2x0o+2x1e+2x1o+2x2e+2x2o", irreps_mlp_mid="16x0e+4x0o+4x1e+4x1o+2x2e+2x2o")
    # --- dot-product attention, energy model
    torch.manual_seed(0)
    m = nets.DotProductAttentionTransformer(irreps_in="5x0e", alpha_drop=0.0, **base, **so3).double().eval()
    y = m(None, pos, batch, z_q)
    assert (y - m(None, pos @ R.T + 0.3, batch, z_q)).abs().max() < 1e-10
    assert (y - m(None, pos[perm], batch[perm], z_q[perm])).abs().max() < 1e-10
    # --- E(3) irreps: rotation AND inversion; forces are polar vectors
    torch.manual_seed(0)
    m = nets.GraphAttentionTransformerMD17(irreps_in="64x0e", basis_type="exp", nonlinear_message=True, alpha_drop=0.0,
                                           **base, **e3).double().eval()
    E, F = m(z_m, pos.clone(), batch)
    E2, F2 = m(z_m, (pos @ R.T).clone(), batch)
    E3, F3 = m(z_m, (-pos).clone(), batch)
    assert (E - E2).abs().max() < 1e-10 and (F2 - F @ R.T).abs().max() < 1e-9
    assert (E - E3).abs().max() < 1e-10 and (F3 + F).abs().max() < 1e-9
    assert F.abs().max() > 1e-4
    # the SO(3)-flavoured model (all-even harmonics) is NOT inversion invariant: parity is what the E(3) variant adds
    torch.manual_seed(0)
    ms = nets.GraphAttentionTransformerMD17(irreps_in="64x0e", basis_type="exp", nonlinear_message=True, alpha_drop=0.0,
                                            **base, **so3).double().eval()
    Es, _ = ms(z_m, pos.clone(), batch)
    Ei, _ = ms(z_m, (-pos).clone(), batch)
    assert (Es - Ei).abs().max() > 1e-8
    # --- DeNS: energy invariant, output vectors (forces on clean atoms, denoising vectors on corrupted ones) rotate
    torch.manual_seed(0)
    md = nets.Equiformer_MD17_DeNS(irreps_feature="32x0e+16x1e+8x2e", irreps_pre_attn="16x0e+8x1e+4x2e",
                                   irreps_equivariant_inputs="1x0e+1x1e+1x2e", nonlinear_message=True, alpha_drop=0.0,
                                   **{k: v for k, v in base.items() if k != "irreps_feature"}, **so3).double().eval()
    force = torch.randn(B * Na, 3, generator=g, dtype=torch.float64)
    mask = torch.rand(B * Na, generator=g) < 0.4
    assert mask.any() and (~mask).any()
    e, dy = md(SimpleNamespace(z=z_m, pos=pos.clone(), batch=batch, force=force, noise_mask=mask))
    e2, dy2 = md(SimpleNamespace(z=z_m, pos=(pos @ R.T).clone(), batch=batch, force=force @ R.T, noise_mask=mask))
    assert (e - e2).abs().max() < 1e-10 and (dy2 - dy @ R.T).abs().max() < 1e-9
    e3_, _ = md(SimpleNamespace(z=z_m, pos=pos.clone(), batch=batch, force=2.0 * force, noise_mask=mask))
    assert (e - e3_).abs().max() > 1e-8  # the encoded forces do reach the energy


def test_kat6_kat7_md17_forces():
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(5)
    cfg = _small_qm9(irreps_in="64x0e", basis_type="exp")
    m = nets.GraphAttentionTransformerMD17(**cfg).double().eval()
    B, Na = 2, 8
    pos = torch.rand(B * Na, 3, generator=g, dtype=torch.float64) * 3.5
    batch = torch.arange(B).repeat_interleave(Na)
    z = torch.randint(1, 9, (B * Na,), generator=g)
    E, F = m(z, pos.clone(), batch)
    R = _rot(g)
    E2, F2 = m(z, (pos @ R.T).clone(), batch)
    assert (E - E2).abs().max() < 1e-10 and (F @ R.T - F2).abs().max() < 1e-9
    eps = 1e-5
    for (i, d) in [(0, 0), (5, 1), (11, 2)]:
        pp, pm = pos.clone(), pos.clone()
        pp[i, d] += eps
        pm[i, d] -= eps
        fd = -((m(z, pp, batch)[0] - m(z, pm, batch)[0]).sum() / (2 * eps)).item()
        assert abs(fd - F[i, d].item()) < 1e-6 * max(1.0, abs(fd))


def test_kat9_segment_softmax():
    g = torch.Generator().manual_seed(6)
    x = torch.randn(12, 4, generator=g, dtype=torch.float64)
    idx = torch.tensor([0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2])
    a = nets.segment_softmax(x, idx, 3)
    assert (nets.scatter_sum(a, idx, 3) - 1).abs().max() < 1e-12
    assert (a[:4] - torch.softmax(x[:4], 0)).abs().max() < 1e-12


def test_radius_graph_order():
    g = torch.Generator().manual_seed(7)
    pos = torch.rand(14, 3, generator=g) * 4
    batch = torch.tensor([0] * 6 + [1] * 8)
    src, dst = nets.radius_graph(pos, 2.5, batch)
    assert (dst[1:] >= dst[:-1]).all() and (batch[src] == batch[dst]).all() and (src != dst).all()
    same = dst[1:] == dst[:-1]
    assert (src[1:][same] > src[:-1][same]).all()


# ---------------------------------------------------------------------------- periodic neighbour search (oracle/pbc.py)
def test_pbc_lattice_coordination_numbers():
    """Known answers: coordination shells of the simple-cubic and bcc lattices."""
    from oracle import pbc
    a = 3.0
    cell = (torch.eye(3) * a)[None]
    pos = torch.tensor([[0.3, 0.1, 0.7]])
    for r, n in ((3.1, 6), (4.3, 18), (5.25, 26), (6.05, 32)):  # shells at a, a sqrt2, a sqrt3, 2a
        ei, off, nb = pbc.radius_graph_pbc(pos, cell, [1], r, 1000)
        assert ei.shape[1] == n == int(nb[0]) and (ei == 0).all()
        _, dist, _ = pbc.get_pbc_distances(pos.double(), ei, cell.double(), off, nb)
        assert dist.max() <= r and dist.min() >= a - 1e-9
    # bcc: 8 neighbours at a sqrt3 / 2 = 2.598, 6 at a
    pos2 = torch.tensor([[0.0, 0.0, 0.0], [1.5, 1.5, 1.5]])
    ei, off, nb = pbc.radius_graph_pbc(pos2, cell, [2], 2.7, 1000)
    assert ei.shape[1] == 16 and (ei[0] != ei[1]).all()
    ei, off, nb = pbc.radius_graph_pbc(pos2, cell, [2], 3.05, 1000)
    assert ei.shape[1] == 28
    # exactly-on-the-cut-off pairs are kept (<=), unlike torch_cluster's strict <
    ei, _, _ = pbc.radius_graph_pbc(pos, cell, [1], 3.0, 1000)
    assert ei.shape[1] == 6


def test_pbc_triclinic_matches_wide_enumeration_and_truncation():
    from oracle import pbc
    g = torch.Generator().manual_seed(11)
    cell = torch.tensor([[[6.0, 0.0, 0.0], [1.5, 5.5, 0.0], [0.8, -1.1, 7.0]],
                         [[4.0, 0.0, 0.0], [0.0, 9.0, 0.0], [0.0, 0.0, 12.0]]])
    natoms = [7, 5]
    frac = torch.rand(12, 3, generator=g)
    pos = torch.cat([frac[:7] @ cell[0], frac[7:] @ cell[1]])
    r = 5.0
    ei, off, nb = pbc.radius_graph_pbc(pos, cell, natoms, r, 1000)
    # independent enumeration over a fixed, generous image range
    want = set()
    start = 0
    for b, n in enumerate(natoms):
        rng = range(-4, 5)
        for i in range(n):
            for j in range(n):
                for u in rng:
                    for v in rng:
                        for w in rng:
                            d = pos[start + j] + torch.tensor([u, v, w], dtype=torch.float32) @ cell[b] - pos[start + i]
                            d2 = float((d.double() ** 2).sum())
                            if 1e-4 < d2 <= r * r:
                                want.add((start + j, start + i, u, v, w))
        start += n
    got = {(int(ei[0, e]), int(ei[1, e]), *[int(x) for x in off[e]]) for e in range(ei.shape[1])}
    assert got == want and int(nb.sum()) == len(want)
    assert (ei[1][1:] >= ei[1][:-1]).all()
    # truncation keeps the nearest max_neighbors of every centre
    k = 9
    ei2, off2, nb2 = pbc.radius_graph_pbc(pos, cell, natoms, r, k)
    _, dist, _ = pbc.get_pbc_distances(pos.double(), ei, cell.double(), off, nb)
    _, dist2, _ = pbc.get_pbc_distances(pos.double(), ei2, cell.double(), off2, nb2)
    for i in range(12):
        full = torch.sort(dist[ei[1] == i]).values
        kept = torch.sort(dist2[ei2[1] == i]).values
        assert kept.numel() == min(k, full.numel()) and torch.allclose(kept, full[:kept.numel()])
