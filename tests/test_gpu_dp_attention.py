"""Dot-product attention family (nets/dp_attention_transformer{,_md17,_oc20}.py of the reference): the two new HIP
operators against a plain torch restatement in the reference's [mul][2l+1] layout (values, gradients, second-order
gradients), and the models against the fp64 CPU oracle with identical weights.  Bar: 1e-4 relative (BASELINE.json)."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets as onets

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as mg  # noqa: E402
from weights import fill_deterministic  # noqa: E402


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# ------------------------------------------------------------------------------------------------ operators
def _vec2heads(x, irreps_head, H):
    """[rows, e3nn layout of (irreps_head * H) sorted+simplified] -> [rows, H, dim(irreps_head)]
    [ref: Vec2AttnHeads, nets/graph_attention_transformer.py:252-280]"""
    out, ix = [], 0
    for mul, l in irreps_head:
        w = H * mul * (2 * l + 1)
        out.append(x[:, ix:ix + w].reshape(x.shape[0], H, -1))
        ix += w
    return torch.cat(out, dim=2)


def _ref_logits(q_e3, kv_e3, dst, irreps_head, H):
    """The reference's q / k / v handling on e3nn-ordered rows, fp64 [ref: nets/dp_attention_transformer.py:131-146]."""
    num = sum(m for m, _ in irreps_head)
    scale = torch.cat([torch.full((m * (2 * l + 1),), 1.0 / (num ** 0.5 * (2 * l + 1) ** 0.5), dtype=torch.float64)
                       for m, l in irreps_head])
    q = _vec2heads(q_e3, irreps_head, H) * scale
    kv = _vec2heads(kv_e3, irreps_head, 2 * H)
    k, v = kv[:, :H], kv[:, H:]
    return torch.einsum("bik,bik->bi", q[dst], k), k, v


@pytest.mark.parametrize("head,H", [([(32, 0), (16, 1), (8, 2)], 4), ([(8, 0), (4, 1), (4, 2), (4, 3)], 2),
                                    ([(32, 0), (16, 1)], 8)])
def test_kv_split_and_dp_logits_against_torch(head, H):
    from equiformer_amd import ops
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.layout import RowLayout
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    lay = RowLayout("+".join("%dx%de" % (m * H, l) for m, l in head))
    lay2 = RowLayout("+".join("%dx%de" % (2 * m * H, l) for m, l in head))
    N, E = 23, 301
    dst = torch.sort(torch.randint(0, N - 2, (E,), generator=g)).values  # the last two nodes receive no edge
    src = torch.randint(0, N, (E,), generator=g)
    graph, order = EdgeGraph.from_edges(src.to(dev), dst.to(dev), N, torch.zeros(N, dtype=torch.long, device=dev))
    assert torch.equal(order.cpu(), torch.arange(E))  # already dst-sorted: the stable sort keeps the edge order
    dst_s = graph.dst.long().cpu()
    q_e3 = torch.randn(N, lay.dim, generator=g, dtype=torch.float64).requires_grad_(True)
    kv_e3 = torch.randn(E, lay2.dim, generator=g, dtype=torch.float64).requires_grad_(True)
    logit_r, k_r, v_r = _ref_logits(q_e3, kv_e3, dst_s, head, H)
    # device side: the same rows in the channel-fastest layout
    q = q_e3.detach().float()[:, lay.perm_from_e3nn()].to(dev).requires_grad_(True)
    kv = kv_e3.detach().float()[:, lay2.perm_from_e3nn()].to(dev).requires_grad_(True)
    k, v = ops.kv_split(kv, H, lay)
    # split: bit-exact copies of the reference's narrow()
    to_e3 = lay.perm_to_e3nn().to(dev)
    hv = lambda t: _vec2heads(t[:, to_e3].double().cpu(), head, H)  # noqa: E731
    assert torch.equal(hv(k.detach()), k_r.detach().float().double())
    assert torch.equal(hv(v.detach()), v_r.detach().float().double())
    logit = ops.dp_logits(q, k, graph, H, lay)
    assert logit.shape == (E, H) and _rel(logit, logit_r) < 1e-5
    # first order, against fp64 autograd of the restatement; v gets its own cotangent so the merge is exercised
    cl = torch.randn(E, H, generator=g, dtype=torch.float64)
    cv = torch.randn(E, H, v_r.shape[2], generator=g, dtype=torch.float64)
    gq_r, gkv_r = torch.autograd.grad((logit_r * cl).sum() + (v_r * cv).sum(), [q_e3, kv_e3], create_graph=True)
    # cotangent of v in the device layout: heads -> e3nn vector -> CF
    cv_vec, ix = [], 0
    for m, l in head:
        w = m * (2 * l + 1)
        cv_vec.append(cv[:, :, ix:ix + w].reshape(E, -1))
        ix += w
    cv_dev = torch.cat(cv_vec, 1).float()[:, lay.perm_from_e3nn()].to(dev)
    gq, gkv = torch.autograd.grad((logit * cl.float().to(dev)).sum() + (v * cv_dev).sum(), [q, kv], create_graph=True)
    assert _rel(gq[:, to_e3], gq_r) < 1e-5
    assert _rel(gkv[:, lay2.perm_to_e3nn().to(dev)], gkv_r) < 1e-5
    assert gq[N - 2:].abs().max().item() == 0.0  # nodes without incoming edges
    # second order: gradient of <a, dq> + <b, dkv> w.r.t. (q, kv)
    a = torch.randn(N, lay.dim, generator=g, dtype=torch.float64)
    b = torch.randn(E, lay2.dim, generator=g, dtype=torch.float64)
    hq_r, hkv_r = torch.autograd.grad((gq_r * a).sum() + (gkv_r * b).sum(), [q_e3, kv_e3])
    a_dev = a.float()[:, lay.perm_from_e3nn()].to(dev)
    b_dev = b.float()[:, lay2.perm_from_e3nn()].to(dev)
    hq, hkv = torch.autograd.grad((gq * a_dev).sum() + (gkv * b_dev).sum(), [q, kv])
    assert _rel(hq[:, to_e3], hq_r) < 1e-5
    assert _rel(hkv[:, lay2.perm_to_e3nn().to(dev)], hkv_r) < 1e-5


# ------------------------------------------------------------------------------------------------ models
def _dp_cfg(small):
    kw = dict(getattr(mg, small))
    kw.pop("nonlinear_message", None)
    return kw


def _grad_check(ref, mod, loss_r, loss, tol):
    gr = torch.autograd.grad(loss_r, list(ref.parameters()), allow_unused=True)
    gg = torch.autograd.grad(loss, list(mod.parameters()), allow_unused=True)
    scale = max(r.abs().max().item() for r in gr if r is not None)
    worst = ("", 0.0)
    for (n, _), a, r in zip(ref.named_parameters(), gg, gr):
        if r is None or r.abs().max() == 0:
            continue
        assert a is not None, n
        e = (a.double().cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-3 * scale)
        if e > worst[1]:
            worst = (n, e)
    assert worst[1] < tol, worst
    return worst


def test_dp_qm9_forward_backward_parity():
    from equiformer_amd.nets.dp_attention_transformer import DotProductAttentionTransformer
    from equiformer_amd.synthetic import qm9_like_batch
    dev = _dev()
    kw = dict(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **_dp_cfg("SMALL_L2"))
    ref = fill_deterministic(onets.DotProductAttentionTransformer(**kw), 31).double().eval()
    mod = DotProductAttentionTransformer(**kw)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
    mod = mod.to(dev).eval()
    d = qm9_like_batch(6, 18, side=6.5, seed=2)
    yr = ref(None, d["pos"].double(), d["batch"], d["z"])
    y = mod(None, d["pos"].to(dev), d["batch"].to(dev), d["z"].to(dev))
    assert y.shape == (6, 1) and _rel(y, yr) < 1e-4
    for flag in ("legacy", False):  # the slower tensor-product paths give the same answer
        mod.set_fused(flag)
        assert _rel(mod(None, d["pos"].to(dev), d["batch"].to(dev), d["z"].to(dev)), yr) < 1e-4
    mod.set_fused(True)
    worst = _grad_check(ref, mod, (yr.squeeze() - d["y"].double()).abs().mean(),
                        (y.squeeze() - d["y"].to(dev)).abs().mean(), 2e-4)
    print("dp qm9: energy rel %.2e, worst parameter gradient %s %.2e" % (_rel(y, yr), *worst))


def test_dp_qm9_full_width_forward():
    """dot_product_attention_transformer_l2 as registered (3.35 M parameters), random init shared with the oracle."""
    from equiformer_amd import nets
    from equiformer_amd.synthetic import qm9_like_batch
    dev = _dev()
    torch.manual_seed(0)
    ref = onets.DotProductAttentionTransformer(
        irreps_in="5x0e", irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_sh="1x0e+1x1e+1x2e",
        max_radius=5.0, number_of_basis=128, fc_neurons=[64, 64], irreps_feature="512x0e",
        irreps_head="32x0e+16x1e+8x2e", num_heads=4, nonlinear_message=False, irreps_mlp_mid="384x0e+192x1e+96x2e",
        alpha_drop=0.2).double().eval()
    mod = nets.model_entrypoint("dot_product_attention_transformer_l2")("5x0e", 5.0)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
    mod = mod.to(dev).eval()
    d = qm9_like_batch(8, 18, side=6.5, seed=0)
    with torch.no_grad():
        yr = ref(None, d["pos"].double(), d["batch"], d["z"])
        y = mod(None, d["pos"].to(dev), d["batch"].to(dev), d["z"].to(dev))
    print("dp l2 full width: energy rel %.3e" % _rel(y, yr))
    assert _rel(y, yr) < 1e-4
    mod.train()  # attention dropout on: runs, is finite, differs from eval
    torch.manual_seed(1)
    yt = mod(None, d["pos"].to(dev), d["batch"].to(dev), d["z"].to(dev))
    yt.sum().backward()
    assert torch.isfinite(yt).all() and _rel(yt, y) > 1e-6
    assert all(torch.isfinite(p.grad).all() for p in mod.parameters() if p.grad is not None)


@pytest.mark.parametrize("small", ["SMALL_L2", "SMALL_L3"])
def test_dp_md17_forces_and_force_loss_gradients(small):
    """dot_product_attention_transformer_exp_l{2,3}_md17 on reduced widths: energies, forces and the second-order
    gradients of L = <a, E> + <B, F> (create_graph forces, main_md17.py:384-390) against the fp64 oracle."""
    from equiformer_amd.nets.dp_attention_transformer import DotProductAttentionTransformerMD17
    from equiformer_amd.synthetic import md17_aspirin_batch
    dev = _dev()
    kw = dict(irreps_in="64x0e", max_radius=5.0, number_of_basis=32, basis_type="exp", **_dp_cfg(small))
    ref = fill_deterministic(onets.DotProductAttentionTransformerMD17(**kw), 32).double().train()
    mod = fill_deterministic(DotProductAttentionTransformerMD17(**kw), 32).to(dev).train()
    d = md17_aspirin_batch(2, seed=3)
    g = torch.Generator().manual_seed(1)
    a = torch.randn(2, 1, generator=g, dtype=torch.float64)
    B = torch.randn(42, 3, generator=g, dtype=torch.float64)
    Er, Fr = ref(d["z"], d["pos"].double(), d["batch"])
    E, F = mod(d["z"].to(dev), d["pos"].to(dev), d["batch"].to(dev))
    assert F.requires_grad
    assert _rel(E, Er) < 1e-4 and _rel(F, Fr) < 1e-4
    worst = _grad_check(ref, mod, (a * Er).sum() + (B * Fr).sum(),
                        (a.float().to(dev) * E).sum() + (B.float().to(dev) * F).sum(), 2e-4)
    print("dp md17 %s: E rel %.2e, F rel %.2e, worst second-order gradient %s %.2e"
          % (small, _rel(E, Er), _rel(F, Fr), *worst))
    mod.eval()
    E2, F2 = mod(d["z"].to(dev), d["pos"].to(dev), d["batch"].to(dev))
    assert not F2.requires_grad and _rel(F2, Fr) < 1e-4


def test_dp_oc20_with_auxiliary_head():
    from test_gpu_oc20_heads import _slab
    from equiformer_amd.nets.dp_attention_transformer import DotProductAttentionTransformerOC20
    dev = _dev()
    cfg = dict(mg.SMALL_OC20, number_of_basis=32, use_auxiliary_task=True, irreps_pre_attn="64x0e+32x1e")
    cfg.pop("nonlinear_message")
    ref = fill_deterministic(onets.DotProductAttentionTransformerOC20(**cfg), 33).double().eval()
    mod = DotProductAttentionTransformerOC20(None, None, 1, **cfg)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
    mod = mod.to(dev).eval()
    pos, batch, Z, tags, ei, off = _slab(2, 24, seed=7)
    er, ar = ref(Z, tags, pos.double(), batch, edge_index=ei, offsets=off.double())
    data = SimpleNamespace(pos=pos.to(dev), batch=batch.to(dev), atomic_numbers=Z.to(dev), tags=tags.to(dev),
                           edge_index=ei.to(dev), offsets=off.to(dev))
    e, a = mod(data)
    assert _rel(e, er) < 1e-4 and _rel(a, ar) < 1e-4
    g = torch.Generator().manual_seed(1)
    te, ta = torch.randn(2, generator=g), torch.randn(48, 3, generator=g)
    worst = _grad_check(ref, mod, (er.squeeze() - te.double()).abs().mean() + (ar - ta.double()).abs().mean(),
                        (e.squeeze() - te.to(dev)).abs().mean() + (a - ta.to(dev)).abs().mean(), 2e-4)
    print("dp oc20 + aux: energy rel %.2e, aux rel %.2e, worst gradient %s %.2e" % (_rel(e, er), _rel(a, ar), *worst))
