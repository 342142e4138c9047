"""OC20 heads beyond the plain energy MLP: auxiliary IS2RS head, attention head + skip connection, the scalar-channel
energy head on an l>0 feature, and per-graph stochastic depth -- HIP model against the fp64 CPU oracle with identical
weights and inputs [ref: nets/graph_attention_transformer_oc20.py:169-208, :337-381; nets/drop.py:45-61].
Bar: 1e-4 relative (BASELINE.json north_star)."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets as onets

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as mg  # noqa: E402  (reduced configurations)
from weights import fill_deterministic  # noqa: E402


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _slab(B, Na, seed):
    """Random atoms in an orthorhombic cell, periodic along x and y: explicit edge list + Cartesian offsets."""
    g = torch.Generator().manual_seed(seed)
    cell = torch.tensor([8.0, 8.0, 30.0])
    pos = torch.rand(B * Na, 3, generator=g) * torch.tensor([8.0, 8.0, 10.0])
    batch = torch.arange(B).repeat_interleave(Na)
    Z = torch.randint(1, 84, (B * Na,), generator=g)
    tags = torch.randint(0, 3, (B * Na,), generator=g)
    src, dst, off = [], [], []
    for b in range(B):
        idx = torch.arange(b * Na, (b + 1) * Na)
        for sx in (-1, 0, 1):
            for sy in (-1, 0, 1):
                shift = torch.tensor([sx * cell[0], sy * cell[1], 0.0])
                dist = (pos[idx][:, None, :] + shift - pos[idx][None, :, :]).norm(dim=-1)
                s, t = ((dist < 5.0) & (dist > 1e-6)).nonzero(as_tuple=True)
                src.append(idx[s]); dst.append(idx[t]); off.append(shift.expand(s.numel(), 3))
    return pos, batch, Z, tags, torch.stack([torch.cat(src), torch.cat(dst)]), torch.cat(off)


def _models(cfg, seed=21):
    from equiformer_amd.nets.graph_attention_transformer_oc20 import GraphAttentionTransformerOC20
    ref = fill_deterministic(onets.GraphAttentionTransformerOC20(**cfg), seed).double()
    mod = GraphAttentionTransformerOC20(None, None, 1, **cfg)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
    return ref, mod.to(_dev())


def _run(ref, mod, inp):
    dev = _dev()
    pos, batch, Z, tags, ei, off = inp
    out_r = ref(Z, tags, pos.double(), batch, edge_index=ei, offsets=off.double())
    data = SimpleNamespace(pos=pos.to(dev), batch=batch.to(dev), atomic_numbers=Z.to(dev), tags=tags.to(dev),
                           edge_index=ei.to(dev), offsets=off.to(dev))
    return out_r, mod(data)


def _loss(out, te, ta):
    if isinstance(out, tuple):
        return (out[0].squeeze() - te.to(out[0])).abs().mean() + 0.5 * (out[1] - ta.to(out[1])).abs().mean()
    return (out.squeeze() - te.to(out)).abs().mean()


HEADS = {
    "aux": dict(use_auxiliary_task=True),
    "attn": dict(use_attention_head=True),
    "attn_aux": dict(use_attention_head=True, use_auxiliary_task=True),
    "aux_l1_feature": dict(use_auxiliary_task=True, irreps_feature="64x0e+32x1e"),
    "aux_linear_message": dict(use_auxiliary_task=True, nonlinear_message=False),
    "l1_feature_energy_only": dict(irreps_feature="64x0e+32x1e"),
}


@pytest.mark.parametrize("kind", sorted(HEADS))
def test_oc20_heads_forward_backward(kind):
    cfg = dict(mg.SMALL_OC20, number_of_basis=32, **HEADS[kind])
    ref, mod = _models(cfg)
    ref.eval(); mod.eval()
    inp = _slab(2, 24, seed=7)
    out_r, out = _run(ref, mod, inp)
    assert isinstance(out, tuple) == isinstance(out_r, tuple)
    pairs = list(zip(out, out_r)) if isinstance(out, tuple) else [(out, out_r)]
    for a, r in pairs:
        assert a.shape == r.shape
        assert _rel(a, r) < 1e-4, (kind, _rel(a, r))
    g = torch.Generator().manual_seed(1)
    te, ta = torch.randn(2, generator=g), torch.randn(48, 3, generator=g)
    gr = torch.autograd.grad(_loss(out_r, te, ta), list(ref.parameters()), allow_unused=True)
    gg = torch.autograd.grad(_loss(out, te, ta), list(mod.parameters()), allow_unused=True)
    worst, scale = 0.0, max(r.abs().max().item() for r in gr if r is not None)
    for (n, _), a, r in zip(ref.named_parameters(), gg, gr):
        if r is None or r.abs().max() == 0:
            assert a is None or a.abs().max().item() < 1e-6 * scale, n
            continue
        assert a is not None, n
        # per-parameter error relative to that gradient's own size, floored at 1e-3 of the largest gradient
        e = (a.double().cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-3 * scale)
        worst = max(worst, e)
        assert e < 2e-4, (kind, n, e)
    print("%s: outputs rel %s, worst parameter-gradient rel %.2e" % (kind, ["%.1e" % _rel(a, r) for a, r in pairs], worst))


def test_oc20_aux_config_full_width():
    """The shipped auxiliary-task configuration (l1_256_nonlinear_aux_g@2_local.yml: feature 512x0e+256x1e, IS2RS head,
    drop_path 0.05), eval mode, random-init weights shared with the oracle."""
    from equiformer_amd import nets
    torch.manual_seed(0)
    over = dict(irreps_feature="512x0e+256x1e", use_auxiliary_task=True, drop_path_rate=0.05, num_layers=3)
    ref = onets.oc20_l1_256_nonlinear(**over).double().eval()
    mod = nets.model_entrypoint("oc20_l1_256_nonlinear_aux")(num_layers=3, otf_graph=False)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
    mod = mod.to(_dev()).eval()
    (er, ar), (e, a) = _run(ref, mod, _slab(2, 40, seed=3))
    print("oc20 aux config: energy rel %.3e, auxiliary vectors rel %.3e" % (_rel(e, er), _rel(a, ar)))
    assert _rel(e, er) < 1e-4 and _rel(a, ar) < 1e-4


def test_drop_path_training_matches_oracle():
    """Stochastic depth in training mode: both sides draw the per-graph keep mask from torch's CPU generator (fp64), so
    the same seed gives the same dropped branches; alpha_drop = 0 keeps the stream free of other draws."""
    cfg = dict(mg.SMALL_OC20, number_of_basis=32, drop_path_rate=0.4, use_auxiliary_task=True)
    ref, mod = _models(cfg)
    inp = _slab(4, 16, seed=9)
    ref.eval(); mod.eval()
    _, (e_eval, _) = _run(ref, mod, inp)
    ref.train(); mod.train()
    torch.manual_seed(5)
    out_r = ref(inp[2], inp[3], inp[0].double(), inp[1], edge_index=inp[4], offsets=inp[5].double())
    torch.manual_seed(5)
    dev = _dev()
    data = SimpleNamespace(pos=inp[0].to(dev), batch=inp[1].to(dev), atomic_numbers=inp[2].to(dev),
                           tags=inp[3].to(dev), edge_index=inp[4].to(dev), offsets=inp[5].to(dev))
    out = mod(data)
    assert _rel(out[0], out_r[0]) < 1e-4 and _rel(out[1], out_r[1]) < 1e-4
    assert _rel(out[0], e_eval) > 1e-3  # some branch was really dropped
    p_r = ref.blocks[0].ga.sep_act.lin.tp.weight
    p = mod.blocks[0].ga.sep_act.lin.tp.weight
    (gr,) = torch.autograd.grad(out_r[0].sum() + out_r[1].sum(), [p_r])
    (gg,) = torch.autograd.grad(out[0].sum() + out[1].sum(), [p])
    assert _rel(gg, gr) < 2e-4


def test_segment_scale_op():
    from equiformer_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 64, generator=g).to(dev).requires_grad_(True)
    seg = torch.sort(torch.randint(0, 5, (37,), generator=g)).values.to(torch.int32).to(dev)
    s = torch.tensor([0.0, 2.0, 1.0, 0.0, 1.25], device=dev)
    y = ops.segment_scale(x, s, seg)
    want = x.detach() * s[seg.long()][:, None]
    assert torch.equal(y.detach(), want)
    dy = torch.randn(37, 64, generator=g).to(dev)
    (dx,) = torch.autograd.grad(y, x, dy)
    assert torch.equal(dx, dy * s[seg.long()][:, None])


E3_FEATURE = dict(irreps_node_embedding="32x0e+16x0o+16x1e+16x1o", irreps_sh="1x0e+1x1o", irreps_head="8x0e+4x0o+4x1e+4x1o",
                  irreps_mlp_mid="64x0e+16x0o+32x1e+16x1o", num_layers=4)


@pytest.mark.parametrize("feature", ["64x0e+16x1e+16x1o", "64x0e+16x0o+16x1e+16x1o"])
@pytest.mark.parametrize("head", ["aux", "attn", "attn_aux"])
def test_oc20_heads_on_e3_feature(head, feature):
    """Auxiliary / attention heads on an E(3) feature [ref: nets/graph_attention_transformer_oc20.py:184-186: the auxiliary
    head has 1x1o output irreps when the feature carries 1o channels; :196-208 attention head], incl. a feature that lacks
    some (degree, parity) segments.  Refused in round 3 after an 11 % error; root cause (round 4, tools/e3_head_bisect.py):
    the per-degree linear tested `l == 0` for "carries the bias", which also matches the 0o segment -- the 0e bias was added
    to the pseudo-scalar outputs.  Four blocks, filled (non-zero) biases: pseudo-scalar contamination reaches the energy."""
    cfg = dict(mg.SMALL_OC20, number_of_basis=32, irreps_feature=feature, **E3_FEATURE, **HEADS[head])
    ref, mod = _models(cfg)
    ref.eval(); mod.eval()
    inp = _slab(2, 24, seed=3)
    out_r, out = _run(ref, mod, inp)
    if not isinstance(out_r, tuple):
        out_r, out = (out_r,), (out,)
    errs = [_rel(a, b) for a, b in zip(out, out_r)]
    print("oc20 %s head on %s: rel err %s" % (head, feature, ["%.2e" % e for e in errs]))
    assert len(out) == len(out_r) and all(e < 1e-4 for e in errs), errs
    if len(out) == 2:
        assert out[1].shape[1] == 3
    g = torch.Generator().manual_seed(1)
    te, ta = torch.randn(2, generator=g), torch.randn(48, 3, generator=g)
    names = [n for n, _ in ref.named_parameters()]
    gr = torch.autograd.grad(_loss(out_r if len(out_r) > 1 else out_r[0], te.double(), ta.double()), list(ref.parameters()),
                             allow_unused=True)
    gg = torch.autograd.grad(_loss(out if len(out) > 1 else out[0], te, ta), list(mod.parameters()), allow_unused=True)
    scale = max(r.abs().max().item() for r in gr if r is not None)
    worst = ("", 0.0)
    for n, a, r in zip(names, gg, gr):
        if r is None or r.abs().max() == 0:
            continue
        assert a is not None, n
        e = (a.double().cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-3 * scale)
        worst = max(worst, (n, e), key=lambda t: t[1])
    print("   worst parameter gradient %s %.2e" % worst)
    assert worst[1] < 2e-4, worst
