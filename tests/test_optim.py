"""Optimizer side of the step: the oracle restatement against torch's own AdamW / clip_grad_norm_ (CPU), and the fused
HIP optimizer against the oracle (GPU)."""
import numpy as np
import pytest
import torch

from oracle import optim as ooptim


def _toy(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(7, 5), (13,), (3, 4, 2), (1,), (129,)]
    ps = [torch.randn(*s, generator=g, dtype=torch.float64) for s in shapes]
    grads = [[torch.randn(*s, generator=g, dtype=torch.float64) * (3.0 if k == 1 else 0.1) for s in shapes]
             for k in range(4)]
    return ps, grads


def test_oracle_matches_torch_adamw_and_clip():
    ps0, grads = _toy()
    wds = [0.0, 0.0, 5e-3, 5e-3, 5e-3]
    tp = [torch.nn.Parameter(p.clone()) for p in ps0]
    opt = torch.optim.AdamW([{"params": tp[:2], "weight_decay": 0.0}, {"params": tp[2:], "weight_decay": 5e-3}],
                            lr=5e-4, betas=(0.9, 0.999), eps=1e-8)
    p = [x.numpy().copy() for x in ps0]
    m = [np.zeros_like(x) for x in p]
    v = [np.zeros_like(x) for x in p]
    for step, gs in enumerate(grads, 1):
        for t, g in zip(tp, gs):
            t.grad = g.clone()
        total = torch.nn.utils.clip_grad_norm_(tp, 1.5)
        opt.step()
        coef, tot = ooptim.clip_coef([g.numpy() for g in gs], 1.5)
        assert abs(tot - float(total)) < 1e-9 * max(1.0, tot)
        for i in range(len(p)):
            p[i], m[i], v[i] = ooptim.adamw_step(p[i], gs[i].numpy() * coef, m[i], v[i], step, 5e-4, 0.9, 0.999, 1e-8,
                                                 wds[i])
            assert np.abs(p[i] - tp[i].detach().numpy()).max() < 1e-12
    assert any(ooptim.clip_coef([g.numpy() for g in gs], 1.5)[0] < 1.0 for gs in grads)


def test_add_weight_decay_name_rules():
    """Same groups as optim_factory.add_weight_decay (reference :27-42) on parameter names of the real model tree."""
    from equiformer_amd.optim import add_weight_decay

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3, 3)
            self.affine_weight = torch.nn.Parameter(torch.ones(3))
            self.norm = torch.nn.Module()
            self.norm.affine_weight = torch.nn.Parameter(torch.ones(3))
            self.norm.affine_bias = torch.nn.Parameter(torch.ones(3))
            self.norm.mean_shift = torch.nn.Parameter(torch.ones(3))
            self.bias = torch.nn.ParameterList([torch.nn.Parameter(torch.ones(2))])
            self.tp = torch.nn.Module()
            self.tp.bias = torch.nn.ParameterList([torch.nn.Parameter(torch.ones(2))])
            self.alpha_dot = torch.nn.Parameter(torch.ones(2))
            self.frozen = torch.nn.Parameter(torch.ones(2), requires_grad=False)

    m = M()
    nd, d = add_weight_decay(m, 0.01, skip_list={"alpha_dot"})
    names = {id(p): n for n, p in m.named_parameters()}
    got_nd = sorted(names[id(p)] for p in nd["params"])
    got_d = sorted(names[id(p)] for p in d["params"])
    assert got_nd == sorted(["lin.bias", "norm.affine_weight", "norm.affine_bias", "norm.mean_shift", "bias.0",
                             "tp.bias.0", "alpha_dot"])
    assert got_d == sorted(["lin.weight", "affine_weight"])  # a top-level 'affine_weight' has no '.' prefix: decays
    assert nd["weight_decay"] == 0.0 and d["weight_decay"] == 0.01


def test_flat_adamw_state_dict_round_trip_keeps_flat_aliasing():
    """Host logic only (no step): parameters / moments alias the flat buffers, state_dict has torch.optim.AdamW's
    layout, load_state_dict copies INTO the flat buffers."""
    from equiformer_amd.optim import FlatAdamW

    class HostOnly(FlatAdamW):  # the GPU-only guard is the single thing a CPU test has to bypass
        @staticmethod
        def _check_params(ps):
            pass

    def make(seed):
        g = torch.Generator().manual_seed(seed)
        ps = [torch.nn.Parameter(torch.randn(3, 4, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g)),
              torch.nn.Parameter(torch.randn(2, 2, generator=g))]
        return ps, HostOnly([{"params": ps[:1], "weight_decay": 0.0}, {"params": ps[1:], "weight_decay": 0.01}],
                            lr=1e-3, ema_decay=0.9)

    ps, a = make(0)
    before = [p.detach().clone() for p in ps]
    assert all(p.data_ptr() >= a.flat_p.data_ptr() and p.data_ptr() < a.flat_p.data_ptr() + 4 * a.n for p in ps)
    assert all(torch.equal(p.detach(), b) for p, b in zip(ps, before))
    # every parameter's slice starts 256-byte aligned (64 floats): offsets 0, 64, 128; the padding is zero
    assert a.offsets == [0, 64, 128] and a.n == 192
    want_wd = torch.zeros(192)
    want_wd[64:69], want_wd[128:132] = 0.01, 0.01
    assert torch.equal(a.flat_wd, want_wd)
    assert all(p.data_ptr() % 16 == 0 for p in ps)
    a.flat_m.copy_(torch.arange(a.n, dtype=torch.float32))
    a.flat_v.copy_(torch.arange(a.n, dtype=torch.float32) * 2)
    a._step = 7
    for p in ps:
        a.state[p]["step"] = 7
    sd = a.state_dict()
    assert sd["state"][1]["exp_avg"].tolist() == list(range(64, 69)) and sd["state"][2]["step"] == 7
    assert [len(g["params"]) for g in sd["param_groups"]] == [1, 2]
    ps2, b = make(1)
    sd["param_groups"][0]["lr"] = sd["param_groups"][1]["lr"] = 5e-4
    b.load_state_dict(sd)
    # the parameters' slices are restored (the test wrote into the padding of `a` as well; `b`'s padding stays zero)
    for o, k in zip(a.offsets, a.sizes):
        assert torch.equal(b.flat_m[o:o + k], a.flat_m[o:o + k]) and torch.equal(b.flat_v[o:o + k], a.flat_v[o:o + k])
    assert b._step == 7
    assert b.param_groups[0]["lr"] == 5e-4 and b.state[ps2[2]]["step"] == 7
    m = b.state[ps2[1]]["exp_avg"]
    assert m.data_ptr() == b.flat_m.data_ptr() + 4 * b.offsets[1]  # still a view of the flat buffer
    em = b.ema_module(torch.nn.ParameterList(ps2))
    assert all(not q.requires_grad for q in em.parameters())
    assert next(iter(em.parameters())).data_ptr() == b.flat_ema.data_ptr()


@pytest.mark.gpu
@pytest.mark.parametrize("clip,ema", [(None, None), (1.5, 0.99)])
def test_flat_adamw_matches_oracle(clip, ema):
    from equiformer_amd.optim import FlatAdamW
    dev = torch.device("cuda:0")
    ps0, grads = _toy(3)
    wds = [0.0, 0.0, 5e-3, 5e-3, 5e-3]
    tp = [torch.nn.Parameter(p.float().to(dev)) for p in ps0]
    opt = FlatAdamW([{"params": tp[:2], "weight_decay": 0.0}, {"params": tp[2:], "weight_decay": 5e-3}], lr=5e-4,
                    betas=(0.9, 0.999), eps=1e-8, clip_grad=clip, ema_decay=ema)
    p = [x.float().double().numpy().copy() for x in ps0]
    m = [np.zeros_like(x) for x in p]
    v = [np.zeros_like(x) for x in p]
    e = [x.copy() for x in p]
    for step, gs in enumerate(grads, 1):
        for t, g in zip(tp, gs):
            t.grad = g.float().to(dev)
        if step == 3:
            tp[3].grad = None  # torch.optim.AdamW leaves a parameter without gradient untouched (own step count too)
        opt.step()
        gn = [g.float().double().numpy() * (0.0 if (step == 3 and i == 3) else 1.0) for i, g in enumerate(gs)]
        coef = ooptim.clip_coef(gn, clip)[0] if clip else 1.0
        for i in range(len(p)):
            if not (step == 3 and i == 3):
                own_step = step - (1 if (i == 3 and step > 3) else 0)
                p[i], m[i], v[i] = ooptim.adamw_step(p[i], gn[i] * coef, m[i], v[i], own_step, 5e-4, 0.9, 0.999, 1e-8, wds[i])
            if ema:
                e[i] = ooptim.ema_update(e[i], p[i], ema)
            assert np.abs(p[i] - tp[i].detach().double().cpu().numpy()).max() < 2e-6 * max(1.0, np.abs(p[i]).max())
    sd = opt.state_dict()
    assert sd["state"][0]["step"] == len(grads) and sd["state"][0]["exp_avg"].shape == ps0[0].shape
    assert sd["state"][3]["step"] == len(grads) - (1 if len(grads) >= 3 else 0)
    assert np.abs(m[2] - opt.state[tp[2]]["exp_avg"].double().cpu().numpy()).max() < 1e-6
    if ema:
        class Holder(torch.nn.Module):
            def __init__(self, ps):
                super().__init__()
                self.ps = torch.nn.ParameterList(ps)
        em = opt.ema_module(Holder(tp))
        for i, q in enumerate(em.ps):
            assert np.abs(e[i] - q.double().cpu().numpy()).max() < 2e-6 * max(1.0, np.abs(e[i]).max())
            assert not q.requires_grad


@pytest.mark.gpu
def test_flat_adamw_trains_model_like_torch_adamw():
    """Three train steps of the reduced QM9 model: fused flat optimizer vs torch.optim.AdamW on identical replicas."""
    import copy
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    from weights import fill_deterministic
    from equiformer_amd.nets.graph_attention_transformer import GraphAttentionTransformer
    from equiformer_amd.optim import FlatAdamW, add_weight_decay
    from equiformer_amd.synthetic import qm9_like_batch
    dev = torch.device("cuda:0")
    a = fill_deterministic(GraphAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32,
                                                     **mg.SMALL_L2), 5).to(dev).eval()
    b = copy.deepcopy(a)
    oa = torch.optim.AdamW(add_weight_decay(a, 5e-3, a.no_weight_decay()), lr=5e-4)
    ob = FlatAdamW(add_weight_decay(b, 5e-3, b.no_weight_decay()), lr=5e-4)
    d = {k: v.to(dev) for k, v in qm9_like_batch(4, 12, side=5.0, seed=2).items()}
    # the same gradients go to both optimizers (two backward passes differ by atomics-order noise, which Adam turns into
    # +-lr steps wherever the true gradient is zero)
    for _ in range(3):
        oa.zero_grad(set_to_none=True)
        ob.zero_grad(set_to_none=True)
        loss = (a(f_in=None, pos=d["pos"], batch=d["batch"], node_atom=d["z"]).squeeze() - d["y"]).abs().mean()
        loss.backward()
        for pa, pb in zip(a.parameters(), b.parameters()):
            pb.grad = None if pa.grad is None else pa.grad.clone()
        oa.step()
        ob.step()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert (pa.detach() - pb.detach()).abs().max() <= 2e-6 * max(1.0, float(pa.detach().abs().max())), n
    # and the re-pointed parameters still drive the model: same prediction from both replicas
    with torch.no_grad():
        ya = a(f_in=None, pos=d["pos"], batch=d["batch"], node_atom=d["z"])
        yb = b(f_in=None, pos=d["pos"], batch=d["batch"], node_atom=d["z"])
    assert (ya - yb).abs().max() <= 1e-4 * max(1.0, float(ya.abs().max()))


@pytest.mark.gpu
def test_flat_adamw_leaves_gradless_parameters_untouched():
    """torch.optim.AdamW skips a parameter whose .grad is None (no weight decay, no moment decay); the fused flat-buffer
    step must do the same, EMA included (the EMA still follows the unchanged weight)."""
    from equiformer_amd.optim import FlatAdamW
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    ta = [torch.randn(n, generator=g).to(dev).requires_grad_(True) for n in (33, 128, 7)]
    tb = [t.detach().clone().requires_grad_(True) for t in ta]
    oa = torch.optim.AdamW(ta, lr=1e-2, weight_decay=0.1)
    ob = FlatAdamW(tb, lr=1e-2, weight_decay=0.1, ema_decay=0.9)
    ema = [t.detach().clone() for t in tb]
    for step in range(4):
        for i, (pa, pb) in enumerate(zip(ta, tb)):
            if i == 1 and step in (1, 2):  # parameter 1 receives no gradient in steps 1 and 2
                pa.grad = pb.grad = None
            else:
                gr = torch.randn(pa.shape, generator=g).to(dev)
                pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        ema = [0.9 * e + 0.1 * p.detach() for e, p in zip(ema, tb)]
    for pa, pb in zip(ta, tb):
        assert (pa.detach() - pb.detach()).abs().max() <= 2e-6 * max(1.0, float(pa.detach().abs().max()))
    for e, p, off in zip(ema, tb, ob.offsets):  # (every slice of the flat buffers starts 256-byte aligned)
        got = ob.flat_ema[off:off + p.numel()]
        assert (got - e).abs().max() <= 2e-6
