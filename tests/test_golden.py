"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): fp64 oracle outputs on fixed inputs and
deterministic weights.  CPU: the oracle still reproduces them (freezes the oracle's arithmetic).  GPU: the HIP hot
path reproduces them through the drop-in `nets` modules within the north-star tolerance (1e-4 relative, fp32)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402
from weights import fill_deterministic  # noqa: E402

TOL = 1e-4  # BASELINE.json north_star: 1e-4 relative, fp32


def _load(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    ins = {k[4:]: z[k] for k in z.files if k.startswith("in::")}
    outs = {k[5:]: z[k] for k in z.files if k.startswith("out::")}
    return ins, outs


def _rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double().cpu() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _oc20_inputs(ins, dev=None, dtype=torch.float32):
    t = lambda k, dt=None: torch.as_tensor(ins[k]).to(dt) if dt else torch.as_tensor(ins[k])
    d = dict(atomic_numbers=t("z"), tags=t("tags"), pos=t("pos", dtype), batch=t("batch"),
             edge_index=t("edge_index"), offsets=t("offsets", dtype))
    if dev is not None:
        d = {k: v.to(dev) for k, v in d.items()}
    return d


# ------------------------------------------------------------------------------------------------- CPU: oracle
def test_oracle_reproduces_qm9_fixture():
    from oracle import nets as onets
    ins, outs = _load("qm9_small")
    m = fill_deterministic(onets.GraphAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32,
                                                           **mg.SMALL_L2).eval(), 11)
    pos, z, batch = torch.as_tensor(ins["pos"]), torch.as_tensor(ins["z"]), torch.as_tensor(ins["batch"])
    y64 = m.double()(None, pos.double(), batch, z)
    assert _rel(y64, outs["energy"]) < 1e-10
    loss = (y64.squeeze() - torch.as_tensor(ins["y"]).double()).abs().mean()
    g = torch.autograd.grad(loss, [m.blocks[0].ga.sep_act.lin.tp.weight, m.blocks[1].ga.alpha_dot])
    assert _rel(g[0], outs["g_sep_act_lin"]) < 1e-9 and _rel(g[1], outs["g_alpha_dot"]) < 1e-9
    y32 = m.float()(None, pos, batch, z)
    assert _rel(y32, outs["energy"]) < TOL


def test_oracle_reproduces_linear_message_fixture():
    from oracle import nets as onets
    ins, outs = _load("qm9_small_linear")
    m = fill_deterministic(onets.GraphAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32,
                                                           **dict(mg.SMALL_L2, nonlinear_message=False)).eval(), 14)
    pos, z, batch = torch.as_tensor(ins["pos"]), torch.as_tensor(ins["z"]), torch.as_tensor(ins["batch"])
    y64 = m.double()(None, pos.double(), batch, z)
    assert _rel(y64, outs["energy"]) < 1e-10
    loss = (y64.squeeze() - torch.as_tensor(ins["y"]).double()).abs().mean()
    g = torch.autograd.grad(loss, [m.blocks[0].ga.sep.lin.tp.weight, m.blocks[0].ga.sep.lin.bias[0]])
    assert _rel(g[0], outs["g_sep_lin"]) < 1e-9 and _rel(g[1], outs["g_sep_bias"]) < 1e-9


@pytest.mark.parametrize("tag,kw", [("md17_small_l2", mg.SMALL_L2), ("md17_small_l3", mg.SMALL_L3)])
def test_oracle_reproduces_md17_fixture(tag, kw):
    from oracle import nets as onets
    ins, outs = _load(tag)
    m = fill_deterministic(onets.GraphAttentionTransformerMD17(irreps_in="64x0e", max_radius=5.0, number_of_basis=32,
                                                               basis_type="exp", **kw).eval(), 12)
    e, f = m.double()(torch.as_tensor(ins["z"]), torch.as_tensor(ins["pos"]).double(), torch.as_tensor(ins["batch"]))
    assert _rel(e, outs["energy"]) < 1e-10 and _rel(f, outs["forces"]) < 1e-9
    e32, f32 = m.float()(torch.as_tensor(ins["z"]), torch.as_tensor(ins["pos"]), torch.as_tensor(ins["batch"]))
    assert _rel(e32, outs["energy"]) < TOL and _rel(f32, outs["forces"]) < TOL


def test_oracle_reproduces_oc20_fixture():
    from oracle import nets as onets
    ins, outs = _load("oc20_small")
    m = fill_deterministic(onets.GraphAttentionTransformerOC20(number_of_basis=32, **mg.SMALL_OC20).eval(), 13)
    e = m.double()(**_oc20_inputs(ins, dtype=torch.float64))
    assert _rel(e, outs["energy"]) < 1e-10


# ---- the other model families (tests/golden/make_golden.py --variants): one table drives the CPU and the GPU tests
def _variant_cases():
    """name -> (oracle class, product (module, class), seed, constructor kwargs, kind)"""
    return {
        "dp_qm9_small": ("DotProductAttentionTransformer", ("dp_attention_transformer", "DotProductAttentionTransformer"), 15,
                         dict(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **mg.SMALL_DP_L2), "qm9"),
        "dp_md17_small": ("DotProductAttentionTransformerMD17",
                          ("dp_attention_transformer", "DotProductAttentionTransformerMD17"), 19,
                          dict(irreps_in="64x0e", max_radius=5.0, number_of_basis=32, basis_type="exp", **mg.SMALL_DP_L2),
                          "md17"),
        "e3_qm9_small": ("GraphAttentionTransformer", ("graph_attention_transformer", "GraphAttentionTransformer"), 16,
                         dict(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **mg.SMALL_E3_L2), "qm9"),
        "oc20_aux_small": ("GraphAttentionTransformerOC20",
                           ("graph_attention_transformer_oc20", "GraphAttentionTransformerOC20"), 17,
                           dict(mg.SMALL_OC20, number_of_basis=32, use_auxiliary_task=True, irreps_feature="64x0e+32x1e"),
                           "oc20"),
        "dens_small": ("Equiformer_MD17_DeNS", ("equiformer_md17_dens", "Equiformer_MD17_DeNS"), 18, dict(mg.SMALL_DENS),
                       "dens"),
    }


def _run_variant(m, ins, kind, dev=None, dtype=torch.float32):
    from types import SimpleNamespace
    t = lambda k, dt=None: (torch.as_tensor(ins[k]).to(dt) if dt else torch.as_tensor(ins[k])).to(dev or "cpu")  # noqa: E731
    if kind == "qm9":
        return dict(energy=m(None, t("pos", dtype), t("batch"), t("z")))
    if kind == "md17":
        e, f = m(t("z"), t("pos", dtype), t("batch"))
        return dict(energy=e, forces=f)
    if kind == "oc20":
        d = _oc20_inputs(ins, dev, dtype)
        if dev is None:
            e, a = m(**d)
        else:
            e, a = m(SimpleNamespace(**d))
        return dict(energy=e, aux=a)
    data = SimpleNamespace(z=t("z"), pos=t("pos", dtype), batch=t("batch"), force=t("force", dtype),
                           noise_mask=t("noise_mask"))
    e, dy = m(data)
    return dict(energy=e, dy=dy)


@pytest.mark.parametrize("name", sorted(_variant_cases()))
def test_oracle_reproduces_variant_fixture(name):
    from oracle import nets as onets
    ocls, _, seed, kw, kind = _variant_cases()[name]
    ins, outs = _load(name)
    m = fill_deterministic(getattr(onets, ocls)(**kw).eval(), seed).double()
    got = _run_variant(m, ins, kind, dtype=torch.float64)
    for k in outs:
        assert _rel(got[k], outs[k]) < 1e-9, (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_variant_cases()))
def test_hip_reproduces_variant_fixture(name):
    _, (module, cls), seed, kw, kind = _variant_cases()[name]
    ins, outs = _load(name)
    kw = dict(kw)
    if kind == "oc20":
        m = _hip_model(cls, module, seed, num_atoms=None, bond_feat_dim=None, num_targets=1, **kw)
    else:
        m = _hip_model(cls, module, seed, **kw)
    got = _run_variant(m, ins, kind, dev=torch.device("cuda:0"))
    for k in outs:
        assert _rel(got[k], outs[k]) < TOL, (name, k, _rel(got[k], outs[k]))


# ------------------------------------------------------------------------------------------------- GPU: HIP path
def _hip_model(cls_name, module, seed, **kw):
    import importlib
    mod = importlib.import_module("equiformer_amd.nets." + module)
    m = fill_deterministic(getattr(mod, cls_name)(**kw).eval(), seed)
    return m.to(torch.device("cuda:0"))


@pytest.mark.gpu
def test_hip_reproduces_qm9_fixture():
    dev = torch.device("cuda:0")
    ins, outs = _load("qm9_small")
    m = _hip_model("GraphAttentionTransformer", "graph_attention_transformer", 11, irreps_in="5x0e", max_radius=5.0,
                   number_of_basis=32, **mg.SMALL_L2)
    pos, z, batch = (torch.as_tensor(ins[k]).to(dev) for k in ("pos", "z", "batch"))
    y = m(None, pos, batch, z)
    assert _rel(y, outs["energy"]) < TOL
    loss = (y.squeeze() - torch.as_tensor(ins["y"]).to(dev)).abs().mean()
    assert abs(loss.item() - float(outs["loss"])) < TOL * max(1.0, abs(float(outs["loss"])))
    g = torch.autograd.grad(loss, [m.blocks[0].ga.sep_act.lin.tp.weight, m.blocks[1].ga.alpha_dot,
                                   m.blocks[0].ga.sep_act.dtp_rad.net[0].weight, m.rbf.mean])
    for got, key in zip(g, ("g_sep_act_lin", "g_alpha_dot", "g_rad0", "g_rbf_mean")):
        assert _rel(got, outs[key]) < 1e-4, key


@pytest.mark.gpu
def test_hip_reproduces_linear_message_fixture():
    dev = torch.device("cuda:0")
    ins, outs = _load("qm9_small_linear")
    m = _hip_model("GraphAttentionTransformer", "graph_attention_transformer", 14, irreps_in="5x0e", max_radius=5.0,
                   number_of_basis=32, **dict(mg.SMALL_L2, nonlinear_message=False))
    pos, z, batch = (torch.as_tensor(ins[k]).to(dev) for k in ("pos", "z", "batch"))
    y = m(None, pos, batch, z)
    assert _rel(y, outs["energy"]) < TOL
    loss = (y.squeeze() - torch.as_tensor(ins["y"]).to(dev)).abs().mean()
    g = torch.autograd.grad(loss, [m.blocks[0].ga.sep.lin.tp.weight, m.blocks[1].ga.alpha_dot,
                                   m.blocks[0].ga.sep.lin.bias[0]])
    for got, key in zip(g, ("g_sep_lin", "g_alpha_dot", "g_sep_bias")):
        assert _rel(got, outs[key]) < 1e-4, key


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw", [("md17_small_l2", mg.SMALL_L2), ("md17_small_l3", mg.SMALL_L3)])
def test_hip_reproduces_md17_fixture(tag, kw):
    dev = torch.device("cuda:0")
    ins, outs = _load(tag)
    m = _hip_model("GraphAttentionTransformerMD17", "graph_attention_transformer_md17", 12, irreps_in="64x0e",
                   max_radius=5.0, number_of_basis=32, basis_type="exp", **kw)
    e, f = m(torch.as_tensor(ins["z"]).to(dev), torch.as_tensor(ins["pos"]).to(dev), torch.as_tensor(ins["batch"]).to(dev))
    assert _rel(e, outs["energy"]) < TOL and _rel(f, outs["forces"]) < TOL


@pytest.mark.gpu
def test_hip_reproduces_oc20_fixture():
    from types import SimpleNamespace
    dev = torch.device("cuda:0")
    ins, outs = _load("oc20_small")
    m = _hip_model("GraphAttentionTransformerOC20", "graph_attention_transformer_oc20", 13, number_of_basis=32,
                   **mg.SMALL_OC20)
    e = m(SimpleNamespace(**_oc20_inputs(ins, dev)))
    assert _rel(e, outs["energy"]) < TOL
