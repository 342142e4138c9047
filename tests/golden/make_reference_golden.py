#!/usr/bin/env python
"""Pin the oracle and the golden fixtures to the REFERENCE'S OWN model code.

    python tests/golden/make_reference_golden.py [--reference /root/reference]            # check (default)
    python tests/golden/make_reference_golden.py [--reference /root/reference] --write    # rewrite the out:: arrays

Imports the reference's `nets` package UNCHANGED (atomicarchitects/equiformer, nets/__init__.py:1-11), builds the same
reduced models as tests/golden/make_golden.py with the same deterministic weights (assigned BY NAME from
tests/golden/weights.py -- parameter names are part of the reference's API; a name present on one
side only is an error), feeds them the inputs stored in the fixtures and compares (or rewrites) the stored outputs in
fp64.  Two ways to satisfy the reference's five un-vendored dependencies:

  * its real environment (e3nn 0.4.4, torch_geometric, torch_scatter, torch_cluster, ocpmodels): nothing to do;
  * this image, where none of them exists: `oracle/refshim` supplies thin stand-ins for the ~25 symbols `nets/` touches
    (each delegating to the restated primitive of oracle/e3.py / oracle/pbc.py) -- the MODEL code that runs is still
    the reference's own source.  That is what `check_all()` does by default and what tests/test_reference_pin.py runs
    on every CPU test pass in the build container (the GPU box has no /root/reference: the test skips there).

The one adaptation needed to run the reference in fp64: its embeddings feed a float32 one-hot into LinearRS
(nets/graph_attention_transformer.py:686-688), which e3nn would reject against float64 weights; a forward pre-hook
casts the input of every LinearRS to the weight dtype.
"""
import argparse
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for _p in (HERE, ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import make_golden as mg  # noqa: E402  (reduced model configurations)
from weights import fill_deterministic  # noqa: E402

TOL = 1e-9  # fp64 against fp64


def load_fixture(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    ins = {k[4:]: z[k] for k in z.files if k.startswith("in::")}
    outs = {k[5:]: z[k] for k in z.files if k.startswith("out::")}
    return ins, outs


def import_reference(reference, shims=None):
    """-> the reference's `nets` module.  shims=None: use oracle/refshim only if the real dependencies are absent."""
    if shims is None:
        try:
            import e3nn  # noqa: F401
            import torch_geometric  # noqa: F401
            shims = "refshim" in os.path.abspath(e3nn.__file__)
        except ImportError:
            shims = True
    if shims:
        from oracle.refshim import load_reference_nets
        return load_reference_nets(reference)
    sys.path.insert(0, reference)
    import nets
    return nets


def as_double(ref_model):
    """reference model -> fp64 + the LinearRS input cast described in the header."""
    from nets.tensor_product_rescale import LinearRS
    ref_model = ref_model.double().eval()
    for m in ref_model.modules():
        if isinstance(m, LinearRS) and not hasattr(m, "_cast_hook"):
            m._cast_hook = m.register_forward_pre_hook(lambda mod, a: (a[0].to(mod.tp.weight.dtype),))
    return ref_model


def copy_by_name(src_model, dst_model):
    """Parameters by NAME, both directions checked; e3nn's extra buffers stay as the reference built them."""
    src = dict(src_model.named_parameters())
    dst = dict(dst_model.named_parameters())
    only_dst, only_src = sorted(set(dst) - set(src)), sorted(set(src) - set(dst))
    if only_dst or only_src:
        raise AssertionError("parameter names differ: destination-only %s, source-only %s" % (only_dst, only_src))
    with torch.no_grad():
        for name, p in dst.items():
            assert tuple(p.shape) == tuple(src[name].shape), (name, p.shape, src[name].shape)
            p.copy_(src[name].to(p.dtype))
    sb, db = dict(src_model.named_buffers()), dict(dst_model.named_buffers())
    for name in set(sb) & set(db):  # e.g. ExpNormalSmearing.means / betas: must already agree
        if sb[name].numel():
            assert torch.allclose(sb[name].double(), db[name].double(), rtol=1e-6, atol=0), name


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _oc20_data(ins, cell=7.0):
    """ocpmodels-shaped batch of the OC20 fixtures (cubic cell of make_golden.py): edge_index + integer cell_offsets +
    per-structure neighbour counts, i.e. the otf_graph=False contract of ..._oc20.py:280-293."""
    B = int(ins["batch"].max()) + 1
    ei = torch.as_tensor(ins["edge_index"])
    batch = torch.as_tensor(ins["batch"])
    return SimpleNamespace(pos=torch.as_tensor(ins["pos"]).double(), batch=batch,
                           atomic_numbers=torch.as_tensor(ins["z"]), tags=torch.as_tensor(ins["tags"]), edge_index=ei,
                           cell=(torch.eye(3) * cell)[None].repeat(B, 1, 1).double(),
                           cell_offsets=torch.as_tensor(np.rint(ins["offsets"] / cell)).double(),
                           neighbors=torch.bincount(batch[ei[1]], minlength=B),
                           natoms=torch.bincount(batch, minlength=B))


def reference_outputs(reference="/root/reference", shims=None, log=print):
    """{fixture tag: {output name: array}} computed by the reference's own model classes, and the same from the oracle
    restatement (oracle/nets.py) with identical weights: -> (ref_outs, oracle_outs)."""
    import_reference(reference, shims)
    from nets.graph_attention_transformer import GraphAttentionTransformer as RefQM9  # [ref: :737-899]
    from nets.graph_attention_transformer_md17 import GraphAttentionTransformerMD17 as RefMD17  # [ref: :127-327]
    from nets.graph_attention_transformer_oc20 import GraphAttentionTransformerOC20 as RefOC20  # [ref: :85-381]
    from nets.dp_attention_transformer import DotProductAttentionTransformer as RefDP  # [ref: :255-411]
    from nets.dp_attention_transformer_md17 import DotProductAttentionTransformerMD17 as RefDPMD17  # [ref: :57-235]
    from nets.equiformer_md17_dens import Equiformer_MD17_DeNS as RefDeNS  # [ref: :55-354]
    from oracle import nets as onets
    R, O = {}, {}
    t = torch.as_tensor

    def pair(ocls, rcls, seed, okw, rargs=(), rkw=None):
        om = fill_deterministic(ocls(**okw).eval(), seed).double()  # weights rounded to fp32 like make_golden.py
        rm = as_double(rcls(*rargs, **(okw if rkw is None else rkw)))
        copy_by_name(om, rm)
        return om, rm

    # ---- QM9-shaped, non-linear and linear messages [ref: GraphAttentionTransformer.forward :864-899]
    for tag, seed, kw, gkeys in (
            ("qm9_small", 11, mg.SMALL_L2, ("g_sep_act_lin", "g_alpha_dot", "g_rad0", "g_rbf_mean")),
            ("qm9_small_linear", 14, dict(mg.SMALL_L2, nonlinear_message=False), ("g_sep_lin", "g_alpha_dot", "g_sep_bias"))):
        ins, _ = load_fixture(tag)
        om, rm = pair(onets.GraphAttentionTransformer, RefQM9, seed,
                      dict(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **kw))
        for m, store in ((rm, R), (om, O)):
            e = m(f_in=None, pos=t(ins["pos"]).double(), batch=t(ins["batch"]), node_atom=t(ins["z"]))
            loss = (e.squeeze() - t(ins["y"]).double()).abs().mean()
            if tag == "qm9_small":
                ps = [m.blocks[0].ga.sep_act.lin.tp.weight, m.blocks[1].ga.alpha_dot,
                      m.blocks[0].ga.sep_act.dtp_rad.net[0].weight, m.rbf.mean]
            else:
                ps = [m.blocks[0].ga.sep.lin.tp.weight, m.blocks[1].ga.alpha_dot, m.blocks[0].ga.sep.lin.bias[0]]
            gs = torch.autograd.grad(loss, ps)
            store[tag] = dict(energy=e.detach().numpy(), loss=loss.item(), **{k: g.numpy() for k, g in zip(gkeys, gs)})

    # ---- MD17-shaped: energy and forces [ref: GraphAttentionTransformerMD17.forward :276-327]
    for tag, ocls, rcls, seed, kw in (
            ("md17_small_l2", onets.GraphAttentionTransformerMD17, RefMD17, 12, mg.SMALL_L2),
            ("md17_small_l3", onets.GraphAttentionTransformerMD17, RefMD17, 12, mg.SMALL_L3),
            ("dp_md17_small", onets.DotProductAttentionTransformerMD17, RefDPMD17, 19, mg.SMALL_DP_L2)):
        ins, _ = load_fixture(tag)
        om, rm = pair(ocls, rcls, seed, dict(irreps_in="64x0e", max_radius=5.0, number_of_basis=32, basis_type="exp", **kw))
        for m, store in ((rm, R), (om, O)):
            e, f = m(node_atom=t(ins["z"]), pos=t(ins["pos"]).double(), batch=t(ins["batch"]))
            store[tag] = dict(energy=e.detach().numpy(), forces=f.detach().numpy())

    # ---- QM9-shaped outputs of the dot-product-attention and E(3) families
    for tag, ocls, rcls, seed, kw in (
            ("dp_qm9_small", onets.DotProductAttentionTransformer, RefDP, 15, mg.SMALL_DP_L2),
            ("e3_qm9_small", onets.GraphAttentionTransformer, RefQM9, 16, mg.SMALL_E3_L2)):
        ins, _ = load_fixture(tag)
        om, rm = pair(ocls, rcls, seed, dict(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **kw))
        for m, store in ((rm, R), (om, O)):
            e = m(f_in=None, pos=t(ins["pos"]).double(), batch=t(ins["batch"]), node_atom=t(ins["z"]))
            store[tag] = dict(energy=e.detach().numpy())

    # ---- OC20-shaped [ref: GraphAttentionTransformerOC20.forward :305-381], precomputed periodic edges
    for tag, seed, kw in (("oc20_small", 13, dict(mg.SMALL_OC20, number_of_basis=32)),
                          ("oc20_aux_small", 17, dict(mg.SMALL_OC20, number_of_basis=32, use_auxiliary_task=True,
                                                      irreps_feature="64x0e+32x1e"))):
        ins, _ = load_fixture(tag)
        om, rm = pair(onets.GraphAttentionTransformerOC20, RefOC20, seed, kw, rargs=(None, None, 1),
                      rkw=dict(kw, use_pbc=True, otf_graph=False))
        ro = rm(_oc20_data(ins))
        oo = om(t(ins["z"]), t(ins["tags"]), t(ins["pos"]).double(), t(ins["batch"]), edge_index=t(ins["edge_index"]),
                offsets=t(ins["offsets"]).double())
        for out, store in ((ro, R), (oo, O)):
            out = out if isinstance(out, tuple) else (out,)
            store[tag] = dict(zip(("energy", "aux"), (x.detach().numpy() for x in out)))

    # ---- DeNS [ref: Equiformer_MD17_DeNS.forward :238-354]
    ins, _ = load_fixture("dens_small")
    om, rm = pair(onets.Equiformer_MD17_DeNS, RefDeNS, 18, dict(mg.SMALL_DENS))
    for m, store in ((rm, R), (om, O)):
        data = SimpleNamespace(z=t(ins["z"]), pos=t(ins["pos"]).double(), batch=t(ins["batch"]),
                               force=t(ins["force"]).double(), noise_mask=t(ins["noise_mask"]))
        e, dy = m(data)
        store["dens_small"] = dict(energy=e.detach().numpy(), dy=dy.detach().numpy())
    return R, O


def check_all(reference="/root/reference", shims=None, write=False, log=print):
    """Compare fixtures and oracle with the reference-executed outputs.  -> {(tag, key): (fixture err, oracle err)}"""
    R, O = reference_outputs(reference, shims, log)
    errs = {}
    for tag, got in R.items():
        ins, outs = load_fixture(tag)
        assert set(outs) == set(got), (tag, sorted(outs), sorted(got))
        for k in sorted(outs):
            errs[(tag, k)] = (rel(outs[k], got[k]), rel(O[tag][k], got[k]))
            log("  %-18s %-14s fixture vs reference %.3e   oracle vs reference %.3e" % ((tag, k) + errs[(tag, k)]))
        if write:
            arrs = {"in::" + k: v for k, v in ins.items()}
            arrs.update({"out::" + k: np.asarray(v) for k, v in got.items()})
            np.savez_compressed(os.path.join(HERE, tag + ".npz"), **arrs)
    missing = sorted(f[:-4] for f in os.listdir(HERE) if f.endswith(".npz") and f[:-4] not in R)
    assert not missing, "fixtures without a reference run: %s" % missing
    return errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference", help="checkout of atomicarchitects/equiformer")
    ap.add_argument("--write", action="store_true", help="overwrite the out:: arrays with the reference's outputs")
    ap.add_argument("--real-deps", action="store_true", help="never use oracle/refshim (the real environment is installed)")
    a = ap.parse_args()
    errs = check_all(a.reference, shims=False if a.real_deps else None, write=a.write)
    worst = max(max(v) for v in errs.values())
    print("worst relative difference vs the reference's own model code: %.3e (tolerance %.0e)" % (worst, TOL))
    if not a.write and worst > TOL:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
