#!/usr/bin/env python
"""Inputs and model configurations of the golden fixtures of tests/golden/ (python tests/golden/make_golden.py).

Two steps make a fixture.  (1) This script fixes the inputs, the reduced model configurations (SMALL_*) and -- through
tests/golden/weights.py (numpy PCG64 stream, independent of torch's RNG) -- the weights, and stores the CPU oracle's fp64
outputs.  (2) tests/golden/make_reference_golden.py --write re-computes every output with the REFERENCE'S OWN model code
(/root/reference/nets imported unchanged, dependency stand-ins from oracle/refshim) and overwrites the out:: arrays; the
committed .npz files are the result of step 2.  tests/test_reference_pin.py re-checks them against the reference on
every CPU run in the build container, tests/test_golden.py checks the oracle (CPU) and the HIP path (GPU) against them.

Models are reduced copies of the BASELINE configs (2 blocks, 32-channel degrees, 64 scalar features) so that a fixture
is a few kB; all code paths (radius graph, SH, RBF, radial MLP, DTP, gate, attention, layer norm, FFN with
shortcut, head, pooling, MD17 forces, OC20 tag embedding + offsets) are exercised.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from equiformer_amd.synthetic import md17_aspirin_batch, qm9_like_batch  # noqa: E402
from oracle import nets as onets  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from weights import fill_deterministic  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

SMALL_L2 = dict(irreps_node_embedding="64x0e+32x1e+32x2e", num_layers=2, irreps_sh="1x0e+1x1e+1x2e",
                fc_neurons=[64, 64], irreps_feature="64x0e", irreps_head="16x0e+8x1e+8x2e", num_heads=4,
                nonlinear_message=True, irreps_mlp_mid="64x0e+32x1e+32x2e", alpha_drop=0.0)
SMALL_L3 = dict(irreps_node_embedding="64x0e+32x1e+32x2e+32x3e", num_layers=2, irreps_sh="1x0e+1x1e+1x2e+1x3e",
                fc_neurons=[64, 64], irreps_feature="64x0e", irreps_head="16x0e+8x1e+8x2e+8x3e", num_heads=4,
                nonlinear_message=True, irreps_mlp_mid="64x0e+32x1e+32x2e+32x3e", alpha_drop=0.0)
SMALL_OC20 = dict(irreps_node_embedding="64x0e+32x1e", num_layers=2, irreps_sh="1x0e+1x1e", max_radius=5.0,
                  fc_neurons=[64, 64], irreps_feature="64x0e", irreps_head="16x0e+8x1e", num_heads=4,
                  nonlinear_message=True, irreps_mlp_mid="128x0e+64x1e", alpha_drop=0.0)


SMALL_E3_L2 = dict(irreps_node_embedding="32x0e+16x0o+16x1e+16x1o+8x2e+8x2o", num_layers=2, irreps_sh="1x0e+1x1o+1x2e",
                   fc_neurons=[64, 64], irreps_feature="64x0e", irreps_head="8x0e+4x0o+4x1e+4x1o+4x2e+4x2o", num_heads=4,
                   nonlinear_message=True, irreps_mlp_mid="64x0e+16x0o+32x1e+16x1o+16x2e+16x2o", alpha_drop=0.0)
SMALL_DP_L2 = {k: v for k, v in SMALL_L2.items() if k != "nonlinear_message"}
SMALL_DENS = dict(SMALL_L2, irreps_equivariant_inputs="1x0e+1x1e+1x2e", irreps_feature="128x0e+64x1e+64x2e",
                  irreps_pre_attn=SMALL_L2["irreps_node_embedding"], number_of_basis=32, basis_type="exp")


def _save(name, model, inputs, outputs):
    arrs = {"in::" + k: np.asarray(v) for k, v in inputs.items()}
    arrs.update({"out::" + k: np.asarray(v) for k, v in outputs.items()})
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(name, "%.0f kB" % (os.path.getsize(path) / 1e3), {k: np.asarray(v).shape for k, v in outputs.items()})


def main():
    torch.manual_seed(1234)
    # ---- QM9-shaped (config #1/#2)
    m = onets.GraphAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **SMALL_L2).eval()
    fill_deterministic(m, 11)
    d = qm9_like_batch(3, 12, side=5.5, seed=7)
    md = m.double()
    pos = d["pos"].double().requires_grad_(True)
    y = md(None, pos, d["batch"], d["z"])
    loss = (y.squeeze() - d["y"].double()).abs().mean()
    grads = torch.autograd.grad(loss, [md.blocks[0].ga.sep_act.lin.tp.weight, md.blocks[1].ga.alpha_dot,
                                       md.blocks[0].ga.sep_act.dtp_rad.net[0].weight, md.rbf.mean])
    _save("qm9_small", m.float(), dict(pos=d["pos"].numpy(), z=d["z"].numpy(), batch=d["batch"].numpy(),
                                      y=d["y"].numpy()),
          dict(energy=y.detach().numpy(), loss=loss.item(), g_sep_act_lin=grads[0].numpy(),
               g_alpha_dot=grads[1].numpy(), g_rad0=grads[2].numpy(), g_rbf_mean=grads[3].numpy()))

    # ---- MD17-shaped, L_max = 2 and 3 (configs #3/#4): energy and forces
    for tag, kw in (("md17_small_l2", SMALL_L2), ("md17_small_l3", SMALL_L3)):
        m = onets.GraphAttentionTransformerMD17(irreps_in="64x0e", max_radius=5.0, number_of_basis=32,
                                                basis_type="exp", **kw).eval()
        fill_deterministic(m, 12)
        d = md17_aspirin_batch(2, seed=3)
        e, f = m.double()(d["z"], d["pos"].double(), d["batch"])
        _save(tag, m.float(), dict(pos=d["pos"].numpy(), z=d["z"].numpy(), batch=d["batch"].numpy()),
              dict(energy=e.detach().numpy(), forces=f.detach().numpy()))

    # ---- OC20-shaped (config #5): explicit edges with Cartesian offsets, tags
    m = onets.GraphAttentionTransformerOC20(number_of_basis=32, **SMALL_OC20).eval()
    fill_deterministic(m, 13)
    rng = np.random.default_rng(5)
    n, B = 20, 2
    cell = 7.0
    pos = rng.uniform(0, cell, size=(B * n, 3)).astype(np.float32).astype(np.float64)
    batch = np.repeat(np.arange(B), n)
    z = rng.integers(1, 84, size=B * n)
    tags = rng.integers(0, 3, size=B * n)
    src, dst, off = [], [], []
    shifts = [np.array([i, j, 0.0]) * cell for i in (-1, 0, 1) for j in (-1, 0, 1)]
    for b in range(B):
        for i in range(b * n, (b + 1) * n):
            for j in range(b * n, (b + 1) * n):
                for s in shifts:
                    if i == j and not s.any():
                        continue
                    if np.linalg.norm(pos[j] + s - pos[i]) < 5.0:
                        src.append(j), dst.append(i), off.append(s)
    e = m.double()(torch.tensor(z), torch.tensor(tags), torch.tensor(pos), torch.tensor(batch),
                   edge_index=torch.tensor(np.array([src, dst])), offsets=torch.tensor(np.array(off)))
    _save("oc20_small", m.float(), dict(pos=pos.astype(np.float32), z=z, tags=tags, batch=batch,
                                       edge_index=np.array([src, dst]), offsets=np.array(off, dtype=np.float32)),
          dict(energy=e.detach().numpy()))


def make_linear():
    """QM9-shaped, linear-message attention (nonlinear_message=False, reference :459-465,497-502); generated separately
    so that the older fixtures are not rewritten."""
    kw = dict(SMALL_L2, nonlinear_message=False)
    m = onets.GraphAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **kw).eval()
    fill_deterministic(m, 14)
    d = qm9_like_batch(3, 12, side=5.5, seed=8)
    md = m.double()
    y = md(None, d["pos"].double(), d["batch"], d["z"])
    loss = (y.squeeze() - d["y"].double()).abs().mean()
    grads = torch.autograd.grad(loss, [md.blocks[0].ga.sep.lin.tp.weight, md.blocks[1].ga.alpha_dot,
                                       md.blocks[0].ga.sep.lin.bias[0]])
    _save("qm9_small_linear", m.float(), dict(pos=d["pos"].numpy(), z=d["z"].numpy(), batch=d["batch"].numpy(),
                                             y=d["y"].numpy()),
          dict(energy=y.detach().numpy(), loss=loss.item(), g_sep_lin=grads[0].numpy(), g_alpha_dot=grads[1].numpy(),
               g_sep_bias=grads[2].numpy()))


def _slab(n, B, cell, seed):
    """n atoms per structure in a cubic cell, periodic along x and y: explicit edges + Cartesian offsets."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0, cell, size=(B * n, 3)).astype(np.float32).astype(np.float64)
    batch = np.repeat(np.arange(B), n)
    z = rng.integers(1, 84, size=B * n)
    tags = rng.integers(0, 3, size=B * n)
    src, dst, off = [], [], []
    shifts = [np.array([i, j, 0.0]) * cell for i in (-1, 0, 1) for j in (-1, 0, 1)]
    for b in range(B):
        for i in range(b * n, (b + 1) * n):
            for j in range(b * n, (b + 1) * n):
                for sft in shifts:
                    if i == j and not sft.any():
                        continue
                    if np.linalg.norm(pos[j] + sft - pos[i]) < 5.0:
                        src.append(j), dst.append(i), off.append(sft)
    return pos, batch, z, tags, np.array([src, dst]), np.array(off)


def make_variants():
    """The other model families (dot-product attention, E(3) irreps, OC20 auxiliary head, DeNS): outputs only."""
    from types import SimpleNamespace
    # ---- dot-product attention, QM9-shaped [ref: nets/dp_attention_transformer.py]
    m = fill_deterministic(onets.DotProductAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32,
                                                                **SMALL_DP_L2).eval(), 15).double()
    d = qm9_like_batch(3, 12, side=5.5, seed=9)
    y = m(None, d["pos"].double(), d["batch"], d["z"])
    _save("dp_qm9_small", m, dict(pos=d["pos"].numpy(), z=d["z"].numpy(), batch=d["batch"].numpy()),
          dict(energy=y.detach().numpy()))
    # ---- dot-product attention, MD17-shaped: energy and forces [ref: nets/dp_attention_transformer_md17.py]
    m = fill_deterministic(onets.DotProductAttentionTransformerMD17(irreps_in="64x0e", max_radius=5.0, number_of_basis=32,
                                                                    basis_type="exp", **SMALL_DP_L2).eval(), 19).double()
    d = md17_aspirin_batch(2, seed=4)
    e, f = m(d["z"], d["pos"].double(), d["batch"])
    _save("dp_md17_small", m, dict(pos=d["pos"].numpy(), z=d["z"].numpy(), batch=d["batch"].numpy()),
          dict(energy=e.detach().numpy(), forces=f.detach().numpy()))
    # ---- E(3) irreps, QM9-shaped [ref: graph_attention_transformer_nonlinear_l2_e3, :940-956]
    m = fill_deterministic(onets.GraphAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32,
                                                           **SMALL_E3_L2).eval(), 16).double()
    d = qm9_like_batch(3, 12, side=5.5, seed=10)
    y = m(None, d["pos"].double(), d["batch"], d["z"])
    _save("e3_qm9_small", m, dict(pos=d["pos"].numpy(), z=d["z"].numpy(), batch=d["batch"].numpy()),
          dict(energy=y.detach().numpy()))
    # ---- OC20 with the auxiliary IS2RS head on an l > 0 feature [ref: ..._oc20.py:182-194, :372-381]
    cfg = dict(SMALL_OC20, number_of_basis=32, use_auxiliary_task=True, irreps_feature="64x0e+32x1e")
    m = fill_deterministic(onets.GraphAttentionTransformerOC20(**cfg).eval(), 17).double()
    pos, batch, z, tags, ei, off = _slab(16, 2, 7.0, 6)
    e, a = m(torch.tensor(z), torch.tensor(tags), torch.tensor(pos), torch.tensor(batch), edge_index=torch.tensor(ei),
             offsets=torch.tensor(off))
    _save("oc20_aux_small", m, dict(pos=pos.astype(np.float32), z=z, tags=tags, batch=batch, edge_index=ei,
                                    offsets=off.astype(np.float32)),
          dict(energy=e.detach().numpy(), aux=a.detach().numpy()))
    # ---- DeNS: corrupted structure with encoded forces [ref: nets/equiformer_md17_dens.py:238-354]
    m = fill_deterministic(onets.Equiformer_MD17_DeNS(**SMALL_DENS).eval(), 18).double()
    d = md17_aspirin_batch(2, seed=5)
    rng = np.random.default_rng(8)
    force = rng.standard_normal((42, 3)).astype(np.float32)
    mask = rng.uniform(size=42) < 0.3
    e, dy = m(SimpleNamespace(z=d["z"], pos=d["pos"].double(), batch=d["batch"], force=torch.tensor(force).double(),
                              noise_mask=torch.tensor(mask)))
    _save("dens_small", m, dict(pos=d["pos"].numpy(), z=d["z"].numpy(), batch=d["batch"].numpy(), force=force,
                                noise_mask=mask),
          dict(energy=e.detach().numpy(), dy=dy.detach().numpy()))


if __name__ == "__main__":
    if "--linear" in sys.argv:
        make_linear()
    elif "--variants" in sys.argv:
        make_variants()
    else:
        main()
        make_linear()
        make_variants()
