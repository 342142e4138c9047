#!/usr/bin/env python
"""Step times of the other BASELINE.json configs on one MI355X (not bench.py lines: for DESIGN.md's record).
   MD17 se_l2 / se_l3: force-loss train step (second-order backward), batch 8 / 5 aspirin frames (scripts' sizes);
   OC20 l1_256_nonlinear: energy train step, 16 slab-shaped structures of 78 atoms with periodic neighbour search."""
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import nets  # noqa: E402
from equiformer_amd.optim import FlatAdamW, add_weight_decay  # noqa: E402
from equiformer_amd.synthetic import md17_aspirin_batch  # noqa: E402

dev = torch.device("cuda:0")


def timed(step, n=10, warm=3):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def md17(name, frames, wf):
    torch.manual_seed(0)
    model = nets.model_entrypoint(name)(irreps_in="64x0e", radius=5.0, num_basis=32).to(dev).train()
    opt = FlatAdamW(add_weight_decay(model, 1e-6, model.no_weight_decay()), lr=5e-4)
    d = {k: v.to(dev) for k, v in md17_aspirin_batch(frames, seed=1).items()}
    ty, tf = torch.randn(frames, 1, device=dev), torch.randn(frames * 21, 3, device=dev)

    def train():
        opt.zero_grad(set_to_none=True)
        E, F = model(node_atom=d["z"], pos=d["pos"], batch=d["batch"])
        loss = (E - ty).abs().mean() + wf * (F - tf).norm(dim=1).mean()
        loss.backward()
        opt.step()

    model.eval()

    def evalf():
        with torch.no_grad():
            model(node_atom=d["z"], pos=d["pos"], batch=d["batch"])

    te = timed(evalf)
    model.train()
    tt = timed(train)
    print("%-55s frames %d: energy+forces eval %.2f ms, force-loss train step %.2f ms (%.0f frames/s)"
          % (name, frames, te, tt, frames / tt * 1e3), flush=True)


def oc20(B=16, Na=78):
    torch.manual_seed(0)
    model = nets.model_entrypoint("oc20_l1_256_nonlinear")().to(dev).train()
    opt = FlatAdamW(add_weight_decay(model, 1e-3, model.no_weight_decay()), lr=2e-4)
    g = torch.Generator().manual_seed(0)
    cell = torch.diag(torch.tensor([11.0, 11.0, 30.0]))[None].repeat(B, 1, 1)
    frac = torch.rand(B * Na, 3, generator=g) * torch.tensor([1.0, 1.0, 0.45])
    pos = frac @ cell[0]
    data = SimpleNamespace(pos=pos.to(dev), batch=torch.arange(B).repeat_interleave(Na).to(dev),
                           atomic_numbers=torch.randint(1, 84, (B * Na,), generator=g).to(dev),
                           tags=torch.randint(0, 3, (B * Na,), generator=g).to(dev), cell=cell.to(dev),
                           natoms=torch.full((B,), Na).to(dev))
    y = torch.randn(B, device=dev)
    from equiformer_amd.graph import EdgeGraph
    gr, _, _ = EdgeGraph.from_radius_pbc(data.pos, data.cell, data.batch, 5.0, 500)

    def train():
        opt.zero_grad(set_to_none=True)
        loss = (model(data).squeeze() - y).abs().mean()
        loss.backward()
        opt.step()

    tt = timed(train)
    print("%-55s %d structures x %d atoms, %d periodic edges: train step %.2f ms (%.0f structures/s)"
          % ("oc20_l1_256_nonlinear (otf_graph, use_pbc)", B, Na, gr.E, tt, B / tt * 1e3), flush=True)


def qm9(name, B=128):
    """QM9-shaped train step (the bench.py workload) of another registered model family."""
    from equiformer_amd.synthetic import qm9_like_batch
    torch.manual_seed(0)
    model = nets.model_entrypoint(name)(irreps_in="5x0e", radius=5.0, num_basis=128).to(dev).train()
    opt = FlatAdamW(add_weight_decay(model, 5e-3, model.no_weight_decay()), lr=5e-4)
    d = {k: v.to(dev) for k, v in qm9_like_batch(B, 18, side=6.5, seed=0).items()}

    def train():
        opt.zero_grad(set_to_none=True)
        y = model(f_in=None, pos=d["pos"], batch=d["batch"], node_atom=d["z"])
        (y.squeeze() - d["y"]).abs().mean().backward()
        opt.step()

    tt = timed(train, n=8, warm=3)
    print("%-55s %d molecules: train step %.2f ms (%.0f molecules/s)" % (name, B, tt, B / tt * 1e3), flush=True)


def dens(frames=5):
    torch.manual_seed(0)
    model = nets.model_entrypoint("equiformer_md17_dens_l2")().to(dev).train()
    opt = FlatAdamW(add_weight_decay(model, 1e-6, model.no_weight_decay()), lr=2e-4)
    d = {k: v.to(dev) for k, v in md17_aspirin_batch(frames, seed=1).items()}
    n = frames * 21
    data = SimpleNamespace(z=d["z"], pos=d["pos"], batch=d["batch"], force=torch.randn(n, 3, device=dev),
                           noise_mask=torch.rand(n, device=dev) < 0.25)
    ty, tf = torch.randn(frames, 1, device=dev), torch.randn(n, 3, device=dev)

    def train():
        opt.zero_grad(set_to_none=True)
        E, Y = model(data)
        ((E - ty).abs().mean() + 80.0 * (Y - tf).norm(dim=1).mean()).backward()
        opt.step()

    tt = timed(train, n=8, warm=3)
    print("%-55s frames %d (25 %% corrupted atoms): train step %.2f ms (%.0f frames/s)"
          % ("equiformer_md17_dens_l2", frames, tt, frames / tt * 1e3), flush=True)


md17("graph_attention_transformer_nonlinear_exp_l2_md17", 8, 80.0)
md17("graph_attention_transformer_nonlinear_exp_l3_md17", 5, 100.0)
oc20()
if "--variants" in sys.argv:
    for name in ("graph_attention_transformer_l2", "graph_attention_transformer_nonlinear_bessel_l2",
                 "dot_product_attention_transformer_l2", "graph_attention_transformer_nonlinear_l2_e3"):
        qm9(name)
    md17("dot_product_attention_transformer_exp_l2_md17", 8, 80.0)
    md17("graph_attention_transformer_nonlinear_attn_exp_l3_md17", 5, 100.0)
    dens()
