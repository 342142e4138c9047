#!/usr/bin/env python
"""Development aid: bisect a parity problem of the full-size QM9 forward (128 molecules x 18 atoms).

For every kernel configuration (default, fp32-MFMA sfc forward, legacy fused path, un-fused path) it prints
  * determinism: the same input twice -> max |y1 - y2|,
  * the invariances of tests/test_gpu_properties.py (rotation, permutation, batch split),
  * the distance to the un-fused configuration on the same input,
and (with --oracle N) the distance to the fp64 CPU oracle on the first N molecules.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib as _lib, nets  # noqa: E402
from equiformer_amd.synthetic import qm9_like_batch  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rot(seed):
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=128)
    ap.add_argument("--oracle", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.load()
    torch.manual_seed(0)
    model = nets.model_entrypoint("graph_attention_transformer_nonlinear_l2")(irreps_in="5x0e", radius=5.0,
                                                                               num_basis=128).to(dev).eval()
    B = a.molecules
    d = {k: v.to(dev) for k, v in qm9_like_batch(B, 18, side=6.5, seed=11).items()}
    R = rot(1).to(dev)
    g = torch.Generator().manual_seed(2)
    perm = torch.cat([m * 18 + torch.randperm(18, generator=g) for m in range(B)]).to(dev)
    half = (B // 2) * 18

    def fwd(pos, batch, z):
        with torch.no_grad():
            return model(f_in=None, pos=pos, batch=batch, node_atom=z)

    results = {}
    configs = [("default", 0, True), ("fp32-mfma fwd (exp 64)", 64, True), ("legacy", 0, "legacy"), ("unfused", 0, False)]
    for name, exp, fused in configs:
        L.eqf_sfc_debug_exp(exp)
        model.set_fused(fused)
        y = fwd(d["pos"], d["batch"], d["z"])
        y2 = fwd(d["pos"], d["batch"], d["z"])
        y_rot = fwd(d["pos"] @ R.T + torch.tensor([0.3, -1.2, 2.0], device=dev), d["batch"], d["z"])
        y_perm = fwd(d["pos"][perm], d["batch"], d["z"][perm])
        y_a = fwd(d["pos"][:half], d["batch"][:half], d["z"][:half])
        y_b = fwd(d["pos"][half:], d["batch"][half:] - B // 2, d["z"][half:])
        results[name] = y
        print("%-26s determinism %.2e  rotation %.2e  permutation %.2e  batch split %.2e"
              % (name, rel(y2, y), rel(y_rot, y), rel(y_perm, y), rel(torch.cat([y_a, y_b]), y)), flush=True)
    L.eqf_sfc_debug_exp(0)
    model.set_fused(True)
    for name in results:
        print("%-26s vs unfused %.2e" % (name, rel(results[name], results["unfused"])), flush=True)
    if a.oracle:
        from oracle import nets as onets  # checker only
        n = a.oracle
        ref = onets.graph_attention_transformer_nonlinear_l2("5x0e", 5.0).double().eval()
        ref.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()}, strict=False)
        with torch.no_grad():
            yr = ref(None, d["pos"][:n * 18].double().cpu(), d["batch"][:n * 18].cpu(), d["z"][:n * 18].cpu())
        for name in results:
            print("%-26s vs fp64 oracle (first %d molecules) %.2e" % (name, n, rel(results[name][:n], yr)), flush=True)


if __name__ == "__main__":
    main()
