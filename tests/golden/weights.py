"""Deterministic, RNG-stream-independent weights for the golden fixtures: every parameter is filled from numpy's
PCG64 generator (stream-stable) with a scale chosen by the parameter's name, so fixtures hold inputs and outputs only."""
import numpy as np
import torch


def fill_deterministic(model, seed):
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            n = rng.standard_normal(tuple(p.shape))
            leaf = name.split(".")[-1]
            if leaf == "frequencies":        # Bessel basis: keep the k pi ladder, perturbed by 1 %
                v = p.detach().double().numpy() * (1.0 + 0.01 * n)
            elif name.startswith("rbf."):
                v = {"mean": np.abs(n) * 0.5, "std": 0.3 + 0.2 * np.abs(n), "weight": 1.0 + 0.05 * n, "bias": 0.05 * n}[leaf]
            elif leaf == "affine_weight" or (leaf == "weight" and p.dim() == 1 and "tp." not in name):
                v = 1.0 + 0.1 * n            # equivariant / radial LayerNorm gains
            elif leaf in ("affine_bias", "bias") or "bias." in name:
                v = 0.1 * n
            elif leaf == "offset":
                v = 0.1 * n
            elif leaf == "alpha_dot":
                v = 0.3 * n
            elif p.dim() >= 2:               # nn.Linear [out, in]
                v = n / np.sqrt(p.shape[-1])
            else:                            # flat e3nn tp.weight
                v = 0.15 * n
            p.copy_(torch.from_numpy(np.asarray(v)).to(p.dtype))
    return model
