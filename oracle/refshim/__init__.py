"""ORACLE (test infrastructure, NOT product code) -- run the REFERENCE'S OWN model code on CPU.

The reference's `nets/` package (plain Python, /root/reference/nets/*.py) cannot be imported as it stands because five
un-vendored third-party packages are absent from this image and there is no network: e3nn 0.4.4, torch_scatter 2.0.9,
torch_cluster 1.6.0, torch_geometric 2.0.3, ocpmodels 0.0.3 (env/env_equiformer.yml, docs/env_setup.md).  `nets/` only
touches ~25 symbols of them (nets/graph_attention_transformer.py:1-27, nets/tensor_product_rescale.py:5-12,
nets/fast_activation.py:9-11,27, nets/graph_attention_transformer_oc20.py:46-51).  `pkgs/` holds thin stand-ins for
exactly those symbols, each delegating to the restated primitive in oracle/e3.py, oracle/nets.py (graph ops, Bessel
basis) or oracle/pbc.py; `load_reference_nets()` puts `pkgs/` and the reference checkout on sys.path and imports the
reference's `nets` package UNCHANGED -- every class of the model code that then runs (TensorProductRescale, LinearRS,
SeparableFCTP, GraphAttention, TransBlock, EquivariantLayerNormV2, Gate / Activation, RadialProfile, the model classes,
the registered factories) is the reference's own source file, executed where it lies.

What this pins and what it does not:
  * pinned by execution: all model code of the reference (4 924 lines of nets/) -- oracle/nets.py and every fixture of
    tests/golden/*.npz are checked against it (tests/test_reference_pin.py, fp64, <= 1e-9);
  * restated (the stand-ins): the dependency primitives -- Irreps algebra, real Wigner-3j, spherical harmonics, the
    TensorProduct contraction in 'uvw' / 'uvu' / 'uuu' mode with path_normalization='none', normalize2mom, scatter,
    segment softmax, radius_graph, radius_graph_pbc / get_pbc_distances, the spherical Bessel RadialBasis.  These are
    pinned from OUTSIDE this code base by tests/test_independent_kat.py (Gaunt integrals of scipy's harmonics by
    quadrature, Gauss-Hermite moments) and tests/test_oracle_kat.py.

/root/reference does not travel to the GPU box: anything that calls load_reference_nets() must skip when the checkout
is missing (reference_available()).  Nothing under equiformer_amd/ imports this package.
"""
import importlib
import os
import sys

PKGS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pkgs")
DEFAULT_REFERENCE = os.environ.get("EQF_REFERENCE", "/root/reference")
SHIMMED = ("e3nn", "torch_scatter", "torch_cluster", "torch_geometric", "ocpmodels")


def reference_available(reference=DEFAULT_REFERENCE):
    return os.path.isfile(os.path.join(reference, "nets", "graph_attention_transformer.py"))


def load_reference_nets(reference=DEFAULT_REFERENCE):
    """Import the reference's `nets` package (unchanged source) with the stand-in dependencies; returns the module."""
    if not reference_available(reference):
        raise FileNotFoundError("no reference checkout at %s" % reference)
    mod = sys.modules.get("nets")
    if mod is not None and os.path.dirname(os.path.dirname(os.path.abspath(mod.__file__))) == os.path.abspath(reference):
        return mod
    for name in SHIMMED:  # a real installation of any of them would silently win: refuse to mix
        m = sys.modules.get(name)
        if m is not None and not os.path.abspath(getattr(m, "__file__", "")).startswith(PKGS):
            raise RuntimeError("%s is already imported from %s" % (name, getattr(m, "__file__", "?")))
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for p in (root, PKGS, os.path.abspath(reference)):
        if p not in sys.path:
            sys.path.insert(0, p) if p != root else sys.path.append(p)
    mod = importlib.import_module("nets")
    assert os.path.abspath(mod.__file__).startswith(os.path.abspath(reference)), mod.__file__
    return mod
