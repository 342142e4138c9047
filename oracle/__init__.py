"""ORACLE -- test infrastructure, NOT product code.

CPU restatement (plain torch / numpy, fp32 or fp64) of the reference's algorithm for the Equiformer hot path and of the
un-vendored third-party semantics it calls (e3nn 0.4.4, torch_scatter 2.0.9, torch_cluster 1.6.0, torch_geometric 2.0.3,
ocpmodels 0.0.3, timm 0.4.12); every function cites the reference file:line it follows.

PARITY UNPINNED: the reference (atomicarchitects/equiformer) ships no golden vectors and cannot be imported in the build
container (its dependencies are absent, there is no network).  What stands in for the pin:
  * known-answer tests of the conventions (tests/test_oracle_kat.py) and answers that do not come from this code base
    (tests/test_independent_kat.py: Gaunt integrals by quadrature of scipy's spherical harmonics, Gauss-Hermite values of
    the activation normalisation constants);
  * tests/golden/*.npz freeze the oracle's outputs (tests/test_golden.py);
  * tests/golden/make_reference_golden.py --reference <checkout of the reference> checks (or rewrites) every fixture
    against the REAL reference on a machine that has its environment.  Until that has been run, this header, the headers
    of the files below and DESIGN.md section 0 keep the words "parity unpinned".

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this package; the product
(equiformer_amd/) never does (tests/test_host.py::test_product_never_imports_oracle).

  e3.py       e3nn semantics: Irreps algebra, real Wigner-3j, spherical harmonics, TensorProduct
  nets.py     the model classes of nets/*.py of the reference (graph attention, dot-product attention, DeNS; QM9 / MD17 / OC20)
  pbc.py      ocpmodels radius_graph_pbc / get_pbc_distances
  optim.py    AdamW + clip_grad_norm + ModelEmaV2 arithmetic
  collate.py  PyG Batch.from_data_list rules
"""
