"""ORACLE -- test infrastructure, NOT product code.

CPU restatement (plain torch / numpy, fp32 or fp64) of the reference's algorithm for the Equiformer hot path and of the
un-vendored third-party semantics it calls (e3nn 0.4.4, torch_scatter 2.0.9, torch_cluster 1.6.0, torch_geometric 2.0.3,
ocpmodels 0.0.3, timm 0.4.12); every function cites the reference file:line it follows.

PARITY PIN (round 3): the reference's `nets/` package is plain Python, but its five third-party dependencies are absent
from the build container and there is no network.  `oracle/refshim` supplies stand-ins for the ~25 dependency symbols
`nets/` touches and imports /root/reference/nets UNCHANGED, so the reference's own model code runs here on CPU:
  * MODEL CODE = REFERENCE EXECUTED: tests/test_reference_pin.py builds every model family through the reference's own
    classes / registered factories, copies weights by name (both directions checked) and finds oracle/nets.py and all ten
    tests/golden/*.npz equal to it in fp64 to <= 1e-9 (measured: 2e-15); the fixtures themselves were (re)written by
    tests/golden/make_reference_golden.py --write, i.e. they are outputs of the reference's model code;
  * DEPENDENCY PRIMITIVES = RESTATED (oracle/e3.py: Irreps algebra, real Wigner-3j, spherical harmonics, the
    TensorProduct contraction, normalize2mom; oracle/nets.py: radius_graph, scatter, segment softmax, Bessel basis;
    oracle/pbc.py): e3nn 0.4.4 / torch_scatter / torch_cluster / torch_geometric / ocpmodels themselves have never run
    here.  They are pinned from outside this code base by tests/test_independent_kat.py (Gaunt integrals of scipy's
    spherical harmonics by quadrature for every w3j table used, Gauss-Hermite moments for the activation constants) and
    by the convention / symmetry KATs of tests/test_oracle_kat.py.  tests/golden/make_reference_golden.py --real-deps
    repeats the whole check against a real installation where one exists.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this package; the product
(equiformer_amd/) never does (tests/test_host.py::test_product_never_imports_oracle).

  e3.py       e3nn semantics: Irreps algebra, real Wigner-3j, spherical harmonics, TensorProduct
  nets.py     the model classes of nets/*.py of the reference (graph attention, dot-product attention, DeNS; QM9 / MD17 / OC20)
  pbc.py      ocpmodels radius_graph_pbc / get_pbc_distances
  optim.py    AdamW + clip_grad_norm + ModelEmaV2 arithmetic
  collate.py  PyG Batch.from_data_list rules
  refshim/    dependency stand-ins + loader that run the reference's own nets/ package (the pin)
"""
