#!/usr/bin/env python
"""us per launch (the library's HIP-event pairs) of the linears of the QM9 step, per shape and matrix mode:
   python tools/gemm_shapes.py [modes, e.g. fp32,split,bf16]
node rows (2 304 nodes): LinearRS 480 -> 480, FFN 480 -> 3x and 3x -> 480; edge rows (25 354): the radial bank's grouped
nn.Linear layers (7 modules side by side: 128 -> 7 x 64, 64 -> 64, 64 -> 960)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib as _lib, ops  # noqa: E402
from equiformer_amd.layout import RowLayout  # noqa: E402

MODES = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fp32", "split"]
DIRECT = "--direct" in sys.argv
dev = torch.device("cuda:0")
n, E = 2304, 25354


def timeit(fn, k=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    _lib.prof_enable("")
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    rep = _lib.prof_report()
    _lib.prof_enable(None)
    return sum(v["total_ms"] for v in rep.values()) * 1e3 / max(1, sum(v["launches"] for v in rep.values()))


_lib.load()
if DIRECT:
    _lib.load().eqf_gemmx_dev_set(0, 0)
for a in sys.argv:
    if a.startswith("--tn-minsteps="):
        _lib.load().eqf_gemmx_dev_set(2, int(a.split("=")[1]))
rows = []
CASES = [("node 480 -> 480", "128x0e+64x1e+32x2e", "128x0e+64x1e+32x2e"),
         ("node ffn 480 -> 3x", "128x0e+64x1e+32x2e", "384x0e+192x1e+96x2e"),
         ("node ffn 3x -> 480", "384x0e+192x1e+96x2e", "128x0e+64x1e+32x2e")]
for name, a, b in CASES:
    li, lo = RowLayout(a), RowLayout(b)
    spec = ops.LinearSpec(li, lo)
    x = torch.randn(n, li.dim, device=dev)
    dy = torch.randn(n, lo.dim, device=dev)
    w = torch.randn(spec.weight_numel, device=dev)
    dw = torch.zeros_like(w)
    for mode in MODES:
        with ops.matrix_mode(mode):
            t1 = timeit(lambda: ops._lin_fwd(x, w, None, spec))
            t2 = timeit(lambda: ops._lin_dgrad(dy, w, spec))
            t3 = timeit(lambda: ops._lin_wgrad(x, dy, spec, dw))
        print("%-22s %-6s fwd %6.1f us  dgrad %6.1f us  wgrad %6.1f us" % (name, mode, t1, t2, t3), flush=True)
G, K = 7, 64
for name, Ns in (("edge 7 x (64 -> 64)", [64] * G), ("edge 7 x (64 -> 960)", [960] * G)):
    x = torch.randn(E, G * K, device=dev)
    Ws = [torch.randn(nn_, K, device=dev) * 0.1 for nn_ in Ns]
    bs = [torch.randn(nn_, device=dev) for nn_ in Ns]
    ldo = sum(Ns)
    out = torch.empty(E, ldo, device=dev)
    dy = torch.randn(E, ldo, device=dev)
    dx = torch.empty_like(x)
    dWs = [torch.zeros_like(W) for W in Ws]
    offs = [sum(Ns[:g]) for g in range(G)]
    st = ops._stream
    r = ops.rows
    fwd = lambda: ops._gemm_group([ops._desc(1, (x, g * K), r(1, G * K, 0), (Ws[g], 0), K, (out, offs[g]), r(1, ldo, 0), bs[g],  # noqa: E731
                                             E, Ns[g], K) for g in range(G)], st())
    dgr = lambda: ops._gemm_group([ops._desc(0, (dy, offs[g]), r(1, ldo, 0), (Ws[g], 0), K, (dx, g * K), r(1, G * K, 0), None,  # noqa: E731
                                             E, K, Ns[g]) for g in range(G)], st())
    wgr = lambda: ops._gemm_group([ops._desc(3, (dy, offs[g]), r(1, ldo, 0), (x, g * K), K, (dWs[g], 0), r(1, G * K, 0), None,  # noqa: E731
                                             Ns[g], K, E) for g in range(G)], st())
    flops = 2.0 * E * K * sum(Ns)
    for mode in MODES:
        with ops.matrix_mode(mode):
            t1, t2, t3 = timeit(fwd), timeit(dgr), timeit(wgr)
        print("%-22s %-6s fwd %6.1f us (%5.1f TF/s)  dgrad %6.1f us (%5.1f)  wgrad %6.1f us (%5.1f)"
              % (name, mode, t1, flops / t1 / 1e6, t2, flops / t2 / 1e6, t3, flops / t3 / 1e6), flush=True)
x = torch.randn(E, 128, device=dev)
W = torch.randn(G * 64, 128, device=dev) * 0.1
b = torch.randn(G * 64, device=dev)
for mode in MODES:
    with ops.matrix_mode(mode):
        t1 = timeit(lambda: ops._dense_fwd(x, W, b))
    print("%-22s %-6s fwd %6.1f us" % ("edge 128 -> 448", mode, t1), flush=True)
