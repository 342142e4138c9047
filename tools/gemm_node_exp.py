#!/usr/bin/env python
"""(needs a development build: EQF_EXTRA_FLAGS="-DEQF_DEV_SWITCHES=1" python -m equiformer_amd.build --force)
Development aid: the node-row grouped GEMMs of a block (2 304 nodes) with phases switched off
(eqf_gemm_debug_exp: 1 no stores, 2 no MFMA, 3 neither)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib as _lib, ops  # noqa: E402
from equiformer_amd.layout import RowLayout  # noqa: E402

dev = torch.device("cuda:0")
n = 2304


def timeit(fn, k=50):
    """mean kernel time from the library's own HIP-event pairs (the Python call itself takes ~15 us, more than the kernel)"""
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


L = _lib.load()
CASES = [("node linear 480 -> 480", "128x0e+64x1e+32x2e", "128x0e+64x1e+32x2e"),
         ("ffn fctp_1 480 -> 3x", "128x0e+64x1e+32x2e", "384x0e+192x1e+96x2e"),
         ("ffn fctp_2 3x -> 480", "384x0e+192x1e+96x2e", "128x0e+64x1e+32x2e")]
for name, a, b in CASES:
    li, lo = RowLayout(a), RowLayout(b)
    spec = ops.LinearSpec(li, lo)
    x = torch.randn(n, li.dim, device=dev)
    dy = torch.randn(n, lo.dim, device=dev)
    w = torch.randn(spec.weight_numel, device=dev)
    dw = torch.zeros_like(w)
    for mask in (0, 1, 2, 3):
        L.eqf_gemm_debug_exp(mask)
        t1 = timeit(lambda: ops._lin_fwd(x, w, None, spec))
        t2 = timeit(lambda: ops._lin_dgrad(dy, w, spec))
        t3 = timeit(lambda: ops._lin_wgrad(x, dy, spec, dw))
        print("%-24s exp=%d  fwd %6.1f us  dgrad %6.1f us  wgrad %6.1f us" % (name, mask, t1, t2, t3), flush=True)
    L.eqf_gemm_debug_exp(0)
