#!/usr/bin/env python
"""(needs a development build: EQF_EXTRA_FLAGS="-DEQF_DEV_SWITCHES=1" python -m equiformer_amd.build --force)
Development aid: the radial-MLP GEMM shapes with phases switched off (eqf_gemm_debug_exp: 1 no stores, 2 no MFMA)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib as _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
E = 25354


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


L = _lib.load()
for (K, N) in ((64, 960), (128, 64)):
    x = torch.randn(E, K, device=dev)
    W = torch.randn(N, K, device=dev)
    dy = torch.randn(E, N, device=dev)
    for mask in (0, 1, 2):
        L.eqf_gemm_debug_exp(mask)
        t1 = timeit(lambda: ops._dense_fwd(x, W, None))
        t2 = timeit(lambda: ops._dense_dgrad(dy, W))
        print("K=%4d N=%4d exp=%d  fwd %7.1f us   dgrad %7.1f us" % (K, N, mask, t1, t2), flush=True)
    L.eqf_gemm_debug_exp(0)
    t0 = timeit(lambda: torch.empty((E, N), device=dev))
    tc = timeit(lambda: dy.clone())
    print("   (torch.empty %.1f us, clone of [E,%d] %.1f us)" % (t0, N, tc))
