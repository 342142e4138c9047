#!/usr/bin/env python
"""(needs a development build: EQF_EXTRA_FLAGS="-DEQF_DEV_SWITCHES=1" python -m equiformer_amd.build --force)
Development aid: time the fused SeparableFCTP kernels with individual phases switched off (eqf_sfc_debug_exp)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib as _lib, ops  # noqa: E402
from equiformer_amd.layout import DtpTable, RowLayout  # noqa: E402
from equiformer_amd.lib import call  # noqa: E402

E = 25354
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def run(name, irr, sh_irr, out_irr, n2, use_w):
    table = DtpTable(irr, sh_irr, irr)
    lay = RowLayout(out_irr)
    spec = ops.SfcSpec(table, lay, n2=n2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(E, table.layout_in.dim, generator=g).to(dev)
    M = torch.randn(E, table.m_numel, generator=g).to(dev)
    w = torch.randn(E, table.weight_numel, generator=g).to(dev) if use_w else None
    weight = torch.randn(spec.weight_numel, generator=g).to(dev)
    weight2 = torch.randn(spec.weight2_numel, generator=g).to(dev) if n2 else None
    o1 = torch.empty(E, lay.dim, device=dev)
    o2 = torch.empty(E, n2, device=dev) if n2 else None
    d1 = torch.randn(E, lay.dim, generator=g).to(dev)
    d2 = torch.randn(E, n2, generator=g).to(dev) if n2 else None
    dx = torch.empty_like(x)
    dw = torch.empty_like(w) if use_w else None
    Wl = ops._ptr_array((d[0], weight.data_ptr() + 4 * o) for d, o in zip(spec.degs, spec.w_offs))
    f = lambda: call("eqf_sfc_fwd", P(x), P(M), P(w), table.c_ref, Wl, None, P(weight2), None, P(o1), lay.c_ref, P(o2),
                     n2, E, st())
    b = lambda: call("eqf_sfc_bwd_data", P(x), P(M), P(w), table.c_ref, Wl, P(weight2), P(d1), lay.c_ref, P(d2), n2,
                     P(dx), P(dw), None, E, st())
    L = _lib.load()
    for mask in (0, 64, 1, 65, 2, 66):
        L.eqf_sfc_debug_exp(mask)
        print("%-10s fwd exp=%2d  %8.1f us" % (name, mask, timeit(f)), flush=True)
    for mask in (0,):
        L.eqf_sfc_debug_exp(mask)
        print("%-10s bwd exp=%2d  %8.1f us" % (name, mask, timeit(b)), flush=True)
    L.eqf_sfc_debug_exp(0)


run("sep_act", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True)
run("sep_value", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", 0, False)
