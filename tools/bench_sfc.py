#!/usr/bin/env python
"""Micro-benchmark of the fused SeparableFCTP kernels at the bench shape (QM9-L2, E = 25354 edges):
   python tools/bench_sfc.py [E]      -> us / call and achieved TFLOP/s for fwd, bwd_data, bwd_weight."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import ops  # noqa: E402
from equiformer_amd.layout import DtpTable, RowLayout  # noqa: E402
from equiformer_amd.lib import call  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 25354
ORDERS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1]
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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


def run(name, irr, sh_irr, out_irr, n2, use_w):
    table = DtpTable(irr, sh_irr, irr)
    lay = RowLayout(out_irr)
    spec = ops.SfcSpec(table, lay, n2=n2)
    assert spec.supported
    g = torch.Generator().manual_seed(0)
    x = torch.randn(E, table.layout_in.dim, generator=g).to(dev)
    M = torch.randn(E, table.m_numel, generator=g).to(dev)
    w = torch.randn(E, table.weight_numel, generator=g).to(dev) if use_w else None
    weight = torch.randn(spec.weight_numel, generator=g).to(dev)
    weight2 = torch.randn(spec.weight2_numel, generator=g).to(dev) if n2 else None
    dweight = torch.zeros_like(weight)
    dweight2 = torch.zeros_like(weight2) if n2 else None
    o1 = torch.empty(E, lay.dim, device=dev)
    o2 = torch.empty(E, n2, device=dev) if n2 else None
    d1 = torch.randn(E, lay.dim, generator=g).to(dev)
    d2 = torch.randn(E, n2, generator=g).to(dev) if n2 else None
    dx = torch.empty_like(x)
    dw = torch.empty_like(w) if use_w else None
    Wl = ops._ptr_array((d[0], weight.data_ptr() + 4 * o) for d, o in zip(spec.degs, spec.w_offs))
    dWl = ops._ptr_array((d[0], dweight.data_ptr() + 4 * o) for d, o in zip(spec.degs, spec.w_offs))
    flops = sum(2.0 * E * (2 * l3 + 1) * K * ncat for (l3, K, _, ncat) in spec.degs)
    f = lambda: call("eqf_sfc_fwd", P(x), P(M), P(w), table.c_ref, Wl, None, P(weight2), None, P(o1), lay.c_ref, P(o2),
                     n2, E, st())
    b = lambda: call("eqf_sfc_bwd_data", P(x), P(M), P(w), table.c_ref, Wl, P(weight2), P(d1), lay.c_ref, P(d2), n2,
                     P(dx), P(dw), None, E, st())
    wg = lambda: call("eqf_sfc_bwd_weight", P(x), P(M), P(w), table.c_ref, P(d1), lay.c_ref, P(d2), n2, dWl, P(dweight2),
                      E, st())
    from equiformer_amd import lib as _lib
    dbg = torch.zeros(8, dtype=torch.int64, device=dev)
    _lib.load().eqf_sfc_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    f()
    torch.cuda.synchronize()
    d = dbg.cpu().tolist()
    tot = sum(d[:7]) or 1
    print("%-10s fwd phase share (wave 0): prologue %.2f commit %.2f barrier %.2f issue(B) %.2f mfma %.2f barrier %.2f issue(x,w) %.2f"
          % ((name,) + tuple(v / tot for v in d[:7])), flush=True)
    dbg.zero_()
    b()
    torch.cuda.synchronize()
    _lib.load().eqf_sfc_debug_buffer(None)
    d = dbg.cpu().tolist()
    nb = ((E + 31) // 32)
    print("%-10s bwd_data phase cycles per workgroup-row (sum over groups / edge tiles): prologue %.0f staging %.0f mfma %.0f "
          "epilogue %.0f store %.0f" % ((name,) + tuple(v / nb for v in d[:5])), flush=True)
    for order in ORDERS:
        _lib.load().eqf_sfc_debug_order(order)
        for tag, fn in (("fwd", f), ("bwd_data", b), ("bwd_weight", wg)):
            us = timeit(fn)
            print("%-10s order %d %-10s E=%d  %8.1f us  %6.1f TFLOP/s  (%.2f GFLOP)"
                  % (name, order, tag, E, us, flops / us / 1e6, flops / 1e9), flush=True)
    _lib.load().eqf_sfc_debug_order(-1)
    # split-precision kernels (csrc/sfcx.hip): mode 0 = 2 + 3 planes, 1 = plain bf16, 2 = 3 + 3 planes
    for mode in (0, 1, 2):
        packed = ops._sfc_pack(weight, weight2, spec, mode)
        PK = ctypes.c_void_p(packed.data_ptr())
        fx = lambda: call("eqf_sfcx_fwd", P(x), P(M), P(w), table.c_ref, PK, None, None, P(o1), lay.c_ref, P(o2), n2, E, mode,
                          st())
        bx = lambda: call("eqf_sfcx_bwd_data", P(x), P(M), P(w), table.c_ref, PK, P(d1), lay.c_ref, P(d2), n2, P(dx), P(dw),
                          None, E, mode, st())
        wx = lambda: call("eqf_sfcx_bwd_weight", P(x), P(M), P(w), table.c_ref, P(d1), lay.c_ref, P(d2), n2, dWl, P(dweight2),
                          E, mode, st())
        pk = lambda: ops._sfc_pack(weight, weight2, spec, mode)
        for tag, fn in (("fwd", fx), ("bwd_data", bx), ("bwd_weight", wx), ("pack", pk)):
            us = timeit(fn)
            print("%-10s sfcx mode %d %-10s E=%d  %8.1f us  %6.1f TFLOP/s" % (name, mode, tag, E, us, flops / us / 1e6),
                  flush=True)


run("sep_act", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True)
run("sep_value", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", 0, False)
