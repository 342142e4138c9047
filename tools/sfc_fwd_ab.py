#!/usr/bin/env python
"""eqf_sfc_fwd at the bench shape: split-precision bf16 x 6 step (default) against the exact-fp32 MFMA step
(development switch 64): us per call and max deviation between the two."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib as _lib, ops  # noqa: E402
from equiformer_amd.layout import DtpTable, RowLayout  # noqa: E402
from equiformer_amd.lib import call  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 25354
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def timeit(fn, n=30):
    for _ in range(5):
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
    Wl = ops._ptr_array((d[0], weight.data_ptr() + 4 * o) for d, o in zip(spec.degs, spec.w_offs))
    L = _lib.load()
    outs = {}
    variants = (("default", 0), ("other (switch 64)", 64))
    o1 = torch.empty(E, lay.dim, device=dev)
    o2 = torch.empty(E, n2, device=dev) if n2 else None
    f = lambda: call("eqf_sfc_fwd", P(x), P(M), P(w), table.c_ref, Wl, None, P(weight2), None, P(o1), lay.c_ref, P(o2),  # noqa: E731
                     n2, E, st())
    for _ in range(200):  # clocks up before anything is timed
        f()
    times = {tag: [] for tag, _ in variants}
    for rnd in range(5):  # variants interleaved: position in the sequence must not decide
        for tag, mask in variants:
            L.eqf_sfc_debug_exp(mask)
            times[tag].append(timeit(f))
            L.eqf_sfc_debug_exp(0)
            if rnd == 0:
                outs[tag] = o1.clone()
    for tag, _ in variants:
        t = sorted(times[tag])
        print("%-10s %-34s median %7.1f us / call (min %.1f, max %.1f of 5 interleaved rounds)"
              % (name, tag, t[2], t[0], t[-1]), flush=True)
    a, b = outs["default"], outs["other (switch 64)"]
    print("%-10s split-precision default: %s; max |difference| between the two steps %.2e of the result scale"
          % (name, bool(L.eqf_sfc_debug_x6_default()), ((a - b).abs().max() / b.abs().max()).item()), flush=True)


run("sep_act", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True)
run("sep_value", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", 0, False)
