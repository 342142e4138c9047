#!/usr/bin/env python
"""Does the weight gradient of the fused SeparableFCTP (a full-occupancy, throughput-bound kernel that nothing in backward waits
for) overlap with a chain of small latency-bound node kernels when it runs on a second HIP stream?
   python tools/overlap_probe.py
Prints: the chain alone, the weight gradient alone, both on one stream, both on two streams."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib as _lib, ops  # noqa: E402
from equiformer_amd.layout import DtpTable, RowLayout  # noqa: E402
from equiformer_amd.lib import call  # noqa: E402

E, N = 25354, 2304
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
L = _lib.load()


def raw(stream):
    return ctypes.c_void_p(stream.cuda_stream)


irr = "128x0e+64x1e+32x2e"
table = DtpTable(irr, "1x0e+1x1e+1x2e", irr)
lay = RowLayout("224x0e+64x1e+32x2e")
spec = ops.SfcSpec(table, lay, n2=128)
g = torch.Generator().manual_seed(0)
x = torch.randn(E, table.layout_in.dim, generator=g).to(dev)
M = torch.randn(E, table.m_numel, generator=g).to(dev)
w = torch.randn(E, table.weight_numel, generator=g).to(dev)
d1 = torch.randn(E, lay.dim, generator=g).to(dev)
d2 = torch.randn(E, 128, generator=g).to(dev)
dweight = torch.zeros(spec.weight_numel, device=dev)
dweight2 = torch.zeros(spec.weight2_numel, device=dev)
dWl = ops._ptr_array((d[0], dweight.data_ptr() + 4 * o) for d, o in zip(spec.degs, spec.w_offs))

nl = RowLayout(irr)
xn = torch.randn(N, nl.dim, generator=g).to(dev)
yn = torch.empty_like(xn)
wln = torch.ones(224, device=dev)
bln = torch.zeros(128, device=dev)
rstd = torch.empty(N * 3, device=dev)
mean0 = torch.empty(N, device=dev)


def wgrad(st):
    call("eqf_sfcx_bwd_weight", P(x), P(M), P(w), table.c_ref, P(d1), lay.c_ref, P(d2), 128, dWl, P(dweight2), E, 0, raw(st))


def chain(st, n=20):
    for _ in range(n):  # dependent chain of node-row kernels (13 us each in the step)
        call("eqf_layernorm_fwd", P(xn), P(wln), P(bln), P(yn), P(rstd), P(mean0), N, nl.c_ref, 1e-5, raw(st))


main = torch.cuda.current_stream()
side = torch.cuda.Stream()
lo = torch.cuda.Stream(priority=0)
hi = torch.cuda.Stream(priority=-1)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n):
        fn()
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def both_two_streams(ws, cs):
    def f():
        ws.wait_stream(main)
        cs.wait_stream(main)
        wgrad(ws)
        chain(cs)
        main.wait_stream(ws)
        main.wait_stream(cs)
    return f


print("chain alone (20 layer norms)        %7.1f us" % timeit(lambda: chain(main)))
print("weight gradient alone               %7.1f us" % timeit(lambda: wgrad(main)))
print("both, one stream                    %7.1f us" % timeit(lambda: (wgrad(main), chain(main))))
print("both, wgrad on a side stream        %7.1f us" % timeit(both_two_streams(side, main)))
print("both, wgrad low / chain high prio   %7.1f us" % timeit(both_two_streams(lo, hi)))
print("both, chain first then wgrad (side) %7.1f us" % timeit(lambda: (side.wait_stream(main), chain(main), wgrad(side), main.wait_stream(side))))
