#!/usr/bin/env python
"""Multi-wave forward of the fused SeparableFCTP (csrc/sfcy.hip) against the one-wave kernel (csrc/sfcx.hip) on the same inputs:
   python tools/bench_sfcy.py [E] [modes, e.g. 0,1]
max |difference| of the outputs (expected 0: same plane products in the same order) and us / call of both, for the QM9 shapes
sep_act (per-edge weights, second consumer), sep_value (plain) and sep_value with the gate folded in."""
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
MODES = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
L = _lib.load()


def timeit(fn, n=30):
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


def run(name, irr, sh_irr, out_irr, n2, use_w, gated=False):
    table = DtpTable(irr, sh_irr, irr)
    lay = RowLayout(out_irr)
    spec = ops.SfcSpec(table, lay, n2=n2)
    g = torch.Generator().manual_seed(0)
    S, G = 128, 96  # the gate of the QM9 model: 128 scalars, 64 + 32 gate scalars
    xdim = table.layout_in.dim + (G if gated else 0)
    x = torch.randn(E, xdim, generator=g).to(dev)
    M = torch.randn(E, table.m_numel, generator=g).to(dev)
    w = torch.randn(E, table.weight_numel, generator=g).to(dev) if use_w else None
    weight = (torch.randn(spec.weight_numel, generator=g) * 0.1).to(dev)
    weight2 = (torch.randn(spec.weight2_numel, generator=g) * 0.1).to(dev) if n2 else None
    bias = torch.randn(lay.mul_of(0), generator=g).to(dev)
    bias2 = torch.randn(n2, generator=g).to(dev) if n2 else None
    flops = sum(2.0 * E * (2 * l3 + 1) * K * ncat for (l3, K, _, ncat) in spec.degs)
    gin = _lib.EqfGateIn(S, G, 1.6791768, 1.8467055)
    for mode in MODES:
        packed = ops._sfc_pack(weight, weight2, spec, mode)
        PK = ctypes.c_void_p(packed.data_ptr())
        outs = {}
        for variant in (1, 2):
            L.eqf_sfcx_dev_set(2, variant)
            o1 = torch.full((E, lay.dim), float("nan"), device=dev)
            o2 = torch.full((E, n2), float("nan"), device=dev) if n2 else None
            if gated:
                fx = lambda: call("eqf_sfcx_fwd_gated", P(x), ctypes.byref(gin), P(M), P(w), table.c_ref, PK, P(bias), P(o1),  # noqa: E731
                                  lay.c_ref, E, mode, st())
            else:
                fx = lambda: call("eqf_sfcx_fwd", P(x), P(M), P(w), table.c_ref, PK, P(bias), P(bias2), P(o1), lay.c_ref,  # noqa: E731
                                  P(o2), n2, E, mode, st())
            fx()
            torch.cuda.synchronize()
            outs[variant] = (o1.clone(), o2.clone() if n2 else None)
            us = timeit(fx)
            print("%-16s mode %d %-10s %7.1f us  (%5.1f TFLOP/s)  finite %s" % (
                name, mode, "one-wave" if variant == 1 else "multi-wave", us, flops / us / 1e6,
                bool(torch.isfinite(o1).all().item())), flush=True)
        L.eqf_sfcx_dev_set(2, 0)
        d1 = (outs[1][0] - outs[2][0]).abs().max().item()
        d2 = (outs[1][1] - outs[2][1]).abs().max().item() if n2 else 0.0
        print("%-16s mode %d max |one-wave - multi-wave| = %.3e (out1)  %.3e (out2)   scale %.2e" % (
            name, mode, d1, d2, outs[1][0].abs().max().item()), flush=True)


QM9 = "128x0e+64x1e+32x2e"
SH = "1x0e+1x1e+1x2e"
run("sep_act", QM9, SH, "224x0e+64x1e+32x2e", 128, True)
run("sep_value", QM9, SH, QM9, 0, False)
run("sep_value_gated", QM9, SH, QM9, 0, False, gated=True)
if "--more" in sys.argv:
    run("oc20_l1", "256x0e+128x1e", "1x0e+1x1e", "256x0e+128x1e", 0, True)
