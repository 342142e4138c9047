#!/usr/bin/env python
"""Micro-benchmark of the split-precision SeparableFCTP kernels (csrc/sfcx.hip) at the bench shapes:
   python tools/bench_sfcx.py [E] [modes, e.g. 0,1] [--l3] [--dm]
us / call of forward, data gradient, weight gradient.  --l3: the L_max = 3 MD17 shapes as well; --dm: the force-evaluation
variant (d_coupling) of the data gradient.  A/B of kernel variants: EQF_LIB_VARIANT=<name> (equiformer_amd/build.py --variant)."""
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
for _a in sys.argv:  # --wgrad-variant=1 (one-wave kernel) / 2 (multi-wave, csrc/sfcw.hip); default 0 = the library's choice
    if _a.startswith("--wgrad-variant="):
        L.eqf_sfcx_dev_set(4, int(_a.split("=")[1]))
    if _a.startswith("--bwd-split-slots="):  # wave slots the path-split data gradient may take (0 = never split)
        v = int(_a.split("=")[1])
        L.eqf_sfcx_dev_set(9, 1 if v == 0 else 0)
        if v:
            L.eqf_sfcx_dev_set(10, v)
    if _a.startswith("--fwd-split="):  # at most this many waves per forward item on small graphs (1, 2, 4)
        L.eqf_sfcx_dev_set(11, int(_a.split("=")[1]))
    if _a.startswith("--wgrad-rounds="):
        L.eqf_sfcx_dev_set(5, int(_a.split("=")[1]))
    if _a.startswith("--wgrad-order="):
        L.eqf_sfcx_dev_set(6, int(_a.split("=")[1]))


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


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def run(name, irr, sh_irr, out_irr, n2, use_w, want_dM=False):
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
    dWl = ops._ptr_array((d[0], dweight.data_ptr() + 4 * o) for d, o in zip(spec.degs, spec.w_offs))
    flops = sum(2.0 * E * (2 * l3 + 1) * K * ncat for (l3, K, _, ncat) in spec.degs)
    for mode in MODES:
        packed = ops._sfc_pack(weight, weight2, spec, mode)
        PK = ctypes.c_void_p(packed.data_ptr())

        dx = torch.full_like(x, float("nan"))
        dw = torch.full_like(w, float("nan")) if use_w else None
        dM = torch.zeros_like(M) if want_dM else None
        bw = lambda: call("eqf_sfcx_bwd_data", P(x), P(M), P(w), table.c_ref, PK, P(d1), lay.c_ref, P(d2), n2,  # noqa: E731
                          P(dx), P(dw), P(dM), E, mode, st())
        us = timeit(bw)
        fin = all(torch.isfinite(a).all().item() for a in (dx, dw) if a is not None)
        print("%-10s mode %d%s bwd_data   %7.1f us  (%5.1f TFLOP/s)  finite %s"
              % (name, mode, " dM" if want_dM else "", us, flops / us / 1e6, fin), flush=True)
        if "--classes" in sys.argv:  # the launch with only the items of one input degree: that class's longest item
            for dd in (1, 3, 5, 7):
                L.eqf_sfcx_dev_set(1, dd)
                print("%-10s            items of d1 = %d only: %7.1f us" % (name, dd, timeit(bw)), flush=True)
            L.eqf_sfcx_dev_set(1, 0)
        if want_dM:
            continue
        fx = lambda: call("eqf_sfcx_fwd", P(x), P(M), P(w), table.c_ref, PK, None, None, P(o1), lay.c_ref, P(o2), n2, E,  # noqa: E731
                          mode, st())
        wx = lambda: call("eqf_sfcx_bwd_weight", P(x), P(M), P(w), table.c_ref, P(d1), lay.c_ref, P(d2), n2, dWl,  # noqa: E731
                          P(dweight2), E, mode, st())
        for tag, fn in (("fwd", fx), ("bwd_weight", wx)):
            us = timeit(fn)
            print("%-10s mode %d %-10s %7.1f us  (%5.1f TFLOP/s)" % (name, mode, tag, us, flops / us / 1e6), flush=True)
        if "--wtypes" in sys.argv:  # the multi-wave weight gradient with the workgroups of one type only (index after the cost sort)
            for ty in range(24):
                L.eqf_sfcx_dev_set(7, ty)
                print("%-10s            wgrad workgroups of type %2d only: %7.1f us" % (name, ty, timeit(wx)), flush=True)
            L.eqf_sfcx_dev_set(7, -1)
        if "--wclasses" in sys.argv:  # the weight-gradient launch with the items of one (input degree, output degree) class only
            for di in (1, 3, 5):
                for do in (1, 3, 5):
                    L.eqf_sfcx_dev_set(3, 1 + 10 * di + do)
                    print("%-10s            wgrad items of (d1, d3) = (%d, %d) only: %7.1f us" % (name, di, do, timeit(wx)), flush=True)
            L.eqf_sfcx_dev_set(3, 0)


run("sep_act", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True)
run("sep_value", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", 0, False)
if "--dm" in sys.argv:
    run("sep_act", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True, want_dM=True)
if "--l3" in sys.argv:  # graph_attention_transformer_nonlinear_exp_l3_md17: 5 aspirin frames = 1 742 edges in the bench
    I3, S3 = "128x0e+64x1e+64x2e+32x3e", "1x0e+1x1e+1x2e+1x3e"
    run("l3_act", I3, S3, "288x0e+64x1e+64x2e+32x3e", 128, True)
    run("l3_act", I3, S3, "288x0e+64x1e+64x2e+32x3e", 128, True, want_dM=True)
    run("l3_value", I3, S3, I3, 0, False, want_dM=True)
