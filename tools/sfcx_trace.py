#!/usr/bin/env python
"""Per-step clock samples of the sfcx forward kernel (dev build of csrc/sfcx.hip with -DEQF_XTRACE=1 installed as the
library): for the first workgroups of one XCD, the cycles a step waits for its operands and the cycles its generation +
matrix instructions take once they are there.
The instrumented waves are serialised by the samples; the other ~4700 waves run normally beside them.
   python tools/sfcx_trace.py [sep_act|sep_value] [mode]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib as _lib, ops  # noqa: E402
from equiformer_amd.layout import DtpTable, RowLayout  # noqa: E402

CASES = {"sep_act": ("224x0e+64x1e+32x2e", 128, True), "sep_value": ("128x0e+64x1e+32x2e", 0, False)}
name = sys.argv[1] if len(sys.argv) > 1 else "sep_act"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
which = sys.argv[3] if len(sys.argv) > 3 else "fwd"
E = 25354
dev = torch.device("cuda:0")
irr, sh = "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e"
out_irr, n2, use_w = CASES[name]
table, lay = DtpTable(irr, sh, irr), RowLayout(out_irr)
spec = ops.SfcSpec(table, lay, n2=n2)
g = torch.Generator().manual_seed(0)
x = torch.randn(E, table.layout_in.dim, generator=g).to(dev)
M = torch.randn(E, table.m_numel, generator=g).to(dev)
w = torch.randn(E, table.weight_numel, generator=g).to(dev) if use_w else None
weight = torch.randn(spec.weight_numel, generator=g).to(dev)
weight2 = torch.randn(spec.weight2_numel, generator=g).to(dev) if n2 else None
packed = ops._sfc_pack(weight, weight2, spec, mode)
L = _lib.load()
trace = torch.zeros(64 * 64, dtype=torch.int64, device=dev)
d1 = torch.randn(E, lay.dim, generator=g).to(dev)
d2 = torch.randn(E, n2, generator=g).to(dev) if n2 else None
if which == "fwd":
    run = lambda: ops._sfc_fwd(x, M, w, weight, None, weight2, None, spec, mode, packed)
else:
    run = lambda: ops._sfc_bwd_data(x, M, w, weight, weight2, d1, d2, spec, False, mode, packed)
for _ in range(3):
    run()
torch.cuda.synchronize()
L.eqf_sfcx_dev_set_trace.argtypes = [ctypes.c_void_p]
L.eqf_sfcx_dev_set_trace(ctypes.c_void_p(trace.data_ptr()))
run()
torch.cuda.synchronize()
L.eqf_sfcx_dev_set_trace(None)
t = trace.cpu().view(64, 64)
if which != "fwd":
    # data gradient: non-serialising phase marks (see the decode below)
    print("%s mode %d data gradient: per traced workgroup (item = 32 edges x 32-channel slab), cycles per phase" % (name, mode))
    agg = {}
    for b in range(64):
        h = int(t[b, 0])
        D1, npath = h >> 32, h & 0xffffffff
        ts = [int(v) for v in t[b, 1:] if int(v) != 0]
        if len(ts) != 2 + 3 * npath:
            continue
        # marks: item start | per path: coupling block staged, matrix loop issued, contraction + dw stores issued | end
        dt = [b_ - a_ for a_, b_ in zip(ts[:-1], ts[1:])]
        stage, loop, contr = dt[0:-1:3], dt[1:-1:3], dt[2:-1:3]
        tail = dt[-1]
        agg.setdefault((D1, npath), []).append((sum(stage), sum(loop), sum(contr), tail, ts[-1] - ts[0]))
        if b < 10:
            print("wg %2d d1 %d paths %d: staging %s  matrix loop %s  contraction %s  dx stores %d  total %d"
                  % (b, D1, npath, stage, loop, contr, tail, ts[-1] - ts[0]))
    print("item type (d1, paths): mean cycles in staging | matrix loop | contraction (incl. MFMA drain) | dx stores | total")
    for key in sorted(agg):
        v = agg[key]
        n = len(v)
        print("  d1 %d paths %d: %7.0f | %7.0f | %7.0f | %6.0f | %7.0f   (n=%d)"
              % (key + tuple(sum(c[k] for c in v) / n for k in range(5)) + (n,)))
    sys.exit(0)
print("%s mode %d: per workgroup (item): degree index, d3, first column tile, steps; per step cycles (s_memtime, shader clock) "
      "waiting for the operands | generating + multiplying once they are there" % (name, mode))
tot = {}
for b in range(64):
    h = int(t[b, 0])
    di, ct0, d3 = (h >> 32) & 0xffff, h & 0xffffffff, h >> 48
    steps = [(int(v) & 0xffffffff, int(v) >> 32) for v in t[b, 1:] if int(v) != 0]
    if not steps:
        continue
    wait = [a for a, _ in steps]
    comp = [c for _, c in steps]
    tot.setdefault((di, ct0), []).append((sum(wait), sum(comp), len(steps)))
    if b < 12:
        print("wg %2d deg %d d3 %d ct0 %2d steps %2d  wait %s" % (b, di, d3, ct0, len(steps), wait))
        print("%39s compute %s" % ("", comp))
print("item type (degree, ct0): mean over the traced workgroups of sum(wait), sum(compute) [cycles], steps")
for key in sorted(tot):
    v = tot[key]
    n = len(v)
    print("  deg %d ct0 %2d: wait %7.0f  compute %7.0f  steps %d  (n=%d)"
          % (key[0], key[1], sum(a for a, _, _ in v) / n, sum(c for _, c, _ in v) / n, v[0][2], n))
