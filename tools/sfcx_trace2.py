#!/usr/bin/env python
"""Clock marks of the multi-wave data gradient (dev build with -DEQF_XTRACE=1): per traced workgroup and wave, cycles between
the marks of csrc/sfcx_bwd2.hip (XB2_MARK).    python tools/sfcx_trace2.py [sep_act|sep_value] [mode]"""
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
NWG, NW = 32, 4
trace = torch.zeros(NWG * NW * 64, dtype=torch.int64, device=dev)
d1 = torch.randn(E, lay.dim, generator=g).to(dev)
d2 = torch.randn(E, n2, generator=g).to(dev) if n2 else None
run = lambda: ops._sfc_bwd_data(x, M, w, weight, weight2, d1, d2, spec, False, mode, packed)  # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
L.eqf_sfcx_dev_set_trace2.argtypes = [ctypes.c_void_p]
L.eqf_sfcx_dev_set_trace2(ctypes.c_void_p(trace.data_ptr()))
run()
torch.cuda.synchronize()
L.eqf_sfcx_dev_set_trace2(None)
t = trace.cpu().view(NWG, NW, 64)
NAMES = {1: "start", 3: "coupling->LDS", 4: "planes->LDS", 5: "barrier", 11: "item d1=1", 13: "item d1=3", 15: "item d1=5",
         20: "x arrived", 30: "path: w ready", 31: "matrix loop", 32: "contraction+dw", 40: "dx stores", 50: "done"}
print("%s mode %d: multi-wave data gradient, cycles since the previous mark (s_memtime, 100 MHz x ... see header)" % (name, mode))
tot = {}
for b in range(NWG):
    for wv in range(NW):
        marks = [(int(v) >> 56 & 0xff, int(v) & ((1 << 56) - 1)) for v in t[b, wv] if int(v) != 0]
        if not marks:
            continue
        t0 = marks[0][1]
        line, prev = [], t0
        for tag, tm in marks[1:]:
            dt = tm - prev
            prev = tm
            tot.setdefault(tag, []).append(dt)
            line.append("%s %d" % (NAMES.get(tag, str(tag)), dt))
        if b < 4:
            print("wg %2d wave %d total %7d : %s" % (b, wv, marks[-1][1] - t0, " | ".join(line)))
print("mean cycles per occurrence (count):")
for tag in sorted(tot):
    v = tot[tag]
    print("  %-16s %9.0f  x %5.1f per wave" % (NAMES.get(tag, str(tag)), sum(v) / len(v), len(v) / (NWG * NW)))
ends = [max(int(v) & ((1 << 56) - 1) for v in t[b].reshape(-1) if int(v) != 0) - min(int(v) & ((1 << 56) - 1) for v in t[b].reshape(-1) if int(v) != 0) for b in range(NWG) if (t[b] != 0).any()]
print("workgroup duration (cycles of the s_memtime clock): mean %.0f  min %d  max %d over %d workgroups" % (sum(ends) / len(ends), min(ends), max(ends), len(ends)))
