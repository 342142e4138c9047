#!/usr/bin/env python
"""(needs a development build: EQF_EXTRA_FLAGS="-DEQF_DEV_SWITCHES=1" python -m equiformer_amd.build --force)
Development aid: per-phase cycle counters and phase-switch timings of eqf_sfc_bwd_data at the bench size
(mask 128 selected the streamed-weight variant while the LDS-staged-weight experiment of round 2 was in the tree:
profiles/r02/r02_o_sfc_bwd_wlds_experiment.txt)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiformer_amd import lib as _lib, ops
from equiformer_amd.layout import DtpTable, RowLayout
from equiformer_amd.lib import call
E = 25354
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(name, irr, sh_irr, out_irr, n2, use_w):
    table = DtpTable(irr, sh_irr, irr); lay = RowLayout(out_irr); spec = ops.SfcSpec(table, lay, n2=n2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(E, table.layout_in.dim, generator=g).to(dev); M = torch.randn(E, table.m_numel, generator=g).to(dev)
    w = torch.randn(E, table.weight_numel, generator=g).to(dev) if use_w else None
    weight = torch.randn(spec.weight_numel, generator=g).to(dev)
    weight2 = torch.randn(spec.weight2_numel, generator=g).to(dev) if n2 else None
    d1 = torch.randn(E, lay.dim, generator=g).to(dev); d2 = torch.randn(E, n2, generator=g).to(dev) if n2 else None
    dx = torch.empty_like(x); dw = torch.empty_like(w) if use_w else None
    Wl = ops._ptr_array((d[0], weight.data_ptr() + 4 * o) for d, o in zip(spec.degs, spec.w_offs))
    b = lambda: call("eqf_sfc_bwd_data", P(x), P(M), P(w), table.c_ref, Wl, P(weight2), P(d1), lay.c_ref, P(d2), n2, P(dx), P(dw), None, E, st())
    L = _lib.load()
    for mask in (0,):
        L.eqf_sfc_debug_exp(mask)
        dbg = torch.zeros(8, dtype=torch.int64, device=dev)
        b(); torch.cuda.synchronize()
        L.eqf_sfc_debug_buffer(ctypes.c_void_p(dbg.data_ptr())); b(); torch.cuda.synchronize(); L.eqf_sfc_debug_buffer(None)
        d = dbg.cpu().tolist(); nb = (E + 31) // 32
        print("%-10s mask %3d bwd_data phase cycles / tile row: prologue %.0f staging %.0f mfma %.0f epilogue %.0f store %.0f" % ((name, mask) + tuple(v / nb for v in d[:5])), flush=True)
        for m2 in (0, 1, 2, 3):
            L.eqf_sfc_debug_exp(mask | m2)
            for _ in range(3): b()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): b()
            e.record(); torch.cuda.synchronize()
            print("   exp %d: %.1f us" % (m2, a.elapsed_time(e) * 100), flush=True)
    L.eqf_sfc_debug_exp(0)
run("sep_act", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True)
run("sep_value", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", 0, False)
