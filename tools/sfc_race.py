#!/usr/bin/env python
"""Development aid: eqf_sfc_fwd at the bench size, split-precision step vs exact-fp32 step, run-to-run determinism and
the pattern (degree segment, column, edge tile) of any mismatch."""
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
STATS = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 0
MASKS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 0, 128, 256, 512, 1024, 32]
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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

    def fwd(mask):
        L.eqf_sfc_debug_exp(mask)
        o1 = torch.full((E, lay.dim), float("nan"), device=dev)
        o2 = torch.full((E, n2), float("nan"), device=dev) if n2 else None
        call("eqf_sfc_fwd", P(x), P(M), P(w), table.c_ref, Wl, None, P(weight2), None, P(o1), lay.c_ref, P(o2), n2, E, st())
        torch.cuda.synchronize()
        L.eqf_sfc_debug_exp(0)
        return o1 if o2 is None else torch.cat([o1, o2], 1)

    X6_BIT = 0 if L.eqf_sfc_debug_x6_default() else 64   # mask bit 64 selects the non-default matrix step
    ref = fwd(64 ^ X6_BIT)  # exact-fp32 MFMA step
    scale = ref.abs().max().item()
    if STATS:
        dbg = torch.zeros(8, dtype=torch.int64, device=dev)
        for mask in MASKS:
            nbad = nrows = 0
            dbg.zero_()
            if mask & (262144 | 524288 | 1048576 | 4194304):
                L.eqf_sfc_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
            for rep in range(STATS):
                a = fwd(mask ^ X6_BIT)  # masks are relative to the split-precision step
                bad = (a - ref).abs() > 1e-5 * scale
                if bad.any() and nbad < 3 and "--match" in sys.argv:
                    # is a bad output row some OTHER row's correct result (or a mix)?  compare with every row of its tile
                    rows = bad.any(1).nonzero().flatten()
                    for rr in rows[:6].tolist():
                        t0 = (rr // 64) * 64
                        cols = bad[rr].nonzero().flatten()
                        blk = ref[t0:t0 + 64][:, cols]
                        d = (blk - a[rr, cols][None]).abs().max(1).values / scale
                        best = int(d.argmin())
                        per_tile = [(int(c0), "%.1e" % ((a[rr, c0:c0 + 32] - ref[rr, c0:c0 + 32]).abs().max() / scale).item())
                                    for c0 in range(int(cols[0]) // 32 * 32, int(cols[-1]) + 1, 32)]
                        print("   row %d (in-tile %d): closest reference row of the tile = in-tile %d (max diff %.1e of scale); "
                              "a/ref first bad cols %s / %s; per 32-col tile max err %s"
                              % (rr, rr % 64, best, d[best].item(), [round(v, 3) for v in a[rr, cols[:4]].tolist()],
                                 [round(v, 3) for v in ref[rr, cols[:4]].tolist()], per_tile), flush=True)
                if bad.any() and nbad < 4:  # pattern of the first few failures
                    rows = bad.any(1).nonzero().flatten()
                    for rr in rows[:4].tolist():
                        cols = bad[rr].nonzero().flatten()
                        rel = ((a[rr] - ref[rr]).abs().max() / scale).item()
                        print("   bad row %d (tile %d, row-in-tile %d): %d bad columns %s..%s, max err %.2e of scale"
                              % (rr, rr // 64, rr % 64, cols.numel(), cols[:3].tolist(), cols[-3:].tolist(), rel), flush=True)
                nbad += int(bad.any())
                nrows += int(bad.any(1).sum())
            L.eqf_sfc_debug_buffer(None)
            print("%s mask %5d: %d of %d runs wrong, %d bad rows in total%s" % (name, mask, nbad, STATS, nrows,
                  ("; readback mismatches after the write barrier %d, at the end of the MFMA phase %d"
                   % (dbg[6].item(), dbg[7].item())) if mask & 262144 else "")
                  + (("; prefetched x registers differing from a fresh load %d, w registers %d" % (dbg[4].item(), dbg[5].item()))
                     if mask & 524288 else "")
                  + (("; generation expressions that differ when evaluated twice %d" % dbg[3].item()) if mask & 1048576 else "")
                  + (("; same registers, arithmetic twice: %d differ; same address, LDS read twice: %d differ"
                      % (dbg[2].item(), dbg[1].item())) if mask & 4194304 else ""),
                  flush=True)
        return
    for rep, mask in enumerate(MASKS):
        a = fwd(mask ^ X6_BIT)
        err = (a - ref).abs()
        bad = err > 1e-5 * scale
        print("%s mask %d: max err %.3e of scale, %d bad elements of %d (nan %d)"
              % (name, mask, err.max().item() / scale, int(bad.sum()), bad.numel(), int(torch.isnan(a).sum())), flush=True)
        if bad.any():
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            tiles = torch.unique(rows // 64)
            print("   bad rows %d (first %s) in %d tiles of 64 (first %s)" % (rows.numel(), rows[:8].tolist(), tiles.numel(),
                                                                              tiles[:12].tolist()))
            print("   rows within tile histogram (8 bins):", torch.histc((rows % 64).float(), 8, 0, 64).int().tolist())
            print("   bad columns %d: first %s last %s" % (cols.numel(), cols[:12].tolist(), cols[-6:].tolist()))


run("sep_act", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True)
run("sep_value", "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", 0, False)
