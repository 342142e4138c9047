"""Path-split launch of the fused SeparableFCTP data gradient on small graphs (csrc/sfcx.hip: sfcx_bwd_ps_kernel, round 6).

On the graphs of MD17 (55-100 edge tiles x 7 slab groups for 1 024 SIMDs) a launch takes as long as its longest item.  From round 6
on two waves share an item there: each runs every other path of the slab (its own dw / d_coupling), wave 1 hands its dx sums over
through LDS.  dw is bit-identical to the unsplit launch, dx differs by the fp32 rounding of a regrouped sum of <= 6 path terms,
d_coupling is added atomically in both.  Also the force evaluation (d_coupling requested), the gate folded in, and graphs too large
for the split (the planner leaves them alone: identical bits)."""
import ctypes
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import lib as _lib, ops  # noqa: E402
from equiformer_amd.layout import DtpTable, RowLayout  # noqa: E402
from equiformer_amd.lib import call  # noqa: E402

pytestmark = pytest.mark.gpu

QM9, SH2 = "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e"
CASES = {
    "md17_sep_act": (QM9, SH2, "224x0e+64x1e+32x2e", 128, True, False),
    "md17_sep_value": (QM9, SH2, QM9, 0, False, False),
    "md17_sep_value_gated": (QM9, SH2, QM9, 0, False, True),
}
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _bwd(case, E, mode, split, want_dM, seed=0):
    irr, sh, out_irr, n2, use_w, gated = CASES[case]
    dev = torch.device("cuda:0")
    table, lay = DtpTable(irr, sh, irr), RowLayout(out_irr)
    spec = ops.SfcSpec(table, lay, n2=n2)
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x = r(E, table.layout_in.dim + (96 if gated else 0))
    M = r(E, table.m_numel)
    w = r(E, table.weight_numel) if use_w else None
    weight = r(spec.weight_numel) * 0.1
    weight2 = r(spec.weight2_numel) * 0.1 if n2 else None
    d1, d2 = r(E, lay.dim), (r(E, n2) if n2 else None)
    dx = torch.full_like(x, float("nan"))
    dw = torch.full_like(w, float("nan")) if use_w else None
    dM = torch.zeros_like(M) if want_dM else None
    packed = ops._sfc_pack(weight, weight2, spec, mode)
    PK = ctypes.c_void_p(packed.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = _lib.load()
    L.eqf_sfcx_dev_set(9, 0 if split else 1)
    try:
        if gated:
            gin = _lib.EqfGateIn(128, 96, 1.6791768, 1.8467055)
            call("eqf_sfcx_bwd_data_gated", P(x), ctypes.byref(gin), P(M), P(w), table.c_ref, PK, P(d1), lay.c_ref, P(dx), P(dw), P(dM), E,
                 mode, st)
        else:
            call("eqf_sfcx_bwd_data", P(x), P(M), P(w), table.c_ref, PK, P(d1), lay.c_ref, P(d2), n2, P(dx), P(dw), P(dM), E, mode, st)
        torch.cuda.synchronize()
    finally:
        L.eqf_sfcx_dev_set(9, 0)
    return dx, dw, dM


@pytest.mark.parametrize("E", [37, 1000, 1742, 2790])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", sorted(CASES))
def test_path_split_data_gradient_equals_the_unsplit_launch(case, mode, E):
    for want_dM in ((False,) if "gated" in case else (False, True)):
        ax, aw, aM = _bwd(case, E, mode, False, want_dM)
        bx, bw, bM = _bwd(case, E, mode, True, want_dM)
        assert torch.isfinite(bx).all() and float(bx.abs().max()) > 0
        assert _rel(bx, ax) < 3e-6, (case, mode, E, _rel(bx, ax))  # (same products in every mode; the regrouped sum is fp32)
        if aw is not None:
            assert torch.equal(aw, bw), (case, mode, E)
        if want_dM:
            assert _rel(bM, aM) < 1e-5, (case, mode, E, _rel(bM, aM))


def test_large_graphs_are_left_alone():
    """7 groups x 793 tiles would need 11 102 wave slots: the planner does not split, the bits are those of the unsplit launch."""
    ax, aw, _ = _bwd("md17_sep_act", 25354, 0, False, False)
    bx, bw, _ = _bwd("md17_sep_act", 25354, 0, True, False)
    assert torch.equal(ax, bx) and torch.equal(aw, bw)
