"""Multi-wave weight gradient of the fused SeparableFCTP (csrc/sfcw.hip: the slabs of an output degree share one workgroup and
the bf16 planes of its d_out tiles) on the GPU.

It multiplies the same bf16 planes as the one-wave weight gradient of csrc/sfcx.hip and adds its partial sums atomically like it,
with other chunk boundaries: the two agree to the fp32 noise of that regrouping on every shape the planner accepts -- per-edge
weights and a second consumer (sep_act), without (sep_value), gate folded into the x rows, bias gradients taken along, edge counts
that are not multiples of 16 / 32 -- and both stay within the mode's tolerance of the exact-fp32 kernels (eqf_sfc_bwd_weight, pinned
against the oracle by tests/test_gpu_ops.py / test_gpu_fullsize.py).  Degree-3 models fall through to the one-wave kernel.
[ref: the weight gradients autograd derives for SeparableFCTP.forward, nets/graph_attention_transformer.py:234-248]"""
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

TOL = {0: 1e-4, 1: 3e-2, 2: 5e-6}
QM9, SH2 = "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e"
CASES = {
    "qm9_sep_act": (QM9, SH2, "224x0e+64x1e+32x2e", 128, True, False),
    "qm9_sep_value": (QM9, SH2, QM9, 0, False, False),
    "qm9_sep_value_gated": (QM9, SH2, QM9, 0, False, True),
    "oc20_l1": ("256x0e+128x1e", "1x0e+1x1e", "256x0e+128x1e", 0, True, False),
    "md17_l3": ("128x0e+64x1e+64x2e+32x3e", "1x0e+1x1e+1x2e+1x3e", "128x0e+64x1e+64x2e+32x3e", 0, True, False),
}
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _wgrad(case, E, mode, variant, seed=0, bias=False):
    """dweight, dweight2, dbias, dbias2 of the fused weight gradient, kernel selected by the development switch (1 one-wave, 2
    multi-wave); mode None = the exact-fp32 kernels."""
    irr, sh, out_irr, n2, use_w, gated = CASES[case]
    dev = torch.device("cuda:0")
    table, lay = DtpTable(irr, sh, irr), RowLayout(out_irr)
    spec = ops.SfcSpec(table, lay, n2=n2)
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    S, G = 128, 96  # the gate of the QM9 model: 128 scalars, 64 + 32 gate scalars
    x = r(E, table.layout_in.dim + (G if gated else 0))
    M = r(E, table.m_numel)
    w = r(E, table.weight_numel) if use_w else None
    d1, d2 = r(E, lay.dim), (r(E, n2) if n2 else None)
    dW = torch.zeros(spec.weight_numel, device=dev)
    dW2 = torch.zeros(spec.weight2_numel, device=dev) if n2 else None
    db = torch.zeros(lay.mul_of(0), device=dev) if bias else None
    db2 = torch.zeros(n2, device=dev) if (bias and n2) else None
    if mode is None:
        assert not gated
        ops._sfc_bwd_weight(x, M, w, d1, d2, spec, dW, dW2, None)
        torch.cuda.synchronize()
        return dW, dW2, d1[:, :lay.mul_of(0)].sum(0), (d2.sum(0) if n2 else None)
    dWl = ops._ptr_array((d[0], dW.data_ptr() + 4 * o) for d, o in zip(spec.degs, spec.w_offs))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = _lib.load()
    L.eqf_sfcx_dev_set(4, variant)
    try:
        if gated:
            gin = _lib.EqfGateIn(S, G, 1.6791768, 1.8467055)
            if bias:
                call("eqf_sfcx_bwd_weight_gated_bias", P(x), ctypes.byref(gin), P(M), P(w), table.c_ref, P(d1), lay.c_ref, dWl, P(db), E,
                     mode, st)
            else:
                call("eqf_sfcx_bwd_weight_gated", P(x), ctypes.byref(gin), P(M), P(w), table.c_ref, P(d1), lay.c_ref, dWl, E, mode, st)
        elif bias:
            call("eqf_sfcx_bwd_weight_bias", P(x), P(M), P(w), table.c_ref, P(d1), lay.c_ref, P(d2), n2, dWl, P(dW2), P(db), P(db2), E,
                 mode, st)
        else:
            call("eqf_sfcx_bwd_weight", P(x), P(M), P(w), table.c_ref, P(d1), lay.c_ref, P(d2), n2, dWl, P(dW2), E, mode, st)
        torch.cuda.synchronize()
    finally:
        L.eqf_sfcx_dev_set(4, 0)
    return dW, dW2, db, db2


@pytest.mark.parametrize("E", [37, 1000, 4097, 25354])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", ["qm9_sep_act", "qm9_sep_value", "qm9_sep_value_gated", "oc20_l1"])
def test_multi_wave_weight_gradient_equals_one_wave_weight_gradient(case, mode, E):
    a, a2, _, _ = _wgrad(case, E, mode, 1)
    b, b2, _, _ = _wgrad(case, E, mode, 2)
    assert torch.isfinite(b).all() and float(b.abs().max()) > 0
    # the same bf16 products; fp32 partial sums over other edge ranges, added atomically in another order
    assert _rel(b, a) < 3e-6, (case, mode, E, _rel(b, a))
    if a2 is not None:
        assert _rel(b2, a2) < 3e-6, (case, mode, E, _rel(b2, a2))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", ["qm9_sep_act", "qm9_sep_value", "oc20_l1"])
def test_multi_wave_weight_gradient_against_exact_fp32_kernels(case, mode):
    E = 9000
    r, r2, rb, rb2 = _wgrad(case, E, None, 0)
    b, b2, db, db2 = _wgrad(case, E, mode, 2, bias=True)
    e1 = _rel(b, r)
    e2 = _rel(b2, r2) if r2 is not None else 0.0
    print("%s mode %d E=%d multi-wave weight gradient vs exact fp32: dW %.1e dW2 %.1e" % (case, mode, E, e1, e2))
    assert e1 < TOL[mode] and e2 < TOL[mode], (case, mode, e1, e2)
    # the bias gradients are plain fp32 column sums of d_out, taken along by the tiles' owners
    assert _rel(db, rb) < 1e-5, _rel(db, rb)
    if db2 is not None:
        assert _rel(db2, rb2) < 1e-5, _rel(db2, rb2)


def test_gated_bias_variant_and_the_automatic_choice():
    """eqf_sfcx_bwd_weight_gated_bias through both kernels; variant 0 (automatic) = the multi-wave kernel from EQF_W_MIN_EDGES on."""
    a, _, ab, _ = _wgrad("qm9_sep_value_gated", 25354, 0, 1, bias=True)
    b, _, bb, _ = _wgrad("qm9_sep_value_gated", 25354, 0, 2, bias=True)
    c, _, cb, _ = _wgrad("qm9_sep_value_gated", 25354, 0, 0, bias=True)
    assert _rel(b, a) < 3e-6 and _rel(bb, ab) < 3e-6
    assert _rel(c, b) < 3e-6 and _rel(cb, bb) < 3e-6


def test_shapes_outside_the_multi_wave_tables_fall_through():
    """Degree-3 models: the multi-wave planner refuses (EQF_E_UNSUPPORTED inside the library), the one-wave kernel serves."""
    a, _, _, _ = _wgrad("md17_l3", 3000, 0, 1)
    b, _, _, _ = _wgrad("md17_l3", 3000, 0, 2)
    assert torch.isfinite(b).all() and _rel(b, a) < 3e-6
