"""Multi-wave forward of the fused SeparableFCTP (csrc/sfcy.hip: loader + compute waves, LDS-DMA rings) on the GPU.

The kernel multiplies the same bf16 planes in the same order as the one-wave forward of csrc/sfcx.hip, so the two must agree BIT
FOR BIT on every shape the multi-wave planner accepts -- with per-edge weights and a second consumer (sep_act), without (sep_value),
with the gate folded into the x rows, for edge counts that are not multiples of the 128-edge workgroup tile and for all three
matrix modes -- and both stay within the mode's tolerance of the exact-fp32 kernels (eqf_sfc_*, themselves pinned against the
oracle by tests/test_gpu_ops.py / test_gpu_fullsize.py).  Shapes outside its tables (degree-3 models) fall through to the one-wave
kernel without an error.  [ref: SeparableFCTP.forward, nets/graph_attention_transformer.py:234-248]"""
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


def _forward(case, E, mode, variant, seed=0):
    """out1, out2 of the fused forward with the kernel selected by the development switch (1 one-wave, 2 multi-wave)."""
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
    weight = r(spec.weight_numel) * 0.1
    weight2 = r(spec.weight2_numel) * 0.1 if n2 else None
    bias, bias2 = r(lay.mul_of(0)), (r(n2) if n2 else None)
    o1 = torch.full((E, lay.dim), float("nan"), device=dev)
    o2 = torch.full((E, n2), float("nan"), device=dev) if n2 else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = _lib.load()
    if mode is None:  # exact-fp32 kernels (no gate there: apply it first)
        assert not gated
        return ops._sfc_fwd(x, M, w, weight, bias, weight2, bias2, spec, None)
    packed = ops._sfc_pack(weight, weight2, spec, mode)
    PK = ctypes.c_void_p(packed.data_ptr())
    L.eqf_sfcx_dev_set(2, variant)
    try:
        if gated:
            gin = _lib.EqfGateIn(S, G, 1.6791768, 1.8467055)
            call("eqf_sfcx_fwd_gated", P(x), ctypes.byref(gin), P(M), P(w), table.c_ref, PK, P(bias), P(o1), lay.c_ref, E, mode, st)
        else:
            call("eqf_sfcx_fwd", P(x), P(M), P(w), table.c_ref, PK, P(bias), P(bias2), P(o1), lay.c_ref, P(o2), n2, E, mode, st)
        torch.cuda.synchronize()
    finally:
        L.eqf_sfcx_dev_set(2, 0)
    return o1, o2


@pytest.mark.parametrize("E", [37, 1000, 4097, 9001, 25354])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", ["qm9_sep_act", "qm9_sep_value", "qm9_sep_value_gated", "oc20_l1"])
def test_multi_wave_forward_equals_one_wave_forward(case, mode, E):
    """Bit for bit where the one-wave kernel runs one wave per item (from ~5 500 edges on).  On smaller graphs it splits every
    item over two waves and adds the two partial sums at the end -- the same plane products in another fp32 summation order: equal
    to 2e-6 of the output scale there."""
    a1, a2 = _forward(case, E, mode, 1)
    b1, b2 = _forward(case, E, mode, 2)
    assert torch.isfinite(b1).all() and (b2 is None or torch.isfinite(b2).all())
    if E >= 9000:
        assert torch.equal(a1, b1), (case, mode, E, (a1 - b1).abs().max().item())
        assert a2 is None or torch.equal(a2, b2), (case, mode, E)
    else:
        assert _rel(b1, a1) < 2e-6, (case, mode, E, _rel(b1, a1))
        assert a2 is None or _rel(b2, a2) < 2e-6, (case, mode, E)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", ["qm9_sep_act", "qm9_sep_value", "oc20_l1"])
def test_multi_wave_forward_against_exact_fp32_kernels(case, mode):
    E = 9000
    r1, r2 = _forward(case, E, None, 0)
    b1, b2 = _forward(case, E, mode, 2)
    e1 = _rel(b1, r1)
    e2 = _rel(b2, r2) if r2 is not None else 0.0
    print("%s mode %d E=%d multi-wave vs exact fp32: out1 %.1e out2 %.1e" % (case, mode, E, e1, e2))
    assert e1 < TOL[mode] and e2 < TOL[mode], (case, mode, e1, e2)


def test_shapes_outside_the_multi_wave_tables_fall_through():
    """Degree-3 models: the multi-wave planner refuses (EQF_E_UNSUPPORTED inside the library), the one-wave kernel serves."""
    a1, _ = _forward("md17_l3", 9000, 0, 1)
    b1, _ = _forward("md17_l3", 9000, 0, 2)
    assert torch.isfinite(b1).all() and torch.equal(a1, b1)


def test_multi_wave_forward_is_the_default_at_the_bench_size_and_bit_reproducible():
    """variant 0 (automatic) = the multi-wave kernel from EQF_Y_MIN_EDGES edges on; six launches give the same bits."""
    a1, a2 = _forward("qm9_sep_act", 25354, 0, 0, seed=3)
    b1, b2 = _forward("qm9_sep_act", 25354, 0, 2, seed=3)
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    for _ in range(5):
        c1, c2 = _forward("qm9_sep_act", 25354, 0, 0, seed=3)
        assert torch.equal(a1, c1) and torch.equal(a2, c2)
