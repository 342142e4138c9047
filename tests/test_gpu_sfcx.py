"""Split-precision SeparableFCTP kernels (csrc/sfcx.hip, C ABI eqf_sfcx_*) on the GPU.

Operator level: forward, data gradient (dx, dw, d_coupling) and weight gradient against the exact-fp32 kernels
(eqf_sfc_*, themselves pinned against the oracle by tests/test_gpu_ops.py / test_gpu_fullsize.py) on the same inputs:
  mode 2 ("split6", 3 + 3 planes)  -> fp32-class: 5e-6 of the result scale
  mode 0 ("split", 2 + 3 planes)   -> 1e-4 of the result scale (measured ~1e-5)
  mode 1 ("bf16")                  -> 3e-2 (plain bf16 operands; the tolerance of BASELINE config #2, stated not assumed:
                                      the measured value is printed)
Model level: the QM9 model under ops.matrix_mode("split") meets the north-star bar (1e-4 vs the fp64 oracle) for energies
and every parameter gradient; under "bf16" the measured errors are printed and bounded.  Bit-reproducibility of the
forward and the data gradient (no atomics on those paths)."""
import ctypes
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import ops  # noqa: E402
from equiformer_amd.layout import DtpTable, RowLayout  # noqa: E402

pytestmark = pytest.mark.gpu

TOL = {0: 1e-4, 1: 3e-2, 2: 5e-6}
CASES = {
    "qm9_sep_act": ("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True),
    "qm9_sep_value": ("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", 0, False),
    "oc20_l1": ("256x0e+128x1e", "1x0e+1x1e", "256x0e+128x1e", 0, True),
    "md17_l3": ("128x0e+64x1e+64x2e+32x3e", "1x0e+1x1e+1x2e+1x3e", "128x0e+64x1e+64x2e+32x3e", 0, True),
    # a custom width whose data gradient exceeds the sfcx planner's slab table (14 input slabs > 12): that launch is served by
    # the exact-fp32 kernel, the other two by sfcx (round-3 advisor finding: it used to raise mid-backward)
    "wide_l2": ("256x0e+128x1e+64x2e", "1x0e+1x1e+1x2e", "256x0e+128x1e+64x2e", 0, True),
}


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _problem(case, E, seed=0):
    irr, sh, out_irr, n2, use_w = CASES[case]
    dev = _dev()
    table = DtpTable(irr, sh, irr)
    lay = RowLayout(out_irr)
    spec = ops.SfcSpec(table, lay, n2=n2)
    assert spec.supported and spec.x_ok
    if case == "wide_l2":
        assert spec.x_mask(0) == 5, spec.x_mask(0)
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, M = r(E, table.layout_in.dim), r(E, table.m_numel)
    w = r(E, table.weight_numel) if use_w else None
    weight = r(spec.weight_numel) * 0.1
    weight2 = r(spec.weight2_numel) * 0.1 if n2 else None
    bias, bias2 = r(lay.mul_of(0)), (r(n2) if n2 else None)
    d1, d2 = r(E, lay.dim), (r(E, n2) if n2 else None)
    return spec, x, M, w, weight, weight2, bias, bias2, d1, d2


def _run(spec, x, M, w, weight, weight2, bias, bias2, d1, d2, mode, want_dM=True):
    o1, o2 = ops._sfc_fwd(x, M, w, weight, bias, weight2, bias2, spec, mode)
    dx, dM, dw = ops._sfc_bwd_data(x, M, w, weight, weight2, d1, d2, spec, want_dM, mode)
    gW = torch.zeros_like(weight)
    gW2 = torch.zeros_like(weight2) if weight2 is not None else None
    ops._sfc_bwd_weight(x, M, w, d1, d2, spec, gW, gW2, mode)
    torch.cuda.synchronize()
    return dict(o1=o1, o2=o2, dx=dx, dM=dM, dw=dw, gW=gW, gW2=gW2)


@pytest.mark.parametrize("E", [1000, 37])
@pytest.mark.parametrize("mode", [2, 0, 1])
@pytest.mark.parametrize("case", sorted(CASES))
def test_operator_against_exact_fp32_kernels(case, mode, E):
    args = _problem(case, E)
    ref = _run(*args, None)
    got = _run(*args, mode)
    worst = {}
    for k, r in ref.items():
        if r is None:
            assert got[k] is None
            continue
        assert torch.isfinite(got[k]).all(), (case, mode, k)
        worst[k] = _rel(got[k], r)
    print("%s mode %d E=%d: %s" % (case, mode, E, {k: "%.1e" % v for k, v in worst.items()}))
    bad = {k: v for k, v in worst.items() if v > TOL[mode]}
    assert not bad, (case, mode, bad)


def test_bench_size_forward_and_data_gradient_are_bit_reproducible():
    args = _problem("qm9_sep_act", 25354, seed=3)
    a = _run(*args, 0, want_dM=False)
    for _ in range(5):
        b = _run(*args, 0, want_dM=False)
        for k in ("o1", "o2", "dx", "dw"):
            assert torch.equal(a[k], b[k]), k
    ref = _run(*args, None, want_dM=False)
    print("E = 25354 split vs fp32 kernels:", {k: "%.1e" % _rel(a[k], ref[k]) for k in ("o1", "o2", "dx", "dw", "gW", "gW2")})
    for k in ("o1", "o2", "dx", "dw", "gW", "gW2"):
        assert _rel(a[k], ref[k]) < 1e-4, k


def _qm9(mode, B=8):
    from equiformer_amd import nets
    from equiformer_amd.synthetic import qm9_like_batch
    from oracle import nets as onets
    dev = _dev()
    torch.manual_seed(0)
    ref = onets.graph_attention_transformer_nonlinear_l2("5x0e", 5.0).eval()
    mod = nets.model_entrypoint("graph_attention_transformer_nonlinear_l2")(irreps_in="5x0e", radius=5.0)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(dev).eval()
    d = qm9_like_batch(B, 18, side=6.5, seed=1)
    ref = ref.double()
    yr = ref(None, d["pos"].double(), d["batch"], d["z"])
    (yr.squeeze() - d["y"].double()).abs().mean().backward()
    with ops.matrix_mode(mode):
        y = mod(None, d["pos"].to(dev), d["batch"].to(dev), d["z"].to(dev))
        (y.squeeze() - d["y"].to(dev)).abs().mean().backward()
    e_err = _rel(y.detach().cpu(), yr.detach())
    g_err = max(_rel(p.grad.cpu(), q.grad) for (n, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters())
                if q.grad is not None and q.grad.abs().max() > 0)
    return e_err, g_err


@pytest.mark.parametrize("case", ["qm9_sep_act", "qm9_sep_value"])
def test_bench_size_properties(case):
    """Size-independent properties at the BASELINE batch (E = 25 354), split mode: the forward is linear in x (bias aside),
    the data gradient is linear in d_out, <d_out, F(x)> = <dx, x> (the data gradient is the transpose of the forward in x),
    and the weight gradient is the transpose of the forward in W: <d_out, F(x; W)> = <gW, W>."""
    E = 25354
    spec, x, M, w, weight, weight2, bias, bias2, d1, d2 = _problem(case, E, seed=5)
    g = torch.Generator().manual_seed(9)
    x2 = torch.randn(x.shape, generator=g).to(x.device)
    zb, zb2 = torch.zeros_like(bias), (torch.zeros_like(bias2) if bias2 is not None else None)
    f = lambda xx: ops._sfc_fwd(xx, M, w, weight, zb, weight2, zb2, spec, 0)  # noqa: E731
    a, b, ab = f(x), f(x2), f(x + x2)
    for k in range(2):
        if a[k] is None:
            continue
        assert _rel(a[k] + b[k], ab[k]) < 2e-5, (case, k, _rel(a[k] + b[k], ab[k]))
    dx1, _, dw1 = ops._sfc_bwd_data(x, M, w, weight, weight2, d1, d2, spec, False, 0)
    dx2, _, _ = ops._sfc_bwd_data(x, M, w, weight, weight2, 2.0 * d1, None if d2 is None else 2.0 * d2, spec, False, 0)
    assert _rel(dx2, 2.0 * dx1) < 1e-6  # scaling by two is exact in every plane
    lhs = (d1.double() * a[0].double()).sum() + (0.0 if d2 is None else (d2.double() * a[1].double()).sum())
    rhs = (dx1.double() * x.double()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 2e-5, (case, float(lhs), float(rhs))
    gW = torch.zeros_like(weight)
    gW2 = torch.zeros_like(weight2) if weight2 is not None else None
    ops._sfc_bwd_weight(x, M, w, d1, d2, spec, gW, gW2, 0)
    rhs_w = (gW.double() * weight.double()).sum() + (0.0 if gW2 is None else (gW2.double() * weight2.double()).sum())
    assert abs(lhs - rhs_w) / abs(lhs) < 2e-5, (case, float(lhs), float(rhs_w))
    if w is not None:  # and in the per-edge weights: <d_out, F> = <dw, w>
        rhs_e = (dw1.double() * w.double()).sum()
        assert abs(lhs - rhs_e) / abs(lhs) < 2e-5, (case, float(lhs), float(rhs_e))


def test_qm9_model_split_mode_meets_the_north_star_bar():
    e, g = _qm9("split")
    print("QM9 model, matrix mode split: energy rel err %.2e, worst parameter-gradient rel err %.2e" % (e, g))
    assert e < 1e-4 and g < 1e-4
    e32, g32 = _qm9("fp32")
    print("QM9 model, matrix mode fp32 : energy rel err %.2e, worst parameter-gradient rel err %.2e" % (e32, g32))
    assert e32 < 1e-4 and g32 < 1e-4


def test_qm9_model_bf16_mode_stated_tolerance():
    """BASELINE config #2: bf16 operands on the matrix cores (fp32 storage, fp32 accumulation, fp32 layer norm / softmax /
    radial basis as the reference pins them, nets/layer_norm.py:89).  CPU emulation of the same arithmetic
    (tools/split_model_error.py): energies 4e-3, gradients 2e-2.  Since round 4 the per-degree linears and the radial MLPs
    run in bf16 as well in this mode (csrc/gemmx.hip); the bench batch is checked in tests/test_gpu_fullsize.py."""
    e, g = _qm9("bf16")
    print("QM9 model, matrix mode bf16: energy rel err %.2e, worst parameter-gradient rel err %.2e" % (e, g))
    assert e < 1e-2 and g < 5e-2


@pytest.mark.parametrize("E", [1000, 37])
@pytest.mark.parametrize("mode", ["split", "bf16"])
def test_gate_folded_into_the_operator(mode, E):
    """eqf_sfcx_*_gated: the Gate in front of sep_value applied where x is loaded, its backward in the data gradient's
    epilogue [ref: nets/graph_attention_transformer.py:494-496, nets/fast_activation.py:132-148] -- against the separate
    gate kernels followed by the plain operator, same arithmetic mode: forward, d x_raw (scalars, gates, gated parts),
    d coupling, weight and bias gradients."""
    from equiformer_amd import so3
    from equiformer_amd.irreps import Irreps
    dev = _dev()
    irr, sh, out_irr = "128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e"
    table = DtpTable(irr, sh, irr)
    lay = RowLayout(out_irr)
    spec = ops.SfcSpec(table, lay, n2=0)
    gated_layout = RowLayout("64x1e+32x2e")
    S, G = 128, 96
    g = torch.Generator().manual_seed(E)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x_raw = r(E, S + G + gated_layout.dim).requires_grad_(True)
    M = r(E, table.m_numel).requires_grad_(True)
    weight = (r(spec.weight_numel) * 0.1).requires_grad_(True)
    bias = r(lay.mul_of(0)).requires_grad_(True)
    c = r(E, lay.dim)
    gate = (S, gated_layout, so3.C_SILU, so3.C_SIGMOID)
    res = {}
    with ops.matrix_mode(mode):
        assert ops.sep_fctp_gated_ok(spec, x_raw.shape[1], S, gated_layout)
        for fused in (True, False):
            if fused:
                y = ops.sep_fctp_gated(x_raw, M, None, weight, bias, spec, gate)
            else:
                y = ops.sep_fctp(ops.gate(x_raw, *gate), M, None, weight, bias, spec)
            grads = torch.autograd.grad((y * c).sum(), [x_raw, M, weight, bias])
            res[fused] = [y.detach()] + [t.detach() for t in grads]
    names = ["out", "d x_raw", "d coupling", "d weight", "d bias"]
    errs = {n: _rel(a, b) for n, a, b in zip(names, res[True], res[False])}
    print("gate folded, mode %s E=%d: %s" % (mode, E, {k: "%.1e" % v for k, v in errs.items()}))
    # same plane arithmetic on both sides; the differences are the last bits of the gated values (__expf vs expf, order of the
    # products), which move a few of their bf16 plane roundings
    assert all(v < (1e-5 if mode == "split" else 5e-3) for v in errs.values()), errs
