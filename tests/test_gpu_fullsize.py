"""Parity at BASELINE.json's full size (the bench workload: 128 molecules x 18 atoms, E ~ 25 k edges):

  * the fused SeparableFCTP kernels against the un-fused composition at E = 25 354, forward and every gradient,
    plus a run-to-run determinism stress of the forward (round 1 shipped a forward that was only tested at E <= 333
    and produced different results from run to run at E = 25 k: tools/sfc_race.py);
  * the whole model against the fp64 CPU oracle on the 128-molecule batch: energies <= 1e-4 (north_star), parameter
    gradients of the L1 training loss <= 1e-4 of the largest gradient entry of the tensor;
  * bit-equal energies for the same input twice.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))  # fullsize.py (fixture format)

from oracle import e3 as oe3
from oracle import nets as onets

pytestmark = pytest.mark.gpu

E_BENCH = 25354
SHAPES = [("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "224x0e+64x1e+32x2e", 128, True),   # sep_act + sep_alpha
          ("128x0e+64x1e+32x2e", "1x0e+1x1e+1x2e", "128x0e+64x1e+32x2e", 0, False)]    # sep_value


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _sfc_inputs(irr, sh_irr, out_irr, n2, use_w, E, dev):
    from equiformer_amd import ops
    from equiformer_amd.layout import DtpTable, RowLayout
    table = DtpTable(irr, sh_irr, irr)
    lay_out = RowLayout(out_irr)
    spec = ops.SfcSpec(table, lay_out, n2=n2)
    assert spec.supported
    lmax = len(oe3.Irreps(sh_irr)) - 1
    g = torch.Generator().manual_seed(21)
    x = torch.randn(E, table.layout_in.dim, generator=g).to(dev).requires_grad_(True)
    sh = oe3.spherical_harmonics(lmax, torch.randn(E, 3, generator=g, dtype=torch.float64)).float().to(dev).requires_grad_(True)
    w = torch.randn(E, table.weight_numel, generator=g).to(dev).requires_grad_(True) if use_w else None
    weight = (torch.randn(spec.weight_numel, generator=g) / 16).to(dev).requires_grad_(True)
    weight2 = (torch.randn(spec.weight2_numel, generator=g) / 16).to(dev).requires_grad_(True) if n2 else None
    bias = torch.randn(lay_out.mul_of(0), generator=g).to(dev).requires_grad_(True)
    bias2 = torch.randn(n2, generator=g).to(dev).requires_grad_(True) if n2 else None
    return table, lay_out, spec, x, sh, w, weight, weight2, bias, bias2, g


@pytest.mark.parametrize("irr,sh_irr,out_irr,n2,use_w", SHAPES)
def test_sfc_bench_size_matches_unfused(irr, sh_irr, out_irr, n2, use_w):
    from equiformer_amd import ops
    dev = _dev()
    table, lay_out, spec, x, sh, w, weight, weight2, bias, bias2, g = _sfc_inputs(irr, sh_irr, out_irr, n2, use_w,
                                                                                 E_BENCH, dev)
    M = ops.dtp_coupling(sh, table)
    outs = ops.sep_fctp(x, M, w, weight, bias, spec, weight2=weight2, bias2=bias2)
    outs = list(outs) if n2 else [outs]
    mid = ops.dtp(x, ops.dtp_coupling(sh, table), w, table)
    refs = [ops.irreps_linear(mid, weight, bias, ops.LinearSpec(table.layout_out, lay_out))]
    if n2:
        K0 = spec.degs[0][1]
        refs.append(mid[:, :K0] @ weight2.view(K0, n2) + bias2)
    for o, r in zip(outs, refs):
        assert _rel(o, r) < 1e-5
    cot = [torch.randn(o.shape, generator=g).to(dev) for o in outs]
    ins = [t for t in (x, sh, w, bias, bias2, weight, weight2) if t is not None]
    ga = torch.autograd.grad(outs, ins, cot)
    gb = torch.autograd.grad(refs, ins, cot)
    for i, (a, b) in enumerate(zip(ga, gb)):
        assert _rel(a, b) < 3e-5, (i, _rel(a, b))


@pytest.mark.parametrize("irr,sh_irr,out_irr,n2,use_w", SHAPES)
def test_sfc_bench_size_is_deterministic(irr, sh_irr, out_irr, n2, use_w):
    """40 launches of the forward and of the data gradient (neither uses atomics) are bit-identical."""
    from equiformer_amd import ops
    dev = _dev()
    table, lay_out, spec, x, sh, w, weight, weight2, bias, bias2, g = _sfc_inputs(irr, sh_irr, out_irr, n2, use_w,
                                                                                 E_BENCH, dev)
    with torch.no_grad():
        M = ops.dtp_coupling(sh, table)
        d1 = torch.randn(E_BENCH, lay_out.dim, generator=g).to(dev)
        d2 = torch.randn(E_BENCH, n2, generator=g).to(dev) if n2 else None
        first = None
        for rep in range(40):
            o1, o2 = ops._sfc_fwd(x, M, w, weight, bias, weight2, bias2, spec)
            dx, _, dw = ops._sfc_bwd_data(x, M, w, weight, weight2, d1, d2, spec, False)
            cur = [t for t in (o1, o2, dx, dw) if t is not None]
            if first is None:
                first = [t.clone() for t in cur]
            else:
                for k, (a, b) in enumerate(zip(cur, first)):
                    assert torch.equal(a, b), (rep, k, float((a - b).abs().max()))


@pytest.mark.parametrize("irr,sh_irr,out_irr,n2,use_w", SHAPES)
def test_sfc_split_precision_step_is_deterministic_and_accurate(irr, sh_irr, out_irr, n2, use_w):
    """The forward's OTHER matrix step (bf16 matrix cores on 3-way split fp32 operands, development switch 64 of
    include/equiformer_hip_dev.h).  It was the source of round 1's run-to-run differences at this size: packed-FP32 VALU
    instructions beside the bf16 MFMAs of a co-resident wave.  Kept honest here: 60 launches bit-equal,
    and within 1e-5 of the exact-fp32 step."""
    from equiformer_amd import lib, ops
    dev = _dev()
    table, lay_out, spec, x, sh, w, weight, weight2, bias, bias2, g = _sfc_inputs(irr, sh_irr, out_irr, n2, use_w,
                                                                                 E_BENCH, dev)
    L = lib.load()
    with torch.no_grad():
        M = ops.dtp_coupling(sh, table)
        ref = [t.clone() for t in ops._sfc_fwd(x, M, w, weight, bias, weight2, bias2, spec) if t is not None]
        try:
            L.eqf_sfc_debug_exp(64)
            first = None
            for rep in range(60):
                cur = [t for t in ops._sfc_fwd(x, M, w, weight, bias, weight2, bias2, spec) if t is not None]
                torch.cuda.synchronize()
                if first is None:
                    first = [t.clone() for t in cur]
                    for a, b in zip(first, ref):
                        assert _rel(a, b) < 1e-5
                    assert any(not torch.equal(a, b) for a, b in zip(first, ref)), "the switch selected no other step"
                else:
                    for k, (a, b) in enumerate(zip(cur, first)):
                        assert torch.equal(a, b), (rep, k, float((a - b).abs().max()))
        finally:
            L.eqf_sfc_debug_exp(0)


def _bench_batch():
    from equiformer_amd.synthetic import qm9_like_batch
    return qm9_like_batch(128, 18, side=6.5, seed=11)


def test_qm9_full_batch_is_deterministic():
    from equiformer_amd import nets
    dev = _dev()
    torch.manual_seed(0)
    model = nets.model_entrypoint("graph_attention_transformer_nonlinear_l2")(irreps_in="5x0e", radius=5.0,
                                                                               num_basis=128).to(dev).eval()
    d = {k: v.to(dev) for k, v in _bench_batch().items()}
    with torch.no_grad():
        ys = [model(f_in=None, pos=d["pos"], batch=d["batch"], node_atom=d["z"]) for _ in range(6)]
    for y in ys[1:]:
        assert torch.equal(y, ys[0]), float((y - ys[0]).abs().max())


def _full_batch_errors(mode):
    """The bench workload itself (reference: nets/graph_attention_transformer.py:864-899 with the model of :921-937,
    L1 loss of engine.py:71) against the oracle in fp64: energies of all 128 molecules and the gradient of every
    parameter tensor, in matrix mode `mode`.  The oracle side (minutes of fp64 CPU time) is a committed fixture written by
    tests/golden/make_fullsize_golden.py (case qm9_l2_bench: same seeds, same generators); the weights are the oracle's own
    initialisation under torch.manual_seed(0), rebuilt here (construction only) and loaded into the HIP model."""
    from equiformer_amd import nets, ops
    import fullsize
    dev = _dev()
    meta, outs, gref = fullsize.load("qm9_l2_bench")
    torch.manual_seed(0)
    ref = onets.graph_attention_transformer_nonlinear_l2("5x0e", 5.0)  # (weights only: the oracle is not evaluated here)
    mod = nets.model_entrypoint("graph_attention_transformer_nonlinear_l2")(irreps_in="5x0e", radius=5.0)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(dev).eval()
    d = _bench_batch()
    assert int(meta["molecules"]) == 128 and int(meta["seed"]) == 11
    yr = outs["energy"]
    with ops.matrix_mode(mode):
        y = mod(None, d["pos"].to(dev), d["batch"].to(dev), d["z"].to(dev))
        gg = torch.autograd.grad((y.squeeze() - d["y"].to(dev)).abs().mean(), list(mod.parameters()), allow_unused=True)
    err = _rel(y.cpu(), yr)
    named = {n: g for (n, _), g in zip(mod.named_parameters(), gg)}
    assert [n for n, _ in mod.named_parameters()] == [n for n, _ in ref.named_parameters()]
    worst = fullsize.compare_summary(named, gref, float("inf"))
    print("128 molecules, matrix mode %s: energy rel err vs fp64 oracle %.3e; worst parameter-gradient rel errs: %s"
          % (mode, err, ["%s %.2e" % (n, e) for e, n in worst[:5]]))
    return err, worst


def test_qm9_full_batch_matches_fp64_oracle():
    """default arithmetic (split: every matrix step -- fused SeparableFCTP, per-degree linears, radial MLPs -- multiplies fp32
    operands as bf16 planes on the bf16 matrix cores): the north-star bar, 1e-4 relative"""
    err, worst = _full_batch_errors("split")
    assert err < 1e-4
    assert worst[0][0] < 1e-4, worst[:5]


def test_qm9_full_batch_bf16_mode_stated_tolerance():
    """BASELINE config #2 at the bench batch: EVERY matrix step (fused SeparableFCTP, per-degree linears, radial MLPs) takes plain
    bf16 operands with fp32 accumulation; storage, layer norm, softmax, radial basis, gate stay fp32 -- the arithmetic of the
    reference's AMP default (autocast of every linear, main_qm9.py:117-119,197-201; layer norm pinned to fp32,
    nets/layer_norm.py:89).  Stated tolerance: energies 2e-2, parameter gradients 1e-1 of the gradient's scale (CPU emulation of
    one-plane arithmetic, tools/split_model_error.py: 4.4e-3 / 2.2e-2; the measured values are printed)."""
    err, worst = _full_batch_errors("bf16")
    assert err < 2e-2
    assert worst[0][0] < 1e-1, worst[:5]

