"""Per-degree / dense linears on the bf16 matrix cores (csrc/gemmx.hip, C ABI eqf_gemmx_group) against an fp64 reference of the
same contraction [ref: LinearRS / FullyConnectedTensorProductRescale, nets/tensor_product_rescale.py:125-136,171-174;
torch.nn.Linear inside RadialProfile, nets/radial_func.py:46-49].  All four descriptor kinds, two-level rows (degree segments
of irreps rows), partial tiles, K tails, bias, accumulate, the bias gradients taken from the staged operands.
Tolerances (relative to the result scale), stated per mode:  split 2e-5 (measured ~2e-6), split6 3e-6, bf16 2e-2."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import ops  # noqa: E402
from equiformer_amd.layout import RowLayout  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = {"split": 2e-5, "split6": 3e-6, "bf16": 2e-2, "fp32": 2e-6}


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _seg(x, lay, i):
    """degree segment i of CF rows [n, D] as a [n, 2l+1, mul] view (fp64, CPU)"""
    mul, l = lay.segs[i]
    o = lay.offsets[i]
    return x.double().cpu()[:, o:o + mul * (2 * l + 1)].reshape(x.shape[0], 2 * l + 1, mul)


CASES = [("128x0e+64x1e+32x2e", "128x0e+64x1e+32x2e", 2304), ("128x0e+64x1e+32x2e", "384x0e+192x1e+96x2e", 517),
         ("384x0e+192x1e+96x2e", "128x0e+64x1e+32x2e", 1000), ("16x0e+8x1e+8x2e", "40x0e+24x1e", 333),
         ("224x0e+384x1e+352x2e", "224x0e+64x1e+32x2e", 700)]


@pytest.mark.parametrize("mode", ["split", "bf16", "split6"])
@pytest.mark.parametrize("case", CASES)
def test_irreps_linear_all_kinds(case, mode):
    irr_in, irr_out, n = case
    dev = _dev()
    li, lo = RowLayout(irr_in), RowLayout(irr_out)
    spec = ops.LinearSpec(li, lo)
    g = torch.Generator().manual_seed(len(irr_in) + n)
    x = torch.randn(n, li.dim, generator=g).to(dev)
    dy = torch.randn(n, lo.dim, generator=g).to(dev)
    w = (torch.randn(spec.weight_numel, generator=g) * 0.1).to(dev)
    bias = torch.randn(spec.bias_dim, generator=g).to(dev) if spec.bias_dim else None
    with ops.matrix_mode(mode):
        y = ops._lin_fwd(x, w, bias, spec)
        dx = ops._lin_dgrad(dy, w, spec)
        dw = torch.zeros_like(w)
        db = torch.zeros(spec.bias_dim, device=dev) if spec.bias_dim else None
        ops._lin_wgrad(x, dy, spec, dw, db)
    torch.cuda.synchronize()
    wr = w.double().cpu()
    worst = {}
    y_ref = torch.zeros(n, lo.dim, dtype=torch.float64)
    dx_ref = torch.zeros(n, li.dim, dtype=torch.float64)
    dw_ref = torch.zeros_like(wr)
    for (l, in_off, K, out_off, N, w_off) in spec.pairs:
        d = 2 * l + 1
        W = wr[w_off:w_off + K * N].view(K, N)
        xs = x.double().cpu()[:, in_off:in_off + d * K].reshape(n, d, K)
        ds = dy.double().cpu()[:, out_off:out_off + d * N].reshape(n, d, N)
        ys = xs @ W
        if spec.has_bias(l, out_off) and bias is not None:
            ys = ys + bias.double().cpu()
        y_ref[:, out_off:out_off + d * N] = ys.reshape(n, -1)
        dx_ref[:, in_off:in_off + d * K] = (ds @ W.T).reshape(n, -1)
        dw_ref[w_off:w_off + K * N] = torch.einsum("ndk,ndm->km", xs, ds).reshape(-1)
    worst["y"], worst["dx"], worst["dw"] = _rel(y, y_ref), _rel(dx, dx_ref), _rel(dw, dw_ref)
    if db is not None:
        j = lo.seg_index(0)
        worst["db"] = _rel(db, dy.double().cpu()[:, lo.offsets[j]:lo.offsets[j] + spec.bias_dim].sum(0))
        assert worst["db"] < 2e-6  # bias gradients come from the fp32 values, whatever the mode
    print("%s -> %s n=%d mode %s: %s" % (irr_in, irr_out, n, mode, {k: "%.1e" % v for k, v in worst.items()}))
    assert all(v < TOL[mode] for v in worst.values()), worst


@pytest.mark.parametrize("mode", ["split", "bf16"])
@pytest.mark.parametrize("shape", [(25354, 128, 64), (1000, 64, 960), (37, 32, 64), (513, 64, 64),
                                   (8200, 64, 960), (9001, 32, 200)])  # the last two: the short-K walk-all-columns kernel
def test_dense_linear_all_kinds(shape, mode):
    """nn.Linear orientation (weight [N, K]) of the radial MLPs: forward, data gradient, weight + bias gradient"""
    M, K, N = shape
    dev = _dev()
    g = torch.Generator().manual_seed(M)
    x, dy = torch.randn(M, K, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    W, b = (torch.randn(N, K, generator=g) * 0.1).to(dev), torch.randn(N, generator=g).to(dev)
    with ops.matrix_mode(mode):
        y = ops._dense_fwd(x, W, b)
        dx = ops._dense_dgrad(dy, W)
        dw, db = torch.zeros_like(W), torch.zeros_like(b)
        ops._dense_wgrad(x, dy, dw, db)
    torch.cuda.synchronize()
    xd, dyd, Wd = x.double().cpu(), dy.double().cpu(), W.double().cpu()
    errs = {"y": _rel(y, xd @ Wd.T + b.double().cpu()), "dx": _rel(dx, dyd @ Wd), "dw": _rel(dw, dyd.T @ xd),
            "db": _rel(db, dyd.sum(0))}
    print("dense %s mode %s: %s" % (shape, mode, {k: "%.1e" % v for k, v in errs.items()}))
    assert errs["db"] < 2e-6
    assert all(v < TOL[mode] for v in errs.values()), errs


def test_accumulate_and_grouped_linear_match_fp32_kernels():
    """the radial bank's grouped nn.Linear (forward, data gradient, weight gradients) in split mode against the exact-fp32 MFMA
    kernels (mode fp32) on the same inputs"""
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    G, K, rows_n = 3, 64, 2000
    Ns = [64, 960, 896]
    x = torch.randn(rows_n, G * K, generator=g).to(dev).requires_grad_(True)
    Ws = [(torch.randn(n, K, generator=g) * 0.1).to(dev).requires_grad_(True) for n in Ns]
    bs = [torch.randn(n, generator=g).to(dev).requires_grad_(True) for n in Ns]
    c = torch.randn(rows_n, sum(Ns), generator=g).to(dev)
    res = {}
    for mode in ("fp32", "split"):
        with ops.matrix_mode(mode):
            y = ops.grouped_linear(x, K, Ws, bs, True)
            grads = torch.autograd.grad((y * c).sum(), [x] + Ws + bs)
        res[mode] = [y.detach()] + [t.detach() for t in grads]
    worst = max(_rel(a, b) for a, b in zip(res["split"], res["fp32"]))
    print("grouped linear split vs fp32 kernels: worst %.1e" % worst)
    assert worst < 2e-5


def test_one_wave_per_tile_kernels_match_the_tiled_ones():
    """development variant (eqf_gemmx_dev_set(0, 0)): one wave per 32 x 32 tile, whole K chunks in flight -- same results"""
    from equiformer_amd import lib as _lib
    dev = _dev()
    li, lo = RowLayout("128x0e+64x1e+32x2e"), RowLayout("384x0e+192x1e+96x2e")
    spec = ops.LinearSpec(li, lo)
    g = torch.Generator().manual_seed(11)
    n = 777
    x, dy = torch.randn(n, li.dim, generator=g).to(dev), torch.randn(n, lo.dim, generator=g).to(dev)
    w = (torch.randn(spec.weight_numel, generator=g) * 0.1).to(dev)
    bias = torch.randn(spec.bias_dim, generator=g).to(dev)
    res = {}
    for direct in (1, 0):
        _lib.load().eqf_gemmx_dev_set(0, direct)
        try:
            dw, db = torch.zeros_like(w), torch.zeros_like(bias)
            y, dx = ops._lin_fwd(x, w, bias, spec), ops._lin_dgrad(dy, w, spec)
            ops._lin_wgrad(x, dy, spec, dw, db)
            torch.cuda.synchronize()
            res[direct] = (y, dx, dw, db)
        finally:
            _lib.load().eqf_gemmx_dev_set(0, 1)
    for a, b in zip(res[0], res[1]):
        assert _rel(a, b) < 2e-6


def test_deferred_weight_gradients_equal_immediate_ones():
    """The weight gradients of node-row linears that belong to leaf parameters are queued during a first-order backward and
    launched together when the pass ends (ops._defer_lin_wgrad): same values through .backward() (AccumulateGrad adopts the
    zero tensor the launch later fills), through torch.autograd.grad (captured tensors), and with a pre-existing .grad
    (not deferred)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    from weights import fill_deterministic
    from equiformer_amd.nets.graph_attention_transformer import GraphAttentionTransformer
    from equiformer_amd.synthetic import qm9_like_batch
    dev = _dev()
    m = GraphAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **dict(mg.SMALL_L2, alpha_drop=0.0))
    m = fill_deterministic(m, 21).to(dev).train()
    d = {k: v.to(dev) for k, v in qm9_like_batch(4, 10, side=5.0, seed=5).items()}
    params = [p for p in m.parameters() if p.requires_grad]

    def loss():
        return (m(None, d["pos"], d["batch"], d["z"]).squeeze() - d["y"]).abs().mean()

    res = {}
    for on in (False, True):
        prev = ops.set_deferred_weight_gradients(on)
        try:
            for p in params:
                p.grad = None
            ops.deferred_weight_gradient_stats(reset=True)
            loss().backward()
            stats = ops.deferred_weight_gradient_stats()
            assert (stats["queued"] > 0 and 0 < stats["flushes"] <= 8) if on else stats["queued"] == 0, stats
            res[on, "backward"] = [None if p.grad is None else p.grad.clone() for p in params]
            res[on, "grad"] = list(torch.autograd.grad(loss(), params, allow_unused=True))
            loss().backward()  # second pass onto existing .grad: 2 x the gradient
            res[on, "twice"] = [None if p.grad is None else p.grad.clone() for p in params]
        finally:
            ops.set_deferred_weight_gradients(prev)
    assert not any(q.entries for q in ops._task_queues.values())
    for key in ("backward", "grad", "twice"):
        for a, b in zip(res[True, key], res[False, key]):
            assert (a is None) == (b is None)
            if a is not None:
                assert _rel(a, b) < 2e-6, key
    for a, b in zip(res[True, "twice"], res[True, "backward"]):
        if a is not None:
            assert _rel(a, 2.0 * b) < 2e-6
