#!/usr/bin/env python
"""Model-level error of split-precision (bf16 planes) matrix steps, emulated on CPU inside the oracle.

Every dense contraction of the QM9 model (the per-degree linears = 'uvw' tensor products with shared weights, and the
radial MLP's nn.Linear layers) is replaced by an autograd function whose forward, data gradient and weight gradient
are GEMMs over operands split into bf16 planes with fp32 accumulation (products of bf16 values are exact in fp32, so
`plane_a.float() @ plane_b.float()` IS what v_mfma_f32_32x32x16_bf16 computes up to summation order).  Reported: energy
error and the worst per-parameter gradient error against the fp64 oracle, next to the plain fp32 oracle -- the numbers
the 1e-4 parity bar (BASELINE.json north_star; tests/test_gpu_fullsize.py) is about.

    python tools/split_model_error.py [molecules]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import e3, nets as onets  # noqa: E402  (checker; this tool is not product code)
from equiformer_amd.synthetic import qm9_like_batch  # noqa: E402

CFG = {"act": 3, "wgt": 3, "grad": 3, "on": False}


def planes(x, n):
    out, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def mm_split(a, b, na, nb):
    """sum over (i, j), i < na, j < nb, i + j <= max(na, nb) - 1 of plane_i(a) @ plane_j(b), small terms first"""
    if a.dtype != torch.float32:
        return a @ b
    pa, pb = planes(a, na), planes(b, nb)
    top = max(na, nb) - 1
    terms = sorted(((i, j) for i in range(na) for j in range(nb) if i + j <= top), key=lambda t: -(t[0] + t[1]))
    acc = None
    for i, j in terms:
        t = pa[i] @ pb[j]
        acc = t if acc is None else acc + t
    return acc


class SplitMM(torch.autograd.Function):
    """y = x @ W with x an activation [rows, K], W a weight [K, N]"""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return mm_split(x, W, CFG["act"], CFG["wgt"])

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx = mm_split(dy, W.t().contiguous(), CFG["grad"], CFG["wgt"])
        dW = mm_split(x.t().contiguous(), dy, CFG["act"], CFG["grad"])
        return dx, dW


_tp_forward = e3.TensorProduct.forward


def tp_forward(self, x1, x2, weight=None):
    """'uvw' products with shared weights and a scalar second input (every LinearRS / FCTP of the model) through SplitMM"""
    if not CFG["on"] or x1.dtype != torch.float32:
        return _tp_forward(self, x1, x2, weight)
    w_all = self.weight if weight is None else weight
    if w_all.dim() != 1 or any(i[3] != "uvw" for i in self.instructions) or self.irreps_in2.dim != 1:
        return _tp_forward(self, x1, x2, weight)
    z = x1.shape[0]
    s1 = self.irreps_in1.slices()
    outs = [None] * len(self.irreps_out)
    off = 0
    for (i1, i2, io, mode, has_w, pw), shape in zip(self.instructions, self.weight_shapes):
        (m1, ir1), (mo, iro) = self.irreps_in1[i1], self.irreps_out[io]
        assert ir1.l == iro.l
        n = m1 * mo
        W = w_all.narrow(0, off, n).reshape(m1, mo)
        off += n
        a = x1[:, s1[i1]].reshape(z, m1, ir1.dim) * x2[:, :1, None]  # scalar second input (ones for LinearRS)
        rows = a.permute(0, 2, 1).reshape(z * ir1.dim, m1)           # rows = (z, m)
        r = SplitMM.apply(rows, W * pw).reshape(z, ir1.dim, mo).permute(0, 2, 1).reshape(z, mo * iro.dim)
        outs[io] = r if outs[io] is None else outs[io] + r
    for k, (mo, iro) in enumerate(self.irreps_out):
        if outs[k] is None:
            outs[k] = x1.new_zeros(z, mo * iro.dim)
    return torch.cat(outs, dim=1)


_linear = torch.nn.functional.linear


def linear(x, W, b=None):
    if not CFG["on"] or x.dtype != torch.float32 or x.dim() != 2:
        return _linear(x, W, b)
    y = SplitMM.apply(x, W.t())
    return y if b is None else y + b


def run(model, d, dtype):
    m = model.to(dtype)
    for p in m.parameters():
        p.grad = None
    y = m(None, d["pos"].to(dtype), d["batch"], d["z"])
    loss = (y.squeeze() - d["y"].to(dtype)).abs().mean()
    loss.backward()
    return y.detach().double(), {n: p.grad.detach().double() for n, p in m.named_parameters() if p.grad is not None}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = onets.graph_attention_transformer_nonlinear_l2("5x0e", 5.0).eval()
    d = qm9_like_batch(B, 18, side=6.5, seed=0)
    e3.TensorProduct.forward = tp_forward
    torch.nn.functional.linear = linear
    e64, g64 = run(model, d, torch.float64)

    def report(tag):
        e, g = run(model, d, torch.float32)
        ee = ((e - e64).abs().max() / e64.abs().max()).item()
        worst = max(((g[n] - g64[n]).abs().max() / g64[n].abs().max().clamp_min(1e-30)).item() for n in g64)
        print("%-34s energy rel err %.2e   worst parameter-gradient rel err %.2e" % (tag, ee, worst), flush=True)

    CFG["on"] = False
    report("fp32 (exact products)")
    CFG["on"] = True
    for act, wgt, grad in ((3, 3, 3), (2, 3, 2), (2, 2, 2), (3, 3, 2), (1, 1, 1)):
        CFG.update(act=act, wgt=wgt, grad=grad)
        report("planes act %d / weight %d / grad %d" % (act, wgt, grad))


if __name__ == "__main__":
    main()
