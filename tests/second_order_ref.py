"""Reference restatements for the second-derivative HIP kernels (TEST INFRASTRUCTURE, not product code).

MD17 force-loss training (`torch.autograd.grad(energy, pos, create_graph=True)` followed by `loss.backward()`,
nets/graph_attention_transformer_md17.py:318-325 and main_md17.py:384-390 of the reference) differentiates the backward
pass.  The product does that with hand-written HIP kernels (equiformer_amd/csrc/second.hip, `eqf_*_bwd2`).  The
functions below restate the forward of the eight non-linear operators with plain torch ops in the channel-fastest
layout, so that autograd can differentiate them twice: tests/test_second_order_restatements.py pins them against the
oracle modules on the CPU (values, gradients, second derivatives), and tests/test_gpu_second_order.py checks every
`eqf_*_bwd2` kernel against their double backward in fp64.  In round 1 this file lived in the product package and WAS
the create_graph path; nothing under equiformer_amd/ imports it any more.

Formulas restate the same reference code as the HIP kernels they shadow (cited per function).
"""
import math

import torch


def vjp(fn, inputs, grad_outputs):
    """Differentiable vector-Jacobian product of `fn` at `inputs` (tensors attached to the outer graph)."""
    with torch.enable_grad():
        outs = fn(*inputs)
        if not isinstance(outs, (tuple, list)):
            outs, grad_outputs = (outs,), (grad_outputs,)
        pairs = [(o, g) for o, g in zip(outs, grad_outputs) if g is not None]
        need = [t for t in inputs if torch.is_tensor(t) and t.requires_grad]
        grads = torch.autograd.grad([o for o, _ in pairs], need, [g for _, g in pairs], create_graph=True,
                                    allow_unused=True)
    it = iter(grads)
    return [next(it) if (torch.is_tensor(t) and t.requires_grad) else None for t in inputs]


# ------------------------------------------------------------------------------------------------- row-local ops
# The second-order step is bound by the number of launches (every tensor op here turns into three or more kernels
# once it has been differentiated twice), so the per-segment loops of the reference are folded into a few products
# with small constant 0/1 matrices that are built once per (layout, device, dtype).
_consts = {}


def _ln_consts(layout, device, dtype):
    key = ("ln", layout.irreps.__repr__(), str(device), dtype)
    c = _consts.get(key)
    if c is None:
        D, nseg = layout.dim, len(layout.segs)
        P = torch.eye(D, dtype=dtype)                       # subtracts the channel mean on the l = 0 segments
        A = torch.zeros(D, nseg, dtype=dtype)               # column -> segment, weighted 1 / (mul (2l+1))
        nw = sum(m for m, _ in layout.segs)
        Cw = torch.zeros(nw, D, dtype=dtype)                # affine_weight index -> columns
        nb = sum(m for m, l in layout.segs if l == 0)
        Cb = torch.zeros(max(nb, 1), D, dtype=dtype)        # affine_bias index -> columns
        iw = ib = 0
        for s, ((mul, l), off) in enumerate(zip(layout.segs, layout.offsets)):
            d = 2 * l + 1
            if l == 0:
                P[off:off + mul, off:off + mul] -= 1.0 / mul
                Cb[torch.arange(ib, ib + mul), torch.arange(off, off + mul)] = 1.0
                ib += mul
            A[off:off + mul * d, s] = 1.0 / (mul * d)
            for m in range(d):
                Cw[torch.arange(iw, iw + mul), torch.arange(off + m * mul, off + (m + 1) * mul)] = 1.0
            iw += mul
        c = tuple(t.to(device) for t in (P, A, (A > 0).to(dtype).t().contiguous(), Cw, Cb))
        _consts[key] = c
    return c


def layer_norm(x, weight, bias, layout, eps):
    """EquivariantLayerNormV2, 'component' normalisation [ref: nets/layer_norm.py:89-152]; rows in CF layout."""
    P, A, B, Cw, Cb = _ln_consts(layout, x.device, x.dtype)
    xc = x @ P                                               # l = 0: minus the mean over channels
    rs = ((xc * xc) @ A + eps).pow(-0.5)                     # [n, segments]: 1 / sqrt(mean over (m, channel) + eps)
    y = xc * (rs @ B) * (weight @ Cw)
    if bias.numel():
        y = y + bias @ Cb
    return y


def _gate_consts(S, gated_layout, device, dtype):
    key = ("gate", S, gated_layout.irreps.__repr__(), str(device), dtype)
    c = _consts.get(key)
    if c is None:
        G = sum(m for m, _ in gated_layout.segs)
        Bg = torch.zeros(G, gated_layout.dim, dtype=dtype)   # gate index -> the columns it multiplies
        ig = 0
        for (mul, l), off in zip(gated_layout.segs, gated_layout.offsets):
            for m in range(2 * l + 1):
                Bg[torch.arange(ig, ig + mul), torch.arange(off + m * mul, off + (m + 1) * mul)] = 1.0
            ig += mul
        c = (G, Bg.to(device))
        _consts[key] = c
    return c


def gate(x, S, gated_layout, c_silu, c_sig):
    """[scalars | gates | gated] -> [c_silu silu(scalars) | gated * c_sig sigmoid(gates)]
    [ref: nets/fast_activation.py:132-148]."""
    G, Bg = _gate_consts(S, gated_layout, x.device, x.dtype)
    gates = torch.sigmoid(x[:, S:S + G]) @ Bg
    return torch.cat([c_silu * torch.nn.functional.silu(x[:, :S]), x[:, S + G:] * gates * c_sig], dim=1)


def scaled_silu(x, c):
    return c * torch.nn.functional.silu(x)


def ln_silu(x, gamma, beta, eps):
    """nn.LayerNorm + nn.SiLU of the radial MLP [ref: nets/radial_func.py:13-36]."""
    return torch.nn.functional.silu(torch.nn.functional.layer_norm(x, (x.shape[1],), gamma, beta, eps))


def alpha_logits(a, alpha_dot, H, Kh, c):
    """sum_k c SmoothLeakyReLU_0.2(a[e,h,k]) alpha_dot[h,k] [ref: nets/graph_attention_transformer.py:54-63,506-507]."""
    a = a.view(-1, H, Kh)
    act = 0.6 * a + 0.4 * a * (2.0 * torch.sigmoid(a) - 1.0)
    return (c * act * alpha_dot.view(1, H, Kh)).sum(-1)


def head_of_column(layout, H, device):
    key = ("hoc", layout.irreps.__repr__(), H, str(device))
    t = _consts.get(key)
    if t is None:
        idx = []
        for (mul, l) in layout.segs:
            mh = mul // H
            for _m in range(2 * l + 1):
                idx.extend(u // mh for u in range(mul))
        t = _consts[key] = torch.tensor(idx, dtype=torch.long, device=device)
    return t


def attn_aggregate(logit, value, graph, H, layout):
    """Per-destination softmax (exp(x - max) / (sum + 1e-16)) and weighted aggregation, no dropout
    [ref: torch_geometric.utils.softmax + scatter, nets/graph_attention_transformer.py:508-514]."""
    dst = graph.dst.long()
    N, E = graph.N, logit.shape[0]
    idx = dst[:, None].expand(E, H)
    mx = torch.full((N, H), float("-inf"), device=logit.device, dtype=logit.dtype).scatter_reduce(
        0, idx, logit.detach(), reduce="amax", include_self=True)
    ex = torch.exp(logit - mx[dst])
    den = torch.zeros((N, H), device=logit.device, dtype=logit.dtype).index_add(0, dst, ex)
    alpha = ex / (den[dst] + 1e-16)
    hoc = head_of_column(layout, H, logit.device)
    return torch.zeros((N, layout.dim), device=value.device, dtype=value.dtype).index_add(0, dst, value * alpha[:, hoc])


# ------------------------------------------------------------------------------------------------- edge geometry
def rbf_expnorm(length, means, betas, alpha, cutoff):
    """[ref: nets/graph_attention_transformer_md17.py:51-81,119-124]"""
    d = length.unsqueeze(-1)
    cut = 0.5 * (torch.cos(d * math.pi / cutoff) + 1.0) * (d < cutoff).to(d.dtype)
    return cut * torch.exp(-betas * (torch.exp(-alpha * d) - means) ** 2)


def spherical_harmonics(lmax, vec):
    """Component-normalised real spherical harmonics of the unit vector, l <= 3, (x, y, z) order with y the polar axis
    [ref: e3nn 0.4.4 o3.spherical_harmonics(normalize=True, normalization='component'), call site
    nets/graph_attention_transformer.py:869-870]."""
    u = torch.nn.functional.normalize(vec, dim=-1)
    x, y, z = u[..., 0], u[..., 1], u[..., 2]
    out = [torch.ones_like(x)]
    if lmax >= 1:
        out += [x, y, z]
    if lmax >= 2:
        s3 = math.sqrt(3.0)
        y2 = y * y
        x2z2 = x * x + z * z
        sh20, sh21, sh22, sh23 = s3 * x * z, s3 * x * y, y2 - 0.5 * x2z2, s3 * y * z
        sh24 = (s3 / 2.0) * (z * z - x * x)
        out += [sh20, sh21, sh22, sh23, sh24]
    if lmax >= 3:
        out += [math.sqrt(5.0 / 6.0) * (sh20 * z + sh24 * x), math.sqrt(5.0) * sh20 * y,
                math.sqrt(3.0 / 8.0) * (4.0 * y2 - x2z2) * x, 0.5 * y * (2.0 * y2 - 3.0 * x2z2),
                math.sqrt(3.0 / 8.0) * z * (4.0 * y2 - x2z2), math.sqrt(5.0) * sh24 * y,
                math.sqrt(5.0 / 6.0) * (sh24 * z - sh20 * x)]
    sh = torch.stack(out, dim=-1)
    scale = torch.cat([torch.full((2 * l + 1,), math.sqrt(2 * l + 1), dtype=sh.dtype, device=sh.device)
                       for l in range(lmax + 1)])
    return sh * scale


def edge_geometry(pos, offsets, graph, lmax):
    """(edge_length, edge_sh) as functions of pos [ref: nets/graph_attention_transformer.py:868-874]."""
    vec = pos[graph.src.long()] - pos[graph.dst.long()]
    if offsets is not None:
        vec = vec + offsets
    return vec.norm(dim=1), spherical_harmonics(lmax, vec)
