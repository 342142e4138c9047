"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the Equiformer hot path.

PINNED against the reference's own model code: tests/test_reference_pin.py imports /root/reference/nets unchanged
(oracle/refshim) and finds every family of this file equal to it in fp64 (<= 1e-9) with weights copied by name.
Op-for-op restatement, in plain torch (fp32 or fp64, CPU), of the model classes of the reference
(paths relative to /root/reference):

  nets/graph_attention_transformer.py      SmoothLeakyReLU :54-63, DepthwiseTensorProduct :157-183, SeparableFCTP :186-248,
                                           Vec2AttnHeads :251-280, AttnHeads2Vec :288-312, GraphAttention :403-527,
                                           FeedForwardNetwork :537-571, TransBlock :575-667, NodeEmbeddingNetwork :670-690,
                                           ScaledScatter :693-702, EdgeDegreeEmbeddingNetwork :709-733,
                                           GraphAttentionTransformer :736-899, factories :902-937
  nets/tensor_product_rescale.py           TensorProductRescale :15-141, FullyConnectedTensorProductRescale :144-162,
                                           LinearRS :165-174, irreps2gate :177-192
  nets/layer_norm.py                       EquivariantLayerNormV2 :62-152
  nets/fast_activation.py                  Activation :15-87, Gate :91-160
  nets/radial_func.py                      RadialProfile :9-49
  nets/gaussian_rbf.py                     gaussian :5-9, GaussianRadialBasisLayer :13-40
  nets/graph_attention_transformer_md17.py CosineCutoff :51-81, ExpNormalSmearing :85-124, model :127-327, factories :407-442
  nets/graph_attention_transformer_oc20.py GraphAttentionTransformerOC20 :85-381 (precomputed-edge form): energy head on the
                                           scalar channels :169-179, auxiliary IS2RS head :182-194, attention head + skip
                                           :196-208, forward :305-381
  nets/drop.py                             drop_path :13-29, GraphDropPath :45-61 (inside TransBlock)
  nets/dp_attention_transformer*.py        ScaleFactor :45-66, DotProductAttention :68-160, DPTransBlock :163-252 and the
                                           three model classes (QM9, MD17, OC20)
  nets/equiformer_md17_dens.py             Equiformer_MD17_DeNS :55-354 (force encoding, energy head, denoising head)
  ocpmodels gemnet layers (un-vendored)    SphericalBesselBasis / polynomial Envelope / RadialBasis as called by :785-787

Third-party ops restated: torch_cluster.radius_graph (1.6.0), torch_scatter.scatter (2.0.9),
torch_geometric.utils.softmax / nn.inits.glorot (2.0.3).
Module / parameter names equal the reference's so that state_dicts are interchangeable with the product
package (`equiformer_amd.nets`).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import math

import torch
from torch import nn

from .e3 import Irrep, Irreps, TensorProduct, normalize2mom_const, sort_irreps_even_first, spherical_harmonics

_RESCALE = True
_USE_BIAS = True


# ---------------------------------------------------------------------------- third-party graph ops
def radius_graph(pos, r, batch, max_num_neighbors=1000):
    """torch_cluster.radius_graph(flow='source_to_target', loop=False): returns (edge_src=neighbour,
    edge_dst=centre), grouped by ascending dst then ascending src, strict d < r, same molecule only."""
    n = pos.shape[0]
    d2 = (pos[:, None, :] - pos[None, :, :]).pow(2).sum(-1)
    same = batch[:, None] == batch[None, :]
    adj = same & (d2 < r * r) & ~torch.eye(n, dtype=torch.bool, device=pos.device)
    dst, src = adj.nonzero(as_tuple=True)  # row-major nonzero: dst ascending, src ascending
    if max_num_neighbors is not None:
        # keep the first max_num_neighbors sources per destination (torch_cluster visits sources in index order)
        rank = torch.cumsum(adj.long(), dim=1)[dst, src]
        keep = rank <= max_num_neighbors
        dst, src = dst[keep], src[keep]
    return src, dst


def scatter_sum(x, index, dim_size):
    out = x.new_zeros((dim_size,) + x.shape[1:])
    return out.index_add(0, index, x)


def segment_softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax (2.0.3): exp(x - max_seg) / (sum_seg + 1e-16)."""
    smax = src.new_full((num_nodes,) + src.shape[1:], float("-inf"))
    smax = smax.scatter_reduce(0, index.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src, reduce="amax")
    out = (src - smax[index]).exp()
    ssum = scatter_sum(out, index, num_nodes)
    return out / (ssum[index] + 1e-16)


# ---------------------------------------------------------------------------- tensor_product_rescale.py
class TensorProductRescale(nn.Module):
    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, bias=True, rescale=True,
                 internal_weights=None, shared_weights=None):
        super().__init__()
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        self.rescale, self.use_bias = rescale, bias
        self.tp = TensorProduct(self.irreps_in1, self.irreps_in2, self.irreps_out, instructions,
                                internal_weights=internal_weights, shared_weights=shared_weights)
        # bias on every 0e slice of the *simplified* output irreps (:55-83)
        self.irreps_bias = self.irreps_out.simplify()
        self.bias_slices = []
        biases = []
        if bias:
            for (mul, ir), sl in zip(self.irreps_bias, self.irreps_bias.slices()):
                if ir.l == 0 and ir.p == 1:
                    biases.append(nn.Parameter(torch.zeros(mul)))
                    self.bias_slices.append(sl)
        self.bias = nn.ParameterList(biases)
        # fan-in per output slice (:85-110)
        fan = {}
        for i1, i2, io, mode, _, _ in self.tp.instructions:
            m1, m2 = self.irreps_in1[i1][0], self.irreps_in2[i2][0]
            f = {"uvw": m1 * m2, "uvu": m2, "uuu": 1}[mode]
            fan[io] = fan.get(io, 0) + f
        out_slices = self.irreps_out.slices()
        self.slices_sqrt_k = {io: (out_slices[io], 1 / fan[io] ** 0.5 if rescale else 1.0) for io in fan}
        if self.tp.internal_weights and rescale:
            with torch.no_grad():
                for w, ins in zip(self.tp.weight_views(), [i for i in self.tp.instructions if i[4]]):
                    w.mul_(self.slices_sqrt_k[ins[2]][1])  # views alias tp.weight storage

    def forward(self, x, y, weight=None):
        out = self.tp(x, y, weight)
        for sl, b in zip(self.bias_slices, self.bias):
            out = torch.cat([out[:, :sl.start], out[:, sl] + b, out[:, sl.stop:]], dim=1)
        return out


class FullyConnectedTensorProductRescale(TensorProductRescale):
    def __init__(self, irreps_in1, irreps_in2, irreps_out, bias=True, rescale=True, internal_weights=None,
                 shared_weights=None):
        irreps_in1, irreps_in2, irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        ins = [(i1, i2, io, "uvw", True, 1.0)
               for i1, (_, a) in enumerate(irreps_in1) for i2, (_, b) in enumerate(irreps_in2)
               for io, (_, c) in enumerate(irreps_out) if c in a * b]
        super().__init__(irreps_in1, irreps_in2, irreps_out, ins, bias=bias, rescale=rescale,
                         internal_weights=internal_weights, shared_weights=shared_weights)


class LinearRS(FullyConnectedTensorProductRescale):
    def __init__(self, irreps_in, irreps_out, bias=True, rescale=True):
        super().__init__(irreps_in, Irreps("1x0e"), irreps_out, bias=bias, rescale=rescale,
                         internal_weights=True, shared_weights=True)

    def forward(self, x):
        return super().forward(x, torch.ones_like(x[:, 0:1]))


def irreps2gate(irreps):
    scalars = Irreps([(m, ir) for m, ir in Irreps(irreps) if ir.l == 0 and ir.p == 1]).simplify()
    gated = Irreps([(m, ir) for m, ir in Irreps(irreps) if not (ir.l == 0 and ir.p == 1)]).simplify()
    gates = Irreps([(m, "0e") for m, _ in gated]).simplify() if gated.dim > 0 else Irreps()
    return scalars, gates, gated


# ---------------------------------------------------------------------------- fast_activation.py
class ScaledAct(nn.Module):
    """normalize2mom(act): act(x) * c, c from the e3nn Monte-Carlo recipe."""

    def __init__(self, act):
        super().__init__()
        self.act = act
        self.cst = normalize2mom_const(act)

    def forward(self, x):
        return self.act(x) * self.cst


class SmoothLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.alpha = negative_slope

    def forward(self, x):
        return ((1 + self.alpha) / 2) * x + ((1 - self.alpha) / 2) * x * (2 * torch.sigmoid(x) - 1)


class Activation(nn.Module):
    """Single-act form only (all call sites in the hot path): applied to the whole tensor (:70-71)."""

    def __init__(self, irreps_in, acts):
        super().__init__()
        self.irreps_in = self.irreps_out = Irreps(irreps_in)
        assert len(acts) == 1 and len(self.irreps_in) == 1 and self.irreps_in[0][1].l == 0
        self.acts = nn.ModuleList([ScaledAct(acts[0])])

    def forward(self, x):
        return self.acts[0](x)


class Gate(nn.Module):
    def __init__(self, irreps_scalars, irreps_gates, irreps_gated):
        super().__init__()
        self.irreps_scalars, self.irreps_gates, self.irreps_gated = irreps_scalars, irreps_gates, irreps_gated
        self.irreps_in = (irreps_scalars + irreps_gates + irreps_gated).simplify()
        self.irreps_out = irreps_scalars + irreps_gated
        self.act_scalars = ScaledAct(torch.nn.functional.silu)
        self.act_gates = ScaledAct(torch.sigmoid)

    def forward(self, x):
        ns, ng = self.irreps_scalars.dim, self.irreps_gates.dim
        scalars, gates, gated = x[:, :ns], x[:, ns:ns + ng], x[:, ns + ng:]
        scalars, gates = self.act_scalars(scalars), self.act_gates(gates)
        out, ig, ix = [scalars], 0, 0
        for mul, ir in self.irreps_gated:  # o3.ElementwiseTensorProduct(gated, gates): 'uuu', C(l,0,l) -> plain product
            blk = gated[:, ix:ix + mul * ir.dim].reshape(-1, mul, ir.dim) * gates[:, ig:ig + mul, None]
            out.append(blk.reshape(-1, mul * ir.dim))
            ig, ix = ig + mul, ix + mul * ir.dim
        return torch.cat(out, dim=1)


def make_gate(irreps_out):
    scalars, gates, gated = irreps2gate(irreps_out)
    if gated.num_irreps == 0:
        return Activation(irreps_out, [torch.nn.functional.silu])
    return Gate(scalars, gates, gated)


# ---------------------------------------------------------------------------- layer_norm.py / radial_func.py / rbf
class EquivariantLayerNormV2(nn.Module):
    def __init__(self, irreps, eps=1e-5):
        super().__init__()
        self.irreps, self.eps = Irreps(irreps), eps
        self.affine_weight = nn.Parameter(torch.ones(self.irreps.num_irreps))
        self.affine_bias = nn.Parameter(torch.zeros(sum(m for m, ir in self.irreps if ir.l == 0 and ir.p == 1)))

    def forward(self, x, **kwargs):
        out, ix, iw, ib = [], 0, 0, 0
        for mul, ir in self.irreps:
            f = x[:, ix:ix + mul * ir.dim].reshape(-1, mul, ir.dim)
            ix += mul * ir.dim
            if ir.l == 0 and ir.p == 1:
                f = f - f.mean(dim=1, keepdim=True)
            nrm = f.pow(2).mean(-1).mean(dim=1, keepdim=True)
            nrm = (nrm + self.eps).pow(-0.5) * self.affine_weight[None, iw:iw + mul]
            iw += mul
            f = f * nrm.reshape(-1, mul, 1)
            if ir.dim == 1 and ir.p == 1:
                f = f + self.affine_bias[ib:ib + mul].reshape(mul, 1)
                ib += mul
            out.append(f.reshape(-1, mul * ir.dim))
        assert ix == x.shape[-1]
        return torch.cat(out, dim=-1)


class RadialProfile(nn.Module):
    def __init__(self, ch_list):
        super().__init__()
        mods = []
        for i in range(1, len(ch_list)):
            last = i == len(ch_list) - 1
            mods.append(nn.Linear(ch_list[i - 1], ch_list[i], bias=not last))
            if last:
                break
            mods += [nn.LayerNorm(ch_list[i]), nn.SiLU()]
        self.net = nn.Sequential(*mods)
        self.offset = nn.Parameter(torch.zeros(ch_list[-1]))
        bound = 1 / math.sqrt(ch_list[-2])
        nn.init.uniform_(self.offset, -bound, bound)

    def forward(self, x):
        return self.net(x) + self.offset.reshape(1, -1)


class GaussianRadialBasisLayer(nn.Module):
    def __init__(self, num_basis, cutoff):
        super().__init__()
        self.num_basis, self.cutoff = num_basis, cutoff + 0.0
        self.mean = nn.Parameter(torch.zeros(1, num_basis))
        self.std = nn.Parameter(torch.zeros(1, num_basis))
        self.weight = nn.Parameter(torch.ones(1, 1))
        self.bias = nn.Parameter(torch.zeros(1, 1))
        nn.init.uniform_(self.mean, 0, 1.0)
        nn.init.uniform_(self.std, 1.0 / num_basis, 1.0)

    def forward(self, dist, *unused):
        x = self.weight * (dist / self.cutoff).unsqueeze(-1) + self.bias
        std = self.std.abs() + 1e-5
        a = (2 * 3.14159) ** 0.5  # sic: truncated pi (gaussian_rbf.py:6)
        return torch.exp(-0.5 * ((x - self.mean) / std) ** 2) / (a * std)


class ExpNormalSmearing(nn.Module):
    def __init__(self, cutoff_lower=0.0, cutoff_upper=5.0, num_rbf=50):
        super().__init__()
        self.cutoff_lower, self.cutoff_upper = cutoff_lower, cutoff_upper
        self.alpha = 5.0 / (cutoff_upper - cutoff_lower)
        start = torch.exp(torch.scalar_tensor(-cutoff_upper + cutoff_lower))
        self.register_buffer("means", torch.linspace(start, 1, num_rbf))
        self.register_buffer("betas", torch.tensor([(2 / num_rbf * (1 - start)) ** -2] * num_rbf))

    def forward(self, dist):
        d = dist.unsqueeze(-1)
        cut = 0.5 * (torch.cos(d * math.pi / self.cutoff_upper) + 1.0) * (d < self.cutoff_upper).to(d.dtype)
        return cut * torch.exp(-self.betas * (torch.exp(self.alpha * (-d + self.cutoff_lower)) - self.means) ** 2)


class SphericalBesselBasis(nn.Module):
    """[dep] ocpmodels 0.0.3 (commit d2aaaeb, docs/env_setup.md:18-26 of the reference), models/gemnet/layers/
    radial_basis.py -- un-vendored; restated from GemNet's published definition: sqrt(2 / c^3) sin(f_k x) / x with
    x = d / c and trainable frequencies f_k initialised to k pi."""

    def __init__(self, num_radial, cutoff):
        super().__init__()
        self.norm_const = math.sqrt(2.0 / (cutoff ** 3))
        self.frequencies = nn.Parameter(torch.tensor([math.pi * k for k in range(1, num_radial + 1)], dtype=torch.float32))

    def forward(self, d_scaled):
        return self.norm_const / d_scaled[:, None] * torch.sin(self.frequencies * d_scaled[:, None])


class RadialBasis(nn.Module):
    """[dep] ocpmodels RadialBasis(num_radial, cutoff, rbf={'name': 'spherical_bessel'}) with its default polynomial
    envelope of exponent 5: env(x) = 1 + a x^p + b x^(p+1) + c x^(p+2) for x < 1 (a = -(p+1)(p+2)/2, b = p(p+2),
    c = -p(p+1)/2), 0 beyond.  [ref call sites: nets/graph_attention_transformer.py:786-788, ..._md17.py:178-180]"""

    def __init__(self, num_radial, cutoff, rbf=None, envelope=None):
        super().__init__()
        assert (rbf or {"name": "spherical_bessel"})["name"] == "spherical_bessel"
        self.inv_cutoff = 1.0 / cutoff
        p = 5
        self.p, self.a, self.b, self.c = p, -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
        self.rbf = SphericalBesselBasis(num_radial, cutoff)

    def forward(self, d, *unused):
        x = d * self.inv_cutoff
        env = 1 + self.a * x ** self.p + self.b * x ** (self.p + 1) + self.c * x ** (self.p + 2)
        env = torch.where(x < 1, env, torch.zeros_like(x))
        return env[:, None] * self.rbf(x)


# ---------------------------------------------------------------------------- graph_attention_transformer.py
def DepthwiseTensorProduct(irreps_in, irreps_edge, irreps_node_output, internal_weights=False, bias=True):
    irreps_in, irreps_edge, irreps_node_output = Irreps(irreps_in), Irreps(irreps_edge), Irreps(irreps_node_output)
    out, ins = [], []
    for i, (mul, ir_in) in enumerate(irreps_in):
        for j, (_, ir_e) in enumerate(irreps_edge):
            for ir_out in ir_in * ir_e:
                if ir_out in irreps_node_output or ir_out == Irrep(0, 1):
                    ins.append((i, j, len(out), "uvu", True))
                    out.append((mul, ir_out))
    out, p, _ = sort_irreps_even_first(Irreps(out))
    ins = [(a, b, p[c], m, t) for a, b, c, m, t in ins]
    return TensorProductRescale(irreps_in, irreps_edge, out, ins, internal_weights=internal_weights,
                                shared_weights=internal_weights, bias=bias, rescale=_RESCALE)


class SeparableFCTP(nn.Module):
    def __init__(self, irreps_in, irreps_edge, irreps_out, fc_neurons, use_activation=False, internal_weights=False):
        super().__init__()
        irreps_out = Irreps(irreps_out)
        self.dtp = DepthwiseTensorProduct(irreps_in, irreps_edge, irreps_out, bias=False,
                                          internal_weights=internal_weights)
        self.dtp_rad = None
        if fc_neurons is not None:
            self.dtp_rad = RadialProfile(fc_neurons + [self.dtp.tp.weight_numel])
            for sl, k in self.dtp.slices_sqrt_k.values():  # k == 1 for 'uvu' with mul2 == 1
                self.dtp_rad.net[-1].weight.data[sl, :] *= k
                self.dtp_rad.offset.data[sl] *= k
        lin_out = irreps_out
        if use_activation:
            s, g, gd = irreps2gate(irreps_out)
            lin_out = (s + g + gd).simplify()
        self.lin = LinearRS(self.dtp.irreps_out.simplify(), lin_out)
        self.gate = make_gate(irreps_out) if use_activation else None

    def forward(self, x, edge_attr, edge_scalars):
        w = self.dtp_rad(edge_scalars) if (self.dtp_rad is not None and edge_scalars is not None) else None
        out = self.lin(self.dtp(x, edge_attr, w))
        return self.gate(out) if self.gate is not None else out


def vec2heads(x, irreps_head, num_heads):
    out, ix = [], 0
    for mul, ir in irreps_head:
        w = num_heads * mul * ir.dim
        out.append(x[:, ix:ix + w].reshape(x.shape[0], num_heads, -1))
        ix += w
    return torch.cat(out, dim=2)


def heads2vec(x, irreps_head):
    out, ix = [], 0
    for mul, ir in irreps_head:
        w = mul * ir.dim
        out.append(x[:, :, ix:ix + w].reshape(x.shape[0], -1))
        ix += w
    return torch.cat(out, dim=1)


class GraphAttention(nn.Module):
    def __init__(self, irreps_node_input, irreps_node_attr, irreps_edge_attr, irreps_node_output, fc_neurons,
                 irreps_head, num_heads, irreps_pre_attn=None, rescale_degree=False, nonlinear_message=False,
                 alpha_drop=0.1, proj_drop=0.1):
        super().__init__()
        self.irreps_node_input = Irreps(irreps_node_input)
        self.irreps_pre_attn = self.irreps_node_input if irreps_pre_attn is None else Irreps(irreps_pre_attn)
        self.irreps_head, self.num_heads = Irreps(irreps_head), num_heads
        self.rescale_degree, self.nonlinear_message = rescale_degree, nonlinear_message
        self.merge_src = LinearRS(self.irreps_node_input, self.irreps_pre_attn, bias=True)
        self.merge_dst = LinearRS(self.irreps_node_input, self.irreps_pre_attn, bias=False)
        heads_all = sort_irreps_even_first(self.irreps_head * num_heads)[0].simplify()
        mul_alpha = sum(m for m, ir in heads_all if ir.l == 0 and ir.p == 1)
        self.mul_alpha_head = mul_alpha // num_heads
        irreps_alpha = Irreps("{}x0e".format(mul_alpha))
        self.irreps_alpha_head = Irreps("{}x0e".format(self.mul_alpha_head))
        if nonlinear_message:
            self.sep_act = SeparableFCTP(self.irreps_pre_attn, irreps_edge_attr, self.irreps_pre_attn, fc_neurons,
                                         use_activation=True, internal_weights=False)
            self.sep_alpha = LinearRS(self.sep_act.dtp.irreps_out, irreps_alpha)
            self.sep_value = SeparableFCTP(self.irreps_pre_attn, irreps_edge_attr, heads_all, fc_neurons=None,
                                           use_activation=False, internal_weights=True)
        else:
            self.sep = SeparableFCTP(self.irreps_pre_attn, irreps_edge_attr, (irreps_alpha + heads_all).simplify(),
                                     fc_neurons, use_activation=False)
            self.irreps_mix_head = (self.irreps_alpha_head + self.irreps_head).simplify()
        self.alpha_act = Activation(self.irreps_alpha_head, [SmoothLeakyReLU(0.2)])
        self.alpha_dot = nn.Parameter(torch.randn(1, num_heads, self.mul_alpha_head))
        stdv = math.sqrt(6.0 / (self.alpha_dot.size(-2) + self.alpha_dot.size(-1)))  # PyG glorot
        self.alpha_dot.data.uniform_(-stdv, stdv)
        self.alpha_dropout = nn.Dropout(alpha_drop) if alpha_drop != 0.0 else None
        self.proj = LinearRS(heads_all, Irreps(irreps_node_output))
        assert proj_drop == 0.0

    def forward(self, node_input, edge_src, edge_dst, edge_attr, edge_scalars):
        message = self.merge_src(node_input)[edge_src] + self.merge_dst(node_input)[edge_dst]
        if self.nonlinear_message:
            weight = self.sep_act.dtp_rad(edge_scalars)
            message = self.sep_act.dtp(message, edge_attr, weight)
            alpha = vec2heads(self.sep_alpha(message), self.irreps_alpha_head, self.num_heads)
            value = self.sep_act.gate(self.sep_act.lin(message))
            value = self.sep_value(value, edge_attr, edge_scalars)
            value = vec2heads(value, self.irreps_head, self.num_heads)
        else:
            message = vec2heads(self.sep(message, edge_attr, edge_scalars), self.irreps_mix_head, self.num_heads)
            alpha, value = message[:, :, :self.mul_alpha_head], message[:, :, self.mul_alpha_head:]
        alpha = self.alpha_act(alpha)
        alpha = torch.einsum("bik,aik->bi", alpha, self.alpha_dot)
        alpha = segment_softmax(alpha, edge_dst, node_input.shape[0]).unsqueeze(-1)
        if self.alpha_dropout is not None:
            alpha = self.alpha_dropout(alpha)
        attn = heads2vec(scatter_sum(value * alpha, edge_dst, node_input.shape[0]), self.irreps_head)
        if self.rescale_degree:
            deg = scatter_sum(torch.ones_like(edge_dst, dtype=attn.dtype), edge_dst, node_input.shape[0])
            attn = attn * deg.view(-1, 1)
        return self.proj(attn)


class DotProductAttention(nn.Module):
    """Scaled dot-product attention over irreps heads [ref: nets/dp_attention_transformer.py:68-160]: queries from the
    destination node, keys and values from ONE SeparableFCTP on the merged source/destination message (2H heads, the
    first H are keys), ScaleFactor (:45-66) on the queries, PyG softmax over incoming edges."""

    def __init__(self, irreps_node_input, irreps_node_attr, irreps_edge_attr, irreps_node_output, fc_neurons,
                 irreps_head, num_heads, irreps_pre_attn=None, rescale_degree=False, alpha_drop=0.1, proj_drop=0.1):
        super().__init__()
        self.irreps_node_input = Irreps(irreps_node_input)
        self.irreps_pre_attn = self.irreps_node_input if irreps_pre_attn is None else Irreps(irreps_pre_attn)
        self.irreps_head, self.num_heads = Irreps(irreps_head), num_heads
        assert not rescale_degree and proj_drop == 0.0
        heads_all = sort_irreps_even_first(self.irreps_head * num_heads)[0].simplify()
        self.query = LinearRS(self.irreps_node_input, heads_all)
        kv_heads = sort_irreps_even_first(self.irreps_head * num_heads * 2)[0].simplify()
        self.merge_src = LinearRS(self.irreps_node_input, self.irreps_pre_attn, bias=True)
        self.merge_dst = LinearRS(self.irreps_node_input, self.irreps_pre_attn, bias=False)
        self.key_value = SeparableFCTP(self.irreps_pre_attn, irreps_edge_attr, kv_heads, fc_neurons, use_activation=False)
        self.alpha_dropout = nn.Dropout(alpha_drop) if alpha_drop != 0.0 else None
        self.proj = LinearRS(heads_all, Irreps(irreps_node_output))
        # ScaleFactor: 1/sqrt(number of irreps of a head) per channel, 1/sqrt(2l+1) per irrep
        chan = 1.0 / (self.irreps_head.num_irreps ** 0.5)
        # plain fp64 attribute, not a buffer: Module.double() of an fp32-built model must not leave an fp32-rounded scale
        # (the reference multiplies by Python floats, :58-62; found by tests/test_reference_pin.py: 3e-9 / 3e-8)
        self._q_scale = torch.cat([torch.full((mul * ir.dim,), chan / ir.dim ** 0.5, dtype=torch.float64)
                                   for mul, ir in self.irreps_head])

    def forward(self, node_input, edge_src, edge_dst, edge_attr, edge_scalars):
        q = vec2heads(self.query(node_input), self.irreps_head, self.num_heads) * self._q_scale.to(node_input.dtype)
        kv = self.merge_src(node_input)[edge_src] + self.merge_dst(node_input)[edge_dst]
        kv = vec2heads(self.key_value(kv, edge_attr, edge_scalars), self.irreps_head, self.num_heads * 2)
        k, v = kv[:, :self.num_heads], kv[:, self.num_heads:]
        alpha = torch.einsum("bik,bik->bi", q[edge_dst], k)
        alpha = segment_softmax(alpha, edge_dst, node_input.shape[0]).unsqueeze(-1)
        if self.alpha_dropout is not None:
            alpha = self.alpha_dropout(alpha)
        attn = heads2vec(scatter_sum(v * alpha, edge_dst, node_input.shape[0]), self.irreps_head)
        return self.proj(attn)


class FCTPSwishGate(FullyConnectedTensorProductRescale):
    """FullyConnectedTensorProductRescaleSwishGate (graph_attention_transformer.py:128-154)."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, bias=True, rescale=True):
        gate = make_gate(irreps_out)
        super().__init__(irreps_in1, irreps_in2, gate.irreps_in, bias=bias, rescale=rescale)
        self.gate = gate

    def forward(self, x, y, weight=None):
        return self.gate(super().forward(x, y, weight))


class FeedForwardNetwork(nn.Module):
    def __init__(self, irreps_node_input, irreps_node_attr, irreps_node_output, irreps_mlp_mid=None, proj_drop=0.0):
        super().__init__()
        mid = Irreps(irreps_mlp_mid) if irreps_mlp_mid is not None else Irreps(irreps_node_input)
        self.fctp_1 = FCTPSwishGate(irreps_node_input, irreps_node_attr, mid, bias=True, rescale=_RESCALE)
        self.fctp_2 = FullyConnectedTensorProductRescale(mid, irreps_node_attr, irreps_node_output, bias=True,
                                                         rescale=_RESCALE)
        assert proj_drop == 0.0

    def forward(self, x, node_attr):
        return self.fctp_2(self.fctp_1(x, node_attr), node_attr)


class TransBlock(nn.Module):
    attn_name = "ga"

    def __init__(self, irreps_node_input, irreps_node_attr, irreps_edge_attr, irreps_node_output, fc_neurons,
                 irreps_head, num_heads, irreps_pre_attn=None, rescale_degree=False, nonlinear_message=False,
                 alpha_drop=0.1, proj_drop=0.1, drop_path_rate=0.0, irreps_mlp_mid=None, norm_layer="layer"):
        super().__init__()
        assert norm_layer == "layer"
        self.drop_path_rate = drop_path_rate
        irreps_node_input, irreps_node_output = Irreps(irreps_node_input), Irreps(irreps_node_output)
        self.norm_1 = EquivariantLayerNormV2(irreps_node_input)
        self._make_attention(irreps_node_input, irreps_node_attr, irreps_edge_attr, fc_neurons, irreps_head, num_heads,
                             irreps_pre_attn, rescale_degree, nonlinear_message, alpha_drop, proj_drop)
        self.norm_2 = EquivariantLayerNormV2(irreps_node_input)
        self.ffn = FeedForwardNetwork(irreps_node_input, irreps_node_attr, irreps_node_output, irreps_mlp_mid, proj_drop)
        self.ffn_shortcut = None
        if irreps_node_input != irreps_node_output:
            self.ffn_shortcut = FullyConnectedTensorProductRescale(irreps_node_input, irreps_node_attr,
                                                                   irreps_node_output, bias=True, rescale=_RESCALE)

    def _make_attention(self, irreps_node_input, irreps_node_attr, irreps_edge_attr, fc_neurons, irreps_head, num_heads,
                        irreps_pre_attn, rescale_degree, nonlinear_message, alpha_drop, proj_drop):
        self.ga = GraphAttention(irreps_node_input, irreps_node_attr, irreps_edge_attr, irreps_node_input, fc_neurons,
                                 irreps_head, num_heads, irreps_pre_attn, rescale_degree, nonlinear_message,
                                 alpha_drop, proj_drop)

    def _drop_path(self, x, batch):
        """GraphDropPath [ref: nets/drop.py:13-29,45-61]: one keep/drop draw per graph, kept rows scaled by 1/keep."""
        if self.drop_path_rate == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_path_rate
        r = torch.rand((int(batch.max()) + 1, 1), dtype=torch.float64)
        return x * ((keep + r).floor_() / keep).to(x.dtype)[batch]

    def forward(self, x, node_attr, edge_src, edge_dst, edge_attr, edge_scalars, batch=None):
        # [ref: nets/graph_attention_transformer.py:637-668]
        out = x + self._drop_path(getattr(self, self.attn_name)(self.norm_1(x), edge_src, edge_dst, edge_attr, edge_scalars), batch)
        f = self._drop_path(self.ffn(self.norm_2(out), node_attr), batch)
        if self.ffn_shortcut is not None:
            out = self.ffn_shortcut(out, node_attr)
        return out + f


class DPTransBlock(TransBlock):
    """[ref: nets/dp_attention_transformer.py:163-252] the same pre-norm block with DotProductAttention as `dpa`."""
    attn_name = "dpa"

    def _make_attention(self, irreps_node_input, irreps_node_attr, irreps_edge_attr, fc_neurons, irreps_head, num_heads,
                        irreps_pre_attn, rescale_degree, nonlinear_message, alpha_drop, proj_drop):
        self.dpa = DotProductAttention(irreps_node_input, irreps_node_attr, irreps_edge_attr, irreps_node_input,
                                       fc_neurons, irreps_head, num_heads, irreps_pre_attn, rescale_degree, alpha_drop,
                                       proj_drop)


class NodeEmbeddingNetwork(nn.Module):
    def __init__(self, irreps_node_embedding, max_atom_type, bias=True):
        super().__init__()
        self.max_atom_type = max_atom_type
        self.atom_type_lin = LinearRS(Irreps("{}x0e".format(max_atom_type)), Irreps(irreps_node_embedding), bias=bias)
        self.atom_type_lin.tp.weight.data.mul_(max_atom_type ** 0.5)

    def forward(self, node_atom):
        onehot = torch.nn.functional.one_hot(node_atom, self.max_atom_type).to(self.atom_type_lin.tp.weight.dtype)
        return self.atom_type_lin(onehot), onehot, onehot


class ScaledScatter(nn.Module):
    def __init__(self, avg):
        super().__init__()
        self.avg_aggregate_num = avg + 0.0

    def forward(self, x, index, dim_size):
        return scatter_sum(x, index, dim_size) / (self.avg_aggregate_num ** 0.5)


class EdgeDegreeEmbeddingNetwork(nn.Module):
    def __init__(self, irreps_node_embedding, irreps_edge_attr, fc_neurons, avg_aggregate_num):
        super().__init__()
        self.exp = LinearRS(Irreps("1x0e"), irreps_node_embedding, bias=_USE_BIAS, rescale=_RESCALE)
        self.dw = DepthwiseTensorProduct(irreps_node_embedding, irreps_edge_attr, irreps_node_embedding,
                                         internal_weights=False, bias=False)
        self.rad = RadialProfile(fc_neurons + [self.dw.tp.weight_numel])
        for sl, k in self.dw.slices_sqrt_k.values():
            self.rad.net[-1].weight.data[sl, :] *= k
            self.rad.offset.data[sl] *= k
        self.proj = LinearRS(self.dw.irreps_out.simplify(), irreps_node_embedding)
        self.scale_scatter = ScaledScatter(avg_aggregate_num)

    def forward(self, node_input, edge_attr, edge_scalars, edge_src, edge_dst):
        f = self.exp(torch.ones_like(node_input[:, 0:1]))
        e = self.proj(self.dw(f[edge_src], edge_attr, self.rad(edge_scalars)))
        return self.scale_scatter(e, edge_dst, f.shape[0])


class _Base(nn.Module):
    """Shared trunk of the three model variants."""
    block_cls = TransBlock

    def _build(self, irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius, number_of_basis,
               basis_type, fc_neurons, irreps_feature, irreps_head, num_heads, irreps_pre_attn, rescale_degree,
               nonlinear_message, irreps_mlp_mid, alpha_drop, max_atom_type, avg_degree, avg_nodes,
               drop_path_rate=0.0):
        self.max_radius, self.number_of_basis = max_radius, number_of_basis
        self.irreps_node_embedding = Irreps(irreps_node_embedding)
        self.irreps_feature = Irreps(irreps_feature)
        self.irreps_edge_attr = Irreps(irreps_sh)
        self.lmax_sh = self.irreps_edge_attr.lmax
        self.fc_neurons = [number_of_basis] + fc_neurons
        self.atom_embed = NodeEmbeddingNetwork(self.irreps_node_embedding, max_atom_type)
        if basis_type == "gaussian":
            self.rbf = GaussianRadialBasisLayer(number_of_basis, cutoff=max_radius)
        elif basis_type == "exp":
            self.rbf = ExpNormalSmearing(0.0, max_radius, number_of_basis)
        elif basis_type == "bessel":
            self.rbf = RadialBasis(number_of_basis, cutoff=max_radius, rbf={"name": "spherical_bessel"})
        else:
            raise ValueError
        self.edge_deg_embed = EdgeDegreeEmbeddingNetwork(self.irreps_node_embedding, self.irreps_edge_attr,
                                                         self.fc_neurons, avg_degree)
        self.blocks = nn.ModuleList()
        for i in range(num_layers):
            out = self.irreps_node_embedding if i != num_layers - 1 else self.irreps_feature
            self.blocks.append(self.block_cls(self.irreps_node_embedding, irreps_node_attr, self.irreps_edge_attr, out,
                                          self.fc_neurons, irreps_head, num_heads, irreps_pre_attn, rescale_degree,
                                          nonlinear_message, alpha_drop, 0.0, drop_path_rate, irreps_mlp_mid, "layer"))
        self.norm = EquivariantLayerNormV2(self.irreps_feature)
        if all(ir.l == 0 for _, ir in self.irreps_feature):  # otherwise the subclass brings its own head (OC20)
            self.head = nn.Sequential(LinearRS(self.irreps_feature, self.irreps_feature, rescale=_RESCALE),
                                      Activation(self.irreps_feature, [torch.nn.functional.silu]),
                                      LinearRS(self.irreps_feature, Irreps("1x0e"), rescale=_RESCALE))
        self.scale_scatter = ScaledScatter(avg_nodes)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _trunk(self, node_embedding, pos, batch, edge_src, edge_dst, edge_vec, num_graphs):
        x, edge_sh, edge_emb = self._features(node_embedding, batch, edge_src, edge_dst, edge_vec)
        out = self.head(x, edge_src, edge_dst, edge_sh, edge_emb) if isinstance(self.head, GraphAttention) else self.head(x)
        return self.scale_scatter(out, batch, num_graphs)

    def _features(self, node_embedding, batch, edge_src, edge_dst, edge_vec, extra=None):
        edge_sh = spherical_harmonics(self.lmax_sh, edge_vec, normalize=True, normalization="component")
        edge_len = edge_vec.norm(dim=1)
        edge_emb = self.rbf(edge_len)
        x = node_embedding + self.edge_deg_embed(node_embedding, edge_sh, edge_emb, edge_src, edge_dst)
        if extra is not None:
            x = x + extra
        node_attr = torch.ones_like(x[:, 0:1])
        for blk in self.blocks:
            x = blk(x, node_attr, edge_src, edge_dst, edge_sh, edge_emb, batch)
        return self.norm(x), edge_sh, edge_emb


_QM9_AVG_NUM_NODES = 18.03065905448718
_QM9_AVG_DEGREE = 15.57930850982666


class GraphAttentionTransformer(_Base):
    def __init__(self, irreps_in="5x0e", irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6,
                 irreps_node_attr="1x0e", irreps_sh="1x0e+1x1e+1x2e", max_radius=5.0, number_of_basis=128,
                 basis_type="gaussian", fc_neurons=[64, 64], irreps_feature="512x0e",
                 irreps_head="32x0e+16x1o+8x2e", num_heads=4, irreps_pre_attn=None, rescale_degree=False,
                 nonlinear_message=False, irreps_mlp_mid="128x0e+64x1e+32x2e", norm_layer="layer", alpha_drop=0.2,
                 proj_drop=0.0, out_drop=0.0, drop_path_rate=0.0, mean=None, std=None, scale=None, atomref=None):
        super().__init__()
        self.task_mean, self.task_std, self.scale = mean, std, scale
        self.register_buffer("atomref", atomref)
        self._build(irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius, number_of_basis,
                    basis_type, list(fc_neurons), irreps_feature, irreps_head, num_heads, irreps_pre_attn,
                    rescale_degree, nonlinear_message, irreps_mlp_mid, alpha_drop, 5, _QM9_AVG_DEGREE,
                    _QM9_AVG_NUM_NODES)

    def forward(self, f_in, pos, batch, node_atom, **kwargs):
        edge_src, edge_dst = radius_graph(pos, self.max_radius, batch, 1000)
        edge_vec = pos[edge_src] - pos[edge_dst]
        node_atom = node_atom.new_tensor([-1, 0, -1, -1, -1, -1, 1, 2, 3, 4])[node_atom]
        emb, _, _ = self.atom_embed(node_atom)
        out = self._trunk(emb, pos, batch, edge_src, edge_dst, edge_vec, int(batch.max()) + 1)
        return out if self.scale is None else self.scale * out


class GraphAttentionTransformerMD17(_Base):
    def __init__(self, irreps_in="64x0e", irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6,
                 irreps_node_attr="1x0e", irreps_sh="1x0e+1x1e+1x2e", max_radius=5.0, number_of_basis=128,
                 basis_type="gaussian", fc_neurons=[64, 64], irreps_feature="512x0e",
                 irreps_head="32x0e+16x1o+8x2e", num_heads=4, irreps_pre_attn=None, rescale_degree=False,
                 nonlinear_message=False, irreps_mlp_mid="128x0e+64x1e+32x2e", use_attn_head=False,
                 norm_layer="layer", alpha_drop=0.2, proj_drop=0.0, out_drop=0.0, drop_path_rate=0.0, mean=None,
                 std=None, scale=None, atomref=None):
        super().__init__()
        self.task_mean, self.task_std, self.scale = mean, std, scale
        self.register_buffer("atomref", atomref)
        self._build(irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius, number_of_basis,
                    basis_type, list(fc_neurons), irreps_feature, irreps_head, num_heads, irreps_pre_attn,
                    rescale_degree, nonlinear_message, irreps_mlp_mid, alpha_drop, 64, _QM9_AVG_DEGREE,
                    _QM9_AVG_NUM_NODES)
        self.use_attn_head = use_attn_head
        if use_attn_head:  # [ref: nets/graph_attention_transformer_md17.py:196-207, :304-308]
            self.head = GraphAttention(self.irreps_feature, irreps_node_attr, self.irreps_edge_attr, Irreps("1x0e"),
                                       self.fc_neurons, irreps_head, num_heads, irreps_pre_attn, rescale_degree,
                                       nonlinear_message, alpha_drop, proj_drop)
            self.apply(self._init_weights)

    @torch.enable_grad()
    def forward(self, node_atom, pos, batch):
        pos = pos.requires_grad_(True)
        edge_src, edge_dst = radius_graph(pos, self.max_radius, batch, 1000)
        edge_vec = pos[edge_src] - pos[edge_dst]
        emb, _, _ = self.atom_embed(node_atom)
        energy = self._trunk(emb, pos, batch, edge_src, edge_dst, edge_vec, int(batch.max()) + 1)
        if self.scale is not None:
            energy = self.scale * energy
        forces = -1 * torch.autograd.grad(energy, pos, grad_outputs=torch.ones_like(energy), create_graph=True)[0]
        return energy, forces


class GraphAttentionTransformerOC20(_Base):
    """Edges (edge_index + per-edge Cartesian offsets) are supplied by the caller, which is what ocpmodels'
    radius_graph_pbc/get_pbc_distances produce upstream (graph_attention_transformer_oc20.py:267-302).  Returns the
    energy, or (energy, per-node auxiliary vectors) with use_auxiliary_task (IS2RS head, :182-208, :352-381)."""

    def __init__(self, num_atoms=None, bond_feat_dim=None, num_targets=1, irreps_node_embedding="256x0e+128x1e",
                 num_layers=6, irreps_node_attr="1x0e", use_node_attr=False, irreps_sh="1x0e+1x1e", max_radius=6.0,
                 number_of_basis=128, fc_neurons=[64, 64], irreps_feature="512x0e", irreps_head="32x0e+16x1e",
                 num_heads=8, irreps_pre_attn=None, rescale_degree=False, nonlinear_message=False,
                 irreps_mlp_mid="768x0e+384x1e", norm_layer="layer", alpha_drop=0.2, proj_drop=0.0, out_drop=0.0,
                 drop_path_rate=0.0, use_auxiliary_task=False, auxiliary_head_dropout=True, use_attention_head=False,
                 max_neighbors=50, **unused):
        super().__init__()
        assert not use_node_attr
        self.max_neighbors = max_neighbors
        self._build(irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius, number_of_basis,
                    "gaussian", list(fc_neurons), irreps_feature, irreps_head, num_heads, irreps_pre_attn,
                    rescale_degree, nonlinear_message, irreps_mlp_mid, alpha_drop, 84, 23.395238876342773, 77.81317,
                    drop_path_rate=drop_path_rate)
        self.tag_embed = NodeEmbeddingNetwork(self.irreps_node_embedding, 3)
        # the OC20 energy head reads the scalar channels of the feature only [ref: :169-179]
        scalars = Irreps([(m, ir) for m, ir in self.irreps_feature if ir.l == 0 and ir.p == 1])
        self.head = nn.Sequential(LinearRS(self.irreps_feature, scalars, rescale=_RESCALE),
                                  Activation(scalars, [torch.nn.functional.silu]),
                                  LinearRS(scalars, Irreps("1x0e")))
        self.use_auxiliary_task, self.use_attention_head = use_auxiliary_task, use_attention_head
        irreps_aux = Irreps("1x1o") if Irrep("1o") in self.irreps_feature else Irreps("1x1e")  # [ref: :185-187]
        head_drop = alpha_drop if auxiliary_head_dropout else 0.0

        def attention(irreps_out):
            if self.block_cls is DPTransBlock:  # [ref: nets/dp_attention_transformer_oc20.py:146-151]
                return DotProductAttention(self.irreps_feature, irreps_node_attr, self.irreps_edge_attr, irreps_out,
                                           self.fc_neurons, irreps_head, num_heads, irreps_pre_attn, rescale_degree,
                                           alpha_drop, proj_drop=0.0)
            return GraphAttention(self.irreps_feature, irreps_node_attr, self.irreps_edge_attr, irreps_out,
                                  self.fc_neurons, irreps_head, num_heads, irreps_pre_attn, rescale_degree,
                                  nonlinear_message, alpha_drop=head_drop, proj_drop=0.0)
        if use_auxiliary_task and not use_attention_head:
            self.auxiliary_head = attention(irreps_aux)
        if use_attention_head:
            irreps_out = Irreps("1x0e") + irreps_aux if use_auxiliary_task else Irreps("1x0e")
            self.head = attention(irreps_out)
            self.head_skip_connect = LinearRS(self.irreps_feature, irreps_out)
        self.apply(self._init_weights)
        # registration order of the reference: tag_embed right after atom_embed [ref: :146-147]
        mods = dict(self._modules)
        tag = mods.pop("tag_embed")
        self._modules = {k2: v2 for k, v in mods.items() for k2, v2 in (((k, v), ("tag_embed", tag)) if k == "atom_embed"
                                                                         else ((k, v),))}

    def forward(self, atomic_numbers, tags, pos, batch, edge_index=None, offsets=None):
        if edge_index is None:
            edge_src, edge_dst = radius_graph(pos, self.max_radius, batch, self.max_neighbors)
        else:
            edge_src, edge_dst = edge_index[0], edge_index[1]
        edge_vec = pos[edge_src] - pos[edge_dst]
        if offsets is not None:
            edge_vec = edge_vec + offsets
        emb, _, _ = self.atom_embed(atomic_numbers.long())
        tag, _, _ = self.tag_embed(tags.long())
        num_graphs = int(batch.max()) + 1
        x, edge_sh, edge_emb = self._features(emb + tag, batch, edge_src, edge_dst, edge_vec)
        if self.use_attention_head:
            out = self.head(x, edge_src, edge_dst, edge_sh, edge_emb) + self.head_skip_connect(x)
            energy = self.scale_scatter(out[:, 0:1], batch, num_graphs)
            return (energy, out[:, 1:4]) if self.use_auxiliary_task else energy
        energy = self.scale_scatter(self.head(x), batch, num_graphs)
        if self.use_auxiliary_task:
            return energy, self.auxiliary_head(x, edge_src, edge_dst, edge_sh, edge_emb)
        return energy


class Equiformer_MD17_DeNS(_Base):
    """[ref: nets/equiformer_md17_dens.py:55-354] MD17 trunk + force encoding of the uncorrupted structure in the input
    embedding + scalar-channel energy head + GraphAttention head predicting the position noise of corrupted atoms."""

    def __init__(self, irreps_in="64x0e", irreps_equivariant_inputs="1x0e+1x1e+1x2e",
                 irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_node_attr="1x0e",
                 irreps_sh="1x0e+1x1e+1x2e", max_radius=5.0, number_of_basis=32, basis_type="exp", fc_neurons=[64, 64],
                 irreps_feature="512x0e+256x1e+128x2e", irreps_head="32x0e+16x1e+8x2e", num_heads=4,
                 irreps_pre_attn="128x0e+64x1e+32x2e", rescale_degree=False, nonlinear_message=True,
                 irreps_mlp_mid="128x0e+64x1e+32x2e", norm_layer="layer", alpha_drop=0.0, proj_drop=0.0, out_drop=0.0,
                 drop_path_rate=0.0, mean=None, std=None, scale=None, atomref=None, use_force_encoding=True):
        super().__init__()
        self.task_mean, self.task_std, self.scale = mean, std, scale
        self.register_buffer("atomref", atomref)
        self.use_force_encoding = use_force_encoding
        self.irreps_eq_in = Irreps(irreps_equivariant_inputs)
        self._build(irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius, number_of_basis,
                    basis_type, list(fc_neurons), irreps_feature, irreps_head, num_heads, irreps_pre_attn,
                    rescale_degree, nonlinear_message, irreps_mlp_mid, alpha_drop, 64, _QM9_AVG_DEGREE,
                    _QM9_AVG_NUM_NODES, drop_path_rate=drop_path_rate)
        if hasattr(self, "head"):
            del self.head
        self.force_embed = LinearRS(self.irreps_eq_in, self.irreps_node_embedding, rescale=_RESCALE)
        scalars = Irreps([(m, ir) for m, ir in self.irreps_feature if ir.l == 0 and ir.p == 1])
        self.energy_head = nn.Sequential(LinearRS(self.irreps_feature, scalars, rescale=_RESCALE),
                                         Activation(scalars, [torch.nn.functional.silu]),
                                         LinearRS(scalars, Irreps("1x0e"), rescale=_RESCALE))
        self.denoising_pos_head = GraphAttention(self.irreps_feature, irreps_node_attr, self.irreps_edge_attr,
                                                 Irreps("1x1e"), self.fc_neurons, irreps_head, num_heads,
                                                 irreps_pre_attn, rescale_degree, nonlinear_message, alpha_drop, proj_drop)
        self.apply(self._init_weights)
        # module (= parameter) order of the reference [ref: :119-176]
        order = ["atom_embed", "rbf", "edge_deg_embed", "force_embed", "blocks", "norm", "energy_head", "scale_scatter",
                 "denoising_pos_head"]
        assert sorted(order) == sorted(self._modules)
        self._modules = {k: self._modules[k] for k in order}

    @torch.enable_grad()
    def forward(self, data):
        node_atom, batch = data.z, data.batch
        pos = data.pos.requires_grad_(True)
        edge_src, edge_dst = radius_graph(pos, self.max_radius, batch, 1000)
        edge_vec = pos[edge_src] - pos[edge_dst]
        emb, _, _ = self.atom_embed(node_atom)
        if hasattr(data, "force") and self.use_force_encoding:
            force_sh = spherical_harmonics(self.irreps_eq_in.lmax, data.force.to(pos.dtype), normalize=True,
                                           normalization="component")
            force_sh = force_sh * data.noise_mask.to(pos.dtype).view(-1, 1)
            force_sh = force_sh * (data.force.to(pos.dtype).norm(dim=1, keepdim=True) / math.sqrt(3.0))
        else:
            force_sh = torch.zeros((pos.shape[0], self.irreps_eq_in.dim), dtype=pos.dtype)
        x, edge_sh, edge_emb = self._features(emb, batch, edge_src, edge_dst, edge_vec, extra=self.force_embed(force_sh))
        energy = self.energy_head(x)
        if hasattr(data, "denoising_mask") and not self.use_force_encoding:
            energy = energy * (~data.denoising_mask).to(energy.dtype).view(-1, 1)
        energy = self.scale_scatter(energy, batch, int(batch.max()) + 1)
        if self.scale is not None:
            energy = self.scale * energy
        forces = -1 * torch.autograd.grad(energy, pos, grad_outputs=torch.ones_like(energy), create_graph=True)[0]
        if not hasattr(data, "noise_mask"):
            return energy, forces
        den = self.denoising_pos_head(x, edge_src, edge_dst, edge_sh, edge_emb)
        out = torch.where(data.noise_mask.view(-1, 1), den, forces)
        if not self.use_force_encoding:
            out = out * (~data.denoising_pos_mask).to(out.dtype).view(-1, 1)
        return energy, out


class DotProductAttentionTransformer(GraphAttentionTransformer):
    """[ref: nets/dp_attention_transformer.py:255-411] the QM9 model with DPTransBlock in place of TransBlock."""
    block_cls = DPTransBlock


class DotProductAttentionTransformerMD17(GraphAttentionTransformerMD17):
    """[ref: nets/dp_attention_transformer_md17.py:57-235]"""
    block_cls = DPTransBlock


class DotProductAttentionTransformerOC20(GraphAttentionTransformerOC20):
    """[ref: nets/dp_attention_transformer_oc20.py:36-347] (no attention head in this family)"""
    block_cls = DPTransBlock


# ---------------------------------------------------------------------------- factories (registered names)
def graph_attention_transformer_l2(irreps_in, radius, num_basis=128, atomref=None, task_mean=None, task_std=None,
                                   **kwargs):
    return GraphAttentionTransformer(
        irreps_in=irreps_in, irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_node_attr="1x0e",
        irreps_sh="1x0e+1x1e+1x2e", max_radius=radius, number_of_basis=num_basis, fc_neurons=[64, 64],
        irreps_feature="512x0e", irreps_head="32x0e+16x1e+8x2e", num_heads=4, rescale_degree=False,
        nonlinear_message=False, irreps_mlp_mid="384x0e+192x1e+96x2e", alpha_drop=0.2, mean=task_mean, std=task_std,
        atomref=atomref)


def graph_attention_transformer_nonlinear_l2(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                             task_std=None, **kwargs):
    return GraphAttentionTransformer(
        irreps_in=irreps_in, irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_node_attr="1x0e",
        irreps_sh="1x0e+1x1e+1x2e", max_radius=radius, number_of_basis=num_basis, fc_neurons=[64, 64],
        irreps_feature="512x0e", irreps_head="32x0e+16x1e+8x2e", num_heads=4, rescale_degree=False,
        nonlinear_message=True, irreps_mlp_mid="384x0e+192x1e+96x2e", alpha_drop=0.2, mean=task_mean, std=task_std,
        atomref=atomref)


def graph_attention_transformer_nonlinear_exp_l2_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                      task_std=None, **kwargs):
    return GraphAttentionTransformerMD17(
        irreps_in=irreps_in, irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_node_attr="1x0e",
        irreps_sh="1x0e+1x1e+1x2e", max_radius=radius, number_of_basis=num_basis, fc_neurons=[64, 64],
        basis_type="exp", irreps_feature="512x0e", irreps_head="32x0e+16x1e+8x2e", num_heads=4,
        rescale_degree=False, nonlinear_message=True, irreps_mlp_mid="384x0e+192x1e+96x2e", alpha_drop=0.0,
        mean=task_mean, std=task_std, atomref=atomref)


def graph_attention_transformer_nonlinear_exp_l3_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                      task_std=None, **kwargs):
    return GraphAttentionTransformerMD17(
        irreps_in=irreps_in, irreps_node_embedding="128x0e+64x1e+64x2e+32x3e", num_layers=6, irreps_node_attr="1x0e",
        irreps_sh="1x0e+1x1e+1x2e+1x3e", max_radius=radius, number_of_basis=num_basis, fc_neurons=[64, 64],
        basis_type="exp", irreps_feature="512x0e", irreps_head="32x0e+16x1e+16x2e+8x3e", num_heads=4,
        rescale_degree=False, nonlinear_message=True, irreps_mlp_mid="384x0e+192x1e+192x2e+96x3e", alpha_drop=0.0,
        mean=task_mean, std=task_std, atomref=atomref)


def oc20_l1_256_nonlinear(**kwargs):
    """oc20/configs/is2re/all/graph_attention_transformer/l1_256_nonlinear_g@2_local.yml model section."""
    cfg = dict(irreps_node_embedding="256x0e+128x1e", num_layers=6, irreps_sh="1x0e+1x1e", max_radius=5.0,
               number_of_basis=128, fc_neurons=[64, 64], irreps_feature="512x0e", irreps_head="32x0e+16x1e",
               num_heads=8, irreps_pre_attn="256x0e+128x1e", nonlinear_message=True,
               irreps_mlp_mid="768x0e+384x1e", alpha_drop=0.2, max_neighbors=500)
    cfg.update(kwargs)
    return GraphAttentionTransformerOC20(**cfg)


def dot_product_attention_transformer_l2(irreps_in, radius, num_basis=128, atomref=None, task_mean=None, task_std=None,
                                         **kwargs):
    """[ref: nets/dp_attention_transformer.py:414-431]"""
    return DotProductAttentionTransformer(
        irreps_in=irreps_in, irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_node_attr="1x0e",
        irreps_sh="1x0e+1x1e+1x2e", max_radius=radius, number_of_basis=num_basis, fc_neurons=[64, 64],
        irreps_feature="512x0e", irreps_head="32x0e+16x1e+8x2e", num_heads=4, rescale_degree=False,
        nonlinear_message=False, irreps_mlp_mid="384x0e+192x1e+96x2e", alpha_drop=0.2, mean=task_mean, std=task_std,
        atomref=atomref)


def dot_product_attention_transformer_exp_l2_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                  task_std=None, **kwargs):
    """[ref: nets/dp_attention_transformer_md17.py:238-254]"""
    return DotProductAttentionTransformerMD17(
        irreps_in=irreps_in, irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_node_attr="1x0e",
        irreps_sh="1x0e+1x1e+1x2e", max_radius=radius, number_of_basis=num_basis, fc_neurons=[64, 64],
        basis_type="exp", irreps_feature="512x0e", irreps_head="32x0e+16x1e+8x2e", num_heads=4, rescale_degree=False,
        nonlinear_message=False, irreps_mlp_mid="384x0e+192x1e+96x2e", alpha_drop=0.0, mean=task_mean, std=task_std,
        atomref=atomref)


ENTRYPOINTS = {
    "dot_product_attention_transformer_l2": dot_product_attention_transformer_l2,
    "dot_product_attention_transformer_exp_l2_md17": dot_product_attention_transformer_exp_l2_md17,
    "graph_attention_transformer_l2": graph_attention_transformer_l2,
    "graph_attention_transformer_nonlinear_l2": graph_attention_transformer_nonlinear_l2,
    "graph_attention_transformer_nonlinear_exp_l2_md17": graph_attention_transformer_nonlinear_exp_l2_md17,
    "graph_attention_transformer_nonlinear_exp_l3_md17": graph_attention_transformer_nonlinear_exp_l3_md17,
}


def model_entrypoint(name):
    return ENTRYPOINTS[name]
