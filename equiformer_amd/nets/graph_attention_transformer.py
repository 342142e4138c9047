"""QM9 graph-attention transformer (Equiformer) on the MI355X hot path.

Drop-in for the reference's nets/graph_attention_transformer.py: class `GraphAttentionTransformer` (:736-899) and the
registered factories (:902-1016), same constructor arguments, `forward(f_in, pos, batch, node_atom, **kwargs)`,
`no_weight_decay()`, `task_mean/task_std`, and the same state_dict keys.  Compute = libequiformer_hip.so.
"""
import torch
from torch import nn

from .. import ops
from ..graph import EdgeGraph
from ..irreps import Irreps
from .layers import (Activation, EdgeContext, EdgeDegreeEmbeddingNetwork, EquivariantLayerNormV2,  # noqa: F401
                     FeedForwardNetwork, FullyConnectedTensorProductRescale, GaussianRadialBasisLayer, GraphAttention,
                     LinearRS, NodeEmbeddingNetwork, RadialBank, RadialBasis, ScaledScatter, SeparableFCTP, TransBlock, get_norm_layer)
from .registry import register_model

_RESCALE = True
_USE_BIAS = True

# QM9 [ref: nets/graph_attention_transformer.py:32-36]
_MAX_ATOM_TYPE = 5
_AVG_NUM_NODES = 18.03065905448718
_AVG_DEGREE = 15.57930850982666


class _Trunk(nn.Module):
    """Everything the QM9 / MD17 / OC20 variants share: embeddings, blocks, head, pooling."""

    def _build_trunk(self, irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius,
                     number_of_basis, fc_neurons, irreps_feature, irreps_head, num_heads, irreps_pre_attn,
                     rescale_degree, nonlinear_message, irreps_mlp_mid, norm_layer, alpha_drop, proj_drop, out_drop,
                     drop_path_rate, max_atom_type, avg_degree, avg_num_nodes):
        if out_drop != 0.0:
            raise NotImplementedError("out_drop != 0 is not used by any registered model")
        self.max_radius = max_radius
        self.number_of_basis = number_of_basis
        self.alpha_drop, self.proj_drop, self.out_drop = alpha_drop, proj_drop, out_drop
        self.drop_path_rate = drop_path_rate
        self.norm_layer = norm_layer
        self.irreps_node_attr = Irreps(irreps_node_attr)
        self.irreps_node_embedding = Irreps(irreps_node_embedding)
        self.lmax = self.irreps_node_embedding.lmax
        self.irreps_feature = Irreps(irreps_feature)
        self.num_layers = num_layers
        self.irreps_edge_attr = (Irreps(irreps_sh) if irreps_sh is not None
                                 else Irreps.spherical_harmonics(self.lmax))
        self.lmax_sh = len(self.irreps_edge_attr) - 1
        self.fc_neurons = [number_of_basis] + list(fc_neurons)
        self.irreps_head = Irreps(irreps_head)
        self.num_heads = num_heads
        self.irreps_pre_attn = irreps_pre_attn
        self.rescale_degree = rescale_degree
        self.nonlinear_message = nonlinear_message
        self.irreps_mlp_mid = Irreps(irreps_mlp_mid)

        self.atom_embed = NodeEmbeddingNetwork(self.irreps_node_embedding, max_atom_type)
        self._make_rbf()
        self.edge_deg_embed = EdgeDegreeEmbeddingNetwork(self.irreps_node_embedding, self.irreps_edge_attr,
                                                         self.fc_neurons, avg_degree)
        self.blocks = nn.ModuleList()
        for i in range(num_layers):
            out = self.irreps_node_embedding if i != num_layers - 1 else self.irreps_feature
            self.blocks.append(self._block_cls(
                irreps_node_input=self.irreps_node_embedding, irreps_node_attr=self.irreps_node_attr,
                irreps_edge_attr=self.irreps_edge_attr, irreps_node_output=out, fc_neurons=self.fc_neurons,
                irreps_head=self.irreps_head, num_heads=num_heads, irreps_pre_attn=irreps_pre_attn,
                rescale_degree=rescale_degree, nonlinear_message=nonlinear_message, alpha_drop=alpha_drop,
                proj_drop=proj_drop, drop_path_rate=drop_path_rate, irreps_mlp_mid=self.irreps_mlp_mid,
                norm_layer=norm_layer))
        self.norm = get_norm_layer(norm_layer)(self.irreps_feature)
        self.out_dropout = None
        if all(ir.l == 0 for _, ir in self.irreps_feature):  # otherwise the subclass brings its own head (OC20)
            self.head = nn.Sequential(
                LinearRS(self.irreps_feature, self.irreps_feature, rescale=_RESCALE),
                Activation(self.irreps_feature, kind="silu"),
                LinearRS(self.irreps_feature, Irreps("1x0e"), rescale=_RESCALE))
        self.scale_scatter = ScaledScatter(avg_num_nodes)
        self.apply(self._init_weights)

    def _make_rbf(self):
        if self.basis_type == "gaussian":
            self.rbf = GaussianRadialBasisLayer(self.number_of_basis, cutoff=self.max_radius)
        elif self.basis_type == "bessel":
            self.rbf = RadialBasis(self.number_of_basis, cutoff=self.max_radius, rbf={"name": "spherical_bessel"})
        else:
            raise ValueError

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        no_wd_list = []
        named = {name for name, _ in self.named_parameters()}
        for module_name, module in self.named_modules():
            if isinstance(module, (nn.Linear, nn.LayerNorm, EquivariantLayerNormV2, GaussianRadialBasisLayer, RadialBasis)):
                for parameter_name, _ in module.named_parameters():
                    if isinstance(module, nn.Linear) and "weight" in parameter_name:
                        continue
                    full = module_name + "." + parameter_name
                    assert full in named
                    no_wd_list.append(full)
        return set(no_wd_list)

    def set_fused(self, flag):
        """True (default): fused SeparableFCTP kernels (eqf_sfc_*); "legacy": the per-degree DTP-generating GEMMs
        (eqf_dtp_linear_*); False: un-fused DTP + linear.  The slower paths are the on-device cross-checks."""
        for m in self.modules():
            if hasattr(m, "use_fused"):
                m.use_fused = "legacy" if flag == "legacy" else bool(flag)

    def _radial_bank(self):
        bank = self.__dict__.get("_bank")
        if bank is None:
            mods = [self.edge_deg_embed.rad]
            for attn in [blk.attention for blk in self.blocks] + self._attention_heads():
                mods.append(attn.radial_module())
            bank = RadialBank(mods)
            self.__dict__["_bank"] = bank  # plain attribute (not a sub-module, not copied into state_dict)
        return bank if (bank.ok and self.use_radial_bank) else None

    use_radial_bank = True
    _block_cls = TransBlock

    def late_gradient_parameters(self):
        """Parameters whose gradient is complete only at the end of backward although they belong to late layers: the
        members of the radial bank are evaluated at the start of the forward.  (Read by parallel.FlatGradAllReduce to lay
        them out in the bucket that is reduced last.)"""
        bank = self._radial_bank()
        return [p for m in bank.modules for p in m.parameters()] if bank is not None else []

    def _attention_heads(self):
        """GraphAttention modules that read the final features (OC20 auxiliary / attention heads); none by default."""
        return []

    def _trunk_forward(self, node_embedding, pos, graph, offsets=None):
        node_features, ectx = self._trunk_features(node_embedding, pos, graph, offsets)
        if isinstance(self.head, GraphAttention):  # attention head (MD17 use_attn_head)
            outputs = self.head(node_features, ectx=ectx)
        else:
            outputs = self.head(node_features)
        return self.scale_scatter(outputs, graph.mol_ptr, graph.batch, graph.num_graphs)

    def _trunk_features(self, node_embedding, pos, graph, offsets=None, extra=None):
        _, edge_length, edge_sh = ops.edge_geometry(pos, offsets, graph, self.lmax_sh)
        edge_scalars = self.rbf(edge_length)
        # the radial MLPs of all blocks side by side (RadialBank): first- and second-order (forces taken with create_graph:
        # MD17 / DeNS training) alike since round 5
        bank = self._radial_bank()
        ectx = EdgeContext(graph, edge_sh, edge_scalars, radial_bank=bank)
        # residual stream kept as a lazy pair (a, b) = a + b: each add is folded into the layer norm that consumes it
        a, b = node_embedding, self.edge_deg_embed(node_embedding, ectx)
        if extra is not None:  # a further per-node term of the input embedding (DeNS force encoding)
            b = b + extra
        for blk in self.blocks:
            a, b = blk.forward_pair(a, b, node_attr=None, ectx=ectx)
        _, node_features = self.norm.forward_sum(a, b)
        return node_features, ectx


class GraphAttentionTransformer(_Trunk):
    def __init__(self, irreps_in="5x0e", irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6,
                 irreps_node_attr="1x0e", irreps_sh="1x0e+1x1e+1x2e", max_radius=5.0, number_of_basis=128,
                 basis_type="gaussian", fc_neurons=[64, 64], irreps_feature="512x0e",
                 irreps_head="32x0e+16x1o+8x2e", num_heads=4, irreps_pre_attn=None, rescale_degree=False,
                 nonlinear_message=False, irreps_mlp_mid="128x0e+64x1e+32x2e", norm_layer="layer", alpha_drop=0.2,
                 proj_drop=0.0, out_drop=0.0, drop_path_rate=0.0, mean=None, std=None, scale=None, atomref=None):
        super().__init__()
        self.task_mean, self.task_std, self.scale = mean, std, scale
        self.register_buffer("atomref", atomref)
        self.irreps_node_input = Irreps(irreps_in)
        self.basis_type = basis_type
        self._build_trunk(irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius,
                          number_of_basis, fc_neurons, irreps_feature, irreps_head, num_heads, irreps_pre_attn,
                          rescale_degree, nonlinear_message, irreps_mlp_mid, norm_layer, alpha_drop, proj_drop,
                          out_drop, drop_path_rate, _MAX_ATOM_TYPE, _AVG_DEGREE, _AVG_NUM_NODES)

    def forward(self, f_in, pos, batch, node_atom, **kwargs) -> torch.Tensor:
        graph = kwargs.get("graph")
        if graph is None:
            graph = EdgeGraph.from_radius(pos, batch, self.max_radius, max_num_neighbors=1000)
        # atomic number -> type index  [ref: nets/graph_attention_transformer.py:872]
        # (the table lives on the device, built once per device: no host-to-device copy per step, legal under graph capture)
        table = self.__dict__.get("_z_table")
        if table is None or table.device != node_atom.device or table.dtype != node_atom.dtype:
            table = node_atom.new_tensor([-1, 0, -1, -1, -1, -1, 1, 2, 3, 4])
            self.__dict__["_z_table"] = table  # plain attribute: not a buffer, not in the state_dict
        node_atom = table[node_atom]
        atom_embedding, _, _ = self.atom_embed(node_atom)
        outputs = self._trunk_forward(atom_embedding, pos.to(torch.float32).contiguous(), graph)
        if self.scale is not None:
            outputs = self.scale * outputs
        return outputs


def _l2_kwargs(irreps_in, radius, num_basis, task_mean, task_std, atomref, **over):
    kw = dict(irreps_in=irreps_in, irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_node_attr="1x0e",
              irreps_sh="1x0e+1x1e+1x2e", max_radius=radius, number_of_basis=num_basis, fc_neurons=[64, 64],
              irreps_feature="512x0e", irreps_head="32x0e+16x1e+8x2e", num_heads=4, irreps_pre_attn=None,
              rescale_degree=False, nonlinear_message=True, irreps_mlp_mid="384x0e+192x1e+96x2e", norm_layer="layer",
              alpha_drop=0.2, proj_drop=0.0, out_drop=0.0, drop_path_rate=0.0, mean=task_mean, std=task_std,
              scale=None, atomref=atomref)
    kw.update(over)
    return kw


@register_model
def graph_attention_transformer_l2(irreps_in, radius, num_basis=128, atomref=None, task_mean=None, task_std=None,
                                   **kwargs):
    return GraphAttentionTransformer(**_l2_kwargs(irreps_in, radius, num_basis, task_mean, task_std, atomref,
                                                  nonlinear_message=False))


@register_model
def graph_attention_transformer_nonlinear_l2(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                             task_std=None, **kwargs):
    return GraphAttentionTransformer(**_l2_kwargs(irreps_in, radius, num_basis, task_mean, task_std, atomref))


@register_model
def graph_attention_transformer_nonlinear_l2_e3(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                task_std=None, **kwargs):
    return GraphAttentionTransformer(**_l2_kwargs(
        irreps_in, radius, num_basis, task_mean, task_std, atomref,
        irreps_node_embedding="128x0e+32x0o+32x1e+32x1o+16x2e+16x2o", irreps_sh="1x0e+1x1o+1x2e",
        irreps_head="32x0e+8x0o+8x1e+8x1o+4x2e+4x2o", irreps_mlp_mid="384x0e+96x0o+96x1e+96x1o+48x2e+48x2o"))


@register_model
def graph_attention_transformer_nonlinear_bessel_l2(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                    task_std=None, **kwargs):
    return GraphAttentionTransformer(**_l2_kwargs(irreps_in, radius, num_basis, task_mean, task_std, atomref,
                                                  basis_type="bessel"))


@register_model
def graph_attention_transformer_nonlinear_bessel_l2_drop01(irreps_in, radius, num_basis=128, atomref=None,
                                                           task_mean=None, task_std=None, **kwargs):
    return GraphAttentionTransformer(**_l2_kwargs(irreps_in, radius, num_basis, task_mean, task_std, atomref,
                                                  basis_type="bessel", alpha_drop=0.1))


@register_model
def graph_attention_transformer_nonlinear_bessel_l2_drop00(irreps_in, radius, num_basis=128, atomref=None,
                                                           task_mean=None, task_std=None, **kwargs):
    return GraphAttentionTransformer(**_l2_kwargs(irreps_in, radius, num_basis, task_mean, task_std, atomref,
                                                  basis_type="bessel", alpha_drop=0.0))
