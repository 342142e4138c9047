"""OC20 IS2RE variant (energy head), drop-in for nets/graph_attention_transformer_oc20.py:73-386 of the reference.

The reference registers the class in ocpmodels' registry as "graph_attention_transformer" and receives an ocpmodels
`Batch`; ocpmodels is un-vendored, so the class is exposed here under the same name through this package's registry
(`oc20_graph_attention_transformer`) and accepts any object with the same attributes
(`pos, batch, atomic_numbers, tags, natoms`, and for periodic inputs either `cell` -- the neighbour search then runs
on the GPU (`EdgeGraph.from_radius_pbc`, the `otf_graph=True` path of the YAML config) -- or a precomputed `edge_index` +
per-edge Cartesian `offsets` as radius_graph_pbc / get_pbc_distances produce upstream).  Auxiliary IS2RS head, attention head and
atom-edge attributes are not used by the `l1_256_nonlinear` config and are rejected.
"""
import torch

from ..graph import EdgeGraph
from ..irreps import Irreps
from .graph_attention_transformer import _Trunk
from .layers import NodeEmbeddingNetwork
from .registry import register_model

_MAX_ATOM_TYPE = 84
_NUM_TAGGED = 3
# the later assignment wins in the reference (graph_attention_transformer_oc20.py:62-66)
_AVG_NUM_NODES = 77.81317
_AVG_DEGREE = 23.395238876342773


class GraphAttentionTransformerOC20(_Trunk):
    def __init__(self, num_atoms=None, bond_feat_dim=None, num_targets=1, irreps_node_embedding="256x0e+128x1e",
                 num_layers=6, irreps_node_attr="1x0e", use_node_attr=False, irreps_sh="1x0e+1x1e", max_radius=6.0,
                 number_of_basis=128, fc_neurons=[64, 64], use_atom_edge_attr=False, irreps_atom_edge_attr="8x0e",
                 irreps_feature="512x0e", irreps_head="32x0e+16x1e", num_heads=8, irreps_pre_attn=None,
                 rescale_degree=False, nonlinear_message=False, irreps_mlp_mid="768x0e+384x1e", norm_layer="layer",
                 alpha_drop=0.2, proj_drop=0.0, out_drop=0.0, drop_path_rate=0.0, use_auxiliary_task=False,
                 auxiliary_head_dropout=True, use_attention_head=False, otf_graph=False, use_pbc=True,
                 max_neighbors=50):
        super().__init__()
        if use_node_attr or use_atom_edge_attr or use_auxiliary_task or use_attention_head:
            raise NotImplementedError("only the plain IS2RE energy head is on the MI355X hot path")
        self.otf_graph, self.use_pbc, self.max_neighbors = otf_graph, use_pbc, max_neighbors
        self.basis_type = "gaussian"
        self._build_trunk(irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius,
                          number_of_basis, fc_neurons, irreps_feature, irreps_head, num_heads, irreps_pre_attn,
                          rescale_degree, nonlinear_message, irreps_mlp_mid, norm_layer, alpha_drop, proj_drop,
                          out_drop, drop_path_rate, _MAX_ATOM_TYPE, _AVG_DEGREE, _AVG_NUM_NODES)
        self.tag_embed = NodeEmbeddingNetwork(self.irreps_node_embedding, _NUM_TAGGED)

    def forward(self, data):
        pos = data.pos.to(torch.float32).contiguous()
        batch = data.batch
        offsets = None
        edge_index = getattr(data, "edge_index", None)
        if edge_index is not None:
            graph, order = EdgeGraph.from_edges(edge_index[0], edge_index[1], pos.shape[0], batch)
            off = getattr(data, "offsets", None)
            if off is not None:
                offsets = off.to(torch.float32)[order].contiguous()
        else:
            # otf_graph [ref: _forward_otf_graph / _forward_use_pbc, :267-302]
            if self.use_pbc:
                cell = getattr(data, "cell", None)
                if cell is None:
                    raise ValueError("use_pbc=True needs data.cell ([B,3,3]) or precomputed data.edge_index / offsets")
                graph, offsets, _ = EdgeGraph.from_radius_pbc(pos, cell, batch, self.max_radius,
                                                              max_num_neighbors=self.max_neighbors)
            else:
                graph = EdgeGraph.from_radius(pos, batch, self.max_radius, max_num_neighbors=self.max_neighbors)
        atom_embedding, _, _ = self.atom_embed(data.atomic_numbers.long())
        tag_embedding, _, _ = self.tag_embed(data.tags.long())
        return self._trunk_forward(atom_embedding + tag_embedding, pos, graph, offsets)

    @property
    def num_params(self):
        return sum(p.numel() for p in self.parameters())


@register_model
def oc20_graph_attention_transformer(**model_attributes):
    """ocpmodels-registry name "graph_attention_transformer"; kwargs = the YAML `model:` section."""
    model_attributes.pop("name", None)
    return GraphAttentionTransformerOC20(None, None, 1, **model_attributes)


@register_model
def oc20_l1_256_nonlinear(**over):
    """oc20/configs/is2re/all/graph_attention_transformer/l1_256_nonlinear_g@2_local.yml"""
    cfg = dict(irreps_node_embedding="256x0e+128x1e", num_layers=6, irreps_node_attr="1x0e", use_node_attr=False,
               irreps_sh="1x0e+1x1e", max_radius=5.0, number_of_basis=128, fc_neurons=[64, 64],
               use_atom_edge_attr=False, irreps_feature="512x0e", irreps_head="32x0e+16x1e", num_heads=8,
               irreps_pre_attn="256x0e+128x1e", rescale_degree=False, nonlinear_message=True,
               irreps_mlp_mid="768x0e+384x1e", norm_layer="layer", alpha_drop=0.2, proj_drop=0.0, out_drop=0.0,
               drop_path_rate=0.0, otf_graph=True, use_pbc=True, max_neighbors=500)
    cfg.update(over)
    return GraphAttentionTransformerOC20(None, None, 1, **cfg)
