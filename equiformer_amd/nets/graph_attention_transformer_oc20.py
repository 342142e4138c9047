"""OC20 IS2RE variant (energy head), drop-in for nets/graph_attention_transformer_oc20.py:73-386 of the reference.

The reference registers the class in ocpmodels' registry as "graph_attention_transformer" and receives an ocpmodels
`Batch`; ocpmodels is un-vendored, so the class is exposed here under the same name through this package's registry
(`oc20_graph_attention_transformer`) and accepts any object with the same attributes
(`pos, batch, atomic_numbers, tags, natoms, cell`, and with `otf_graph=False` the precomputed `edge_index, cell_offsets,
neighbors` of an ocpmodels batch -- see `_graph` for the exact contract; with `otf_graph=True` the periodic neighbour search
runs on the GPU, `EdgeGraph.from_radius_pbc`).  The auxiliary IS2RS head
(`use_auxiliary_task`, the `*_aux_*` configs), the attention head (`use_attention_head`) and per-graph stochastic depth
(`drop_path_rate`) are built as upstream; atom-edge attributes and node attributes are used by no shipped config and are
rejected.
"""
import torch

from ..graph import EdgeGraph
from ..irreps import Irreps
from .graph_attention_transformer import _RESCALE, _Trunk
from .layers import Activation, GraphAttention, LinearRS, NodeEmbeddingNetwork
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
        if use_node_attr or use_atom_edge_attr:
            raise NotImplementedError("node attributes / atom-edge attributes are used by no shipped OC20 config")
        self.otf_graph, self.use_pbc, self.max_neighbors = otf_graph, use_pbc, max_neighbors
        self.basis_type = "gaussian"
        self._build_trunk(irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius,
                          number_of_basis, fc_neurons, irreps_feature, irreps_head, num_heads, irreps_pre_attn,
                          rescale_degree, nonlinear_message, irreps_mlp_mid, norm_layer, alpha_drop, proj_drop,
                          out_drop, drop_path_rate, _MAX_ATOM_TYPE, _AVG_DEGREE, _AVG_NUM_NODES)
        self.tag_embed = NodeEmbeddingNetwork(self.irreps_node_embedding, _NUM_TAGGED)
        # energy head on the scalar channels of the feature [ref: :169-179]
        scalars = Irreps([(m, ir) for m, ir in self.irreps_feature if ir.l == 0 and ir.p == 1])
        self.head = torch.nn.Sequential(LinearRS(self.irreps_feature, scalars, rescale=_RESCALE),
                                        Activation(scalars, kind="silu"),
                                        LinearRS(scalars, Irreps("1x0e")))
        self.use_auxiliary_task, self.use_attention_head = use_auxiliary_task, use_attention_head
        # [ref: :185-187] 1o when the feature carries 1o channels (E(3) variants), 1e otherwise
        irreps_aux = Irreps("1x1o") if any(ir.l == 1 and ir.p == -1 for _, ir in self.irreps_feature) else Irreps("1x1e")
        head_drop = alpha_drop if auxiliary_head_dropout else 0.0

        def attention(irreps_out):
            return self._head_attention(irreps_out, num_heads, irreps_pre_attn, rescale_degree, nonlinear_message,
                                        head_drop)
        if use_auxiliary_task and not use_attention_head:  # IS2RS auxiliary head [ref: :182-194]
            self.auxiliary_head = attention(irreps_aux)
        if use_attention_head:  # GraphAttention for energy (and auxiliary vectors) + linear skip [ref: :196-208]
            irreps_out = Irreps("1x0e") + irreps_aux if use_auxiliary_task else Irreps("1x0e")
            self.head = attention(irreps_out)
            self.head_skip_connect = LinearRS(self.irreps_feature, irreps_out)
        self.apply(self._init_weights)
        # registration order of the reference (tag_embed right after atom_embed, :146-147): parameters() order is what
        # optimizer checkpoints index by (tests/test_reference_pin.py)
        mods = dict(self._modules)
        tag = mods.pop("tag_embed")
        self._modules = {k2: v2 for k, v in mods.items() for k2, v2 in (((k, v), ("tag_embed", tag)) if k == "atom_embed"
                                                                         else ((k, v),))}

    def _head_attention(self, irreps_out, num_heads, irreps_pre_attn, rescale_degree, nonlinear_message, alpha_drop):
        return GraphAttention(self.irreps_feature, self.irreps_node_attr, self.irreps_edge_attr, irreps_out,
                              self.fc_neurons, self.irreps_head, num_heads, irreps_pre_attn, rescale_degree,
                              nonlinear_message, alpha_drop=alpha_drop, proj_drop=0.0)

    def _attention_heads(self):
        if self.use_attention_head:
            return [self.head]
        return [self.auxiliary_head] if self.use_auxiliary_task else []

    def _graph(self, data, pos, batch):
        """The reference's input contract [ref: _forward_otf_graph :267-277, _forward_use_pbc :280-302]:

        * otf_graph=True: the graph is REBUILT from positions (and data.cell when use_pbc) even if the batch carries a
          precomputed edge_index -- on the GPU here (EdgeGraph.from_radius_pbc = radius_graph_pbc + get_pbc_distances);
        * otf_graph=False, use_pbc=True: the ocpmodels batch's data.edge_index [2,E] (neighbour, centre) +
          data.cell_offsets [E,3] (integer images) + data.neighbors [B] (edges per structure) + data.cell [B,3,3];
          Cartesian offsets = cell_offsets @ cell[structure of the edge], zero-length edges dropped
          (get_pbc_distances(return_offsets=True));
        * use_pbc=False: plain radius graph from positions (a precomputed edge_index is ignored, as upstream).

        Extension (not in the reference): with otf_graph=False and no data.cell_offsets, per-edge Cartesian
        `data.offsets` [E,3] next to data.edge_index are taken as they are."""
        if not self.use_pbc:
            return EdgeGraph.from_radius(pos, batch, self.max_radius, max_num_neighbors=self.max_neighbors), None
        if self.otf_graph:
            cell = getattr(data, "cell", None)
            if cell is None:
                raise ValueError("otf_graph=True with use_pbc=True needs data.cell ([B,3,3])")
            graph, offsets, _ = EdgeGraph.from_radius_pbc(pos, cell, batch, self.max_radius,
                                                          max_num_neighbors=self.max_neighbors)
            return graph, offsets
        edge_index = getattr(data, "edge_index", None)
        cell_offsets = getattr(data, "cell_offsets", None)
        if edge_index is None:
            raise ValueError("otf_graph=False with use_pbc=True needs data.edge_index + data.cell_offsets + data.neighbors "
                             "+ data.cell (the ocpmodels batch), or otf_graph=True")
        edge_index = edge_index.to(pos.device)
        if cell_offsets is not None:
            cell, neighbors = getattr(data, "cell", None), getattr(data, "neighbors", None)
            if cell is None or neighbors is None:
                raise ValueError("data.cell_offsets needs data.cell ([B,3,3]) and data.neighbors ([B])")
            cell = cell.to(device=pos.device, dtype=torch.float32).view(-1, 3, 3)
            per_edge = torch.repeat_interleave(cell, neighbors.to(pos.device).long(), dim=0)
            if per_edge.shape[0] != edge_index.shape[1]:
                raise ValueError("data.neighbors sums to %d, data.edge_index has %d edges"
                                 % (per_edge.shape[0], edge_index.shape[1]))
            offsets = torch.bmm(cell_offsets.to(device=pos.device, dtype=torch.float32).view(-1, 1, 3), per_edge).view(-1, 3)
            vec = pos.detach()[edge_index[0]] - pos.detach()[edge_index[1]] + offsets
            keep = (vec != 0).any(dim=1)  # get_pbc_distances drops zero-length edges
            if not bool(keep.all()):
                edge_index, offsets = edge_index[:, keep], offsets[keep]
        else:
            offsets = getattr(data, "offsets", None)
            if offsets is None:
                raise ValueError("use_pbc=True: data.edge_index without data.cell_offsets (ocpmodels) or data.offsets")
            offsets = offsets.to(device=pos.device, dtype=torch.float32)
        graph, order = EdgeGraph.from_edges(edge_index[0], edge_index[1], pos.shape[0], batch)
        return graph, offsets[order].contiguous()

    def forward(self, data):
        pos = data.pos.to(torch.float32).contiguous()
        batch = data.batch
        graph, offsets = self._graph(data, pos, batch)
        atom_embedding, _, _ = self.atom_embed(data.atomic_numbers.long())
        tag_embedding, _, _ = self.tag_embed(data.tags.long())
        node_features, ectx = self._trunk_features(atom_embedding + tag_embedding, pos, graph, offsets)
        scatter = lambda t: self.scale_scatter(t, graph.mol_ptr, graph.batch, graph.num_graphs)  # noqa: E731
        if self.use_attention_head:  # [ref: :352-366]
            outputs = self.head(node_features, ectx=ectx) + self.head_skip_connect(node_features)
            if self.use_auxiliary_task:
                return scatter(outputs.narrow(1, 0, 1).contiguous()), outputs.narrow(1, 1, 3)
            return scatter(outputs)
        energy = scatter(self.head(node_features))
        if self.use_auxiliary_task:  # [ref: :372-379]
            return energy, self.auxiliary_head(node_features, ectx=ectx)
        return energy

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


@register_model
def oc20_l1_256_nonlinear_aux(**over):
    """oc20/configs/is2re/all/graph_attention_transformer/l1_256_nonlinear_aux_g@2_local.yml: feature with l=1 channels,
    IS2RS auxiliary head, stochastic depth 0.05.  Returns (energy [B,1], per-node vectors [N,3])."""
    cfg = dict(irreps_feature="512x0e+256x1e", drop_path_rate=0.05, use_auxiliary_task=True)
    cfg.update(over)
    return oc20_l1_256_nonlinear(**cfg)


@register_model
def oc20_l1_256_blocks18_nonlinear_aux(**over):
    """oc20/configs/is2re/all/graph_attention_transformer/l1_256_blocks@18_nonlinear_aux_g@4_local.yml"""
    cfg = dict(num_layers=18)
    cfg.update(over)
    return oc20_l1_256_nonlinear_aux(**cfg)


@register_model
def oc20_l1_256(**over):
    """oc20/configs/is2re/all/graph_attention_transformer/l1_256_g@2_local.yml (linear messages)"""
    cfg = dict(nonlinear_message=False, num_layers=8)
    cfg.update(over)
    return oc20_l1_256_nonlinear(**cfg)


@register_model
def oc20_l1_256_e3_nonlinear(**over):
    """oc20/configs/is2re/all/graph_attention_transformer/l1_256_e3_nonlinear_g@2_local.yml (E(3) irreps)"""
    cfg = dict(irreps_node_embedding="256x0e+64x0o+64x1e+64x1o", irreps_sh="1x0e+1x1o",
               irreps_head="32x0e+8x0o+8x1e+8x1o", irreps_pre_attn="256x0e+64x0o+64x1e+64x1o",
               irreps_mlp_mid="768x0e+192x0o+192x1e+192x1o")
    cfg.update(over)
    return oc20_l1_256_nonlinear(**cfg)
