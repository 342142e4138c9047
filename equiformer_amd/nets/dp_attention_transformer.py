"""Dot-product-attention ablation family, drop-in for the reference's nets/dp_attention_transformer.py (QM9,
`DotProductAttentionTransformer` :255-411), nets/dp_attention_transformer_md17.py (`DotProductAttentionTransformerMD17`
:57-235, energy + forces) and nets/dp_attention_transformer_oc20.py (`DotProductAttentionTransformerOC20` :36-347,
ocpmodels-registry name "dp_attention_transformer").

Upstream these are copies of the graph-attention models with `DPTransBlock` in place of `TransBlock` (and, on OC20, a
`DotProductAttention` auxiliary head); here they are the same trunks with the block class swapped, so the embeddings,
the feed-forward networks, the fused SeparableFCTP kernels, the radial bank, stochastic depth and the second-order
force path are shared.  Parameter names follow the reference (`blocks.N.dpa.{query,merge_src,merge_dst,key_value,proj}`).
"""
from .graph_attention_transformer import GraphAttentionTransformer, _l2_kwargs
from .graph_attention_transformer_md17 import GraphAttentionTransformerMD17, _md17
from .graph_attention_transformer_oc20 import GraphAttentionTransformerOC20
from .layers import DotProductAttention, DPTransBlock
from .registry import register_model


class DotProductAttentionTransformer(GraphAttentionTransformer):
    _block_cls = DPTransBlock


class DotProductAttentionTransformerMD17(GraphAttentionTransformerMD17):
    _block_cls = DPTransBlock

    def __init__(self, *args, **kwargs):
        if kwargs.get("use_attn_head"):
            raise NotImplementedError("the dot-product family has no attention energy head")
        super().__init__(*args, **kwargs)


class DotProductAttentionTransformerOC20(GraphAttentionTransformerOC20):
    _block_cls = DPTransBlock

    def __init__(self, *args, **kwargs):
        if kwargs.get("use_attention_head"):
            raise NotImplementedError("the dot-product family has no attention energy head")
        super().__init__(*args, **kwargs)

    def _head_attention(self, irreps_out, num_heads, irreps_pre_attn, rescale_degree, nonlinear_message, alpha_drop):
        # [ref: nets/dp_attention_transformer_oc20.py:146-151]
        return DotProductAttention(self.irreps_feature, self.irreps_node_attr, self.irreps_edge_attr, irreps_out,
                                   self.fc_neurons, self.irreps_head, num_heads, irreps_pre_attn, rescale_degree,
                                   alpha_drop, proj_drop=0.0)


@register_model
def dot_product_attention_transformer_l2(irreps_in, radius, num_basis=128, atomref=None, task_mean=None, task_std=None,
                                         **kwargs):
    """[ref: nets/dp_attention_transformer.py:414-431]"""
    return DotProductAttentionTransformer(**_l2_kwargs(irreps_in, radius, num_basis, task_mean, task_std, atomref,
                                                       nonlinear_message=False))


@register_model
def dot_product_attention_transformer_exp_l2_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                  task_std=None, **kwargs):
    """[ref: nets/dp_attention_transformer_md17.py:238-254]"""
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, nonlinear_message=False,
                 cls=DotProductAttentionTransformerMD17)


@register_model
def dot_product_attention_transformer_exp_l3_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                  task_std=None, **kwargs):
    """[ref: nets/dp_attention_transformer_md17.py:257-272]"""
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, nonlinear_message=False,
                 irreps_node_embedding="128x0e+64x1e+64x2e+32x3e", irreps_sh="1x0e+1x1e+1x2e+1x3e",
                 irreps_head="32x0e+16x1e+16x2e+8x3e", irreps_mlp_mid="384x0e+192x1e+192x2e+96x3e",
                 cls=DotProductAttentionTransformerMD17)


@register_model
def oc20_dp_attention_transformer(**model_attributes):
    """ocpmodels-registry name "dp_attention_transformer"; kwargs = the YAML `model:` section."""
    model_attributes.pop("name", None)
    return DotProductAttentionTransformerOC20(None, None, 1, **model_attributes)


@register_model
def oc20_dp_l1_256(**over):
    """oc20/configs/is2re/all/dp_attention_transformer/l1_256_g@2_local.yml"""
    cfg = dict(irreps_node_embedding="256x0e+128x1e", num_layers=8, irreps_node_attr="1x0e", use_node_attr=False,
               irreps_sh="1x0e+1x1e", max_radius=5.0, number_of_basis=128, fc_neurons=[64, 64],
               use_atom_edge_attr=False, irreps_feature="512x0e", irreps_head="32x0e+16x1e", num_heads=8,
               irreps_pre_attn="256x0e+128x1e", rescale_degree=False, nonlinear_message=False,
               irreps_mlp_mid="768x0e+384x1e", norm_layer="layer", alpha_drop=0.2, proj_drop=0.0, out_drop=0.0,
               drop_path_rate=0.0, otf_graph=True, use_pbc=True, max_neighbors=500)
    cfg.update(over)
    return DotProductAttentionTransformerOC20(None, None, 1, **cfg)
