"""ocpmodels 0.0.3 (@ d2aaaeb) stand-in: common.registry.registry, common.utils.{conditional_grad, get_pbc_distances,
radius_graph_pbc}, models.gemnet.layers.radial_basis.RadialBasis -- the call sites are
nets/graph_attention_transformer_oc20.py:46-51,73,267-293 and nets/graph_attention_transformer.py:26,785-787."""
