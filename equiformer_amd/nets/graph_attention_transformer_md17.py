"""MD17 variant: energy + forces (= -dE/dpos through the HIP backward kernels).

Drop-in for the reference's nets/graph_attention_transformer_md17.py (`GraphAttentionTransformerMD17` :127-327 and the
`*_md17` factories :330-519): `forward(node_atom, pos, batch) -> (energy [B,1], forces [N,3])`, works under an outer
`torch.no_grad()`, reads `task_mean/task_std`.

In training mode the forces are produced with `create_graph=True` (reference: nets/graph_attention_transformer_md17.py:
318-325), i.e. `loss.backward()` on a force loss is a second-order pass: every operator's backward is then itself a
differentiable HIP operator (`ops._*Bwd`, kernels `eqf_*_bwd2` in csrc/second.hip; multilinear operators reuse their
first-order kernels).  In eval mode (force evaluation, main_md17.py:444-452) the forces come from the plain
first-order HIP backward and carry no graph.
"""
import torch

from .. import ops
from ..graph import EdgeGraph
from ..irreps import Irreps
from .graph_attention_transformer import _Trunk
from .layers import ExpNormalSmearing, GaussianRadialBasisLayer, GraphAttention, RadialBasis
from .registry import register_model

_MAX_ATOM_TYPE = 64
# the reference reuses the QM9 statistics for MD17 (graph_attention_transformer_md17.py:45-48)
_AVG_NUM_NODES = 18.03065905448718
_AVG_DEGREE = 15.57930850982666


class GraphAttentionTransformerMD17(_Trunk):
    def __init__(self, irreps_in="64x0e", irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6,
                 irreps_node_attr="1x0e", irreps_sh="1x0e+1x1e+1x2e", max_radius=5.0, number_of_basis=128,
                 basis_type="gaussian", fc_neurons=[64, 64], irreps_feature="512x0e",
                 irreps_head="32x0e+16x1o+8x2e", num_heads=4, irreps_pre_attn=None, rescale_degree=False,
                 nonlinear_message=False, irreps_mlp_mid="128x0e+64x1e+32x2e", use_attn_head=False,
                 norm_layer="layer", alpha_drop=0.2, proj_drop=0.0, out_drop=0.0, drop_path_rate=0.0, mean=None,
                 std=None, scale=None, atomref=None):
        super().__init__()
        self.use_attn_head = use_attn_head
        self.task_mean, self.task_std, self.scale = mean, std, scale
        self.register_buffer("atomref", atomref)
        self.irreps_node_input = Irreps(irreps_in)
        self.basis_type = basis_type
        self._build_trunk(irreps_node_embedding, num_layers, irreps_node_attr, irreps_sh, max_radius,
                          number_of_basis, fc_neurons, irreps_feature, irreps_head, num_heads, irreps_pre_attn,
                          rescale_degree, nonlinear_message, irreps_mlp_mid, norm_layer, alpha_drop, proj_drop,
                          out_drop, drop_path_rate, _MAX_ATOM_TYPE, _AVG_DEGREE, _AVG_NUM_NODES)
        if use_attn_head:  # one more attention layer reads the energy out of the equivariant feature [ref: :196-207]
            self.head = GraphAttention(self.irreps_feature, self.irreps_node_attr, self.irreps_edge_attr,
                                       Irreps("1x0e"), self.fc_neurons, self.irreps_head, num_heads, irreps_pre_attn,
                                       rescale_degree, nonlinear_message, alpha_drop, proj_drop)
            self.apply(self._init_weights)

    def _attention_heads(self):
        return [self.head] if self.use_attn_head else []

    def _make_rbf(self):
        if self.basis_type == "gaussian":
            self.rbf = GaussianRadialBasisLayer(self.number_of_basis, cutoff=self.max_radius)
        elif self.basis_type == "exp":
            self.rbf = ExpNormalSmearing(cutoff_lower=0.0, cutoff_upper=self.max_radius, num_rbf=self.number_of_basis,
                                         trainable=False)
        elif self.basis_type == "bessel":
            self.rbf = RadialBasis(self.number_of_basis, cutoff=self.max_radius, rbf={"name": "spherical_bessel"})
        else:
            raise ValueError

    # The reference builds the forces with create_graph=True in every mode (graph_attention_transformer_md17.py:318-325).
    # Here the differentiable (second-order) force pass runs in training mode only -- evaluation gets the plain first-order
    # kernels, 3-4x cheaper -- unless this switch asks for the reference's behaviour in eval mode as well (forces that can be
    # differentiated again, e.g. for a force-loss gradient on a validation batch).
    differentiable_forces_in_eval = False

    @torch.enable_grad()
    def forward(self, node_atom, pos, batch, graph=None):
        """graph: (extension of the reference signature) the radius graph of `pos`, built by the caller -- what a train step
        captured in a HIP graph passes (equiformer_amd/capture.py: the graph's edge count is read back on the host, outside the
        capture)."""
        pos = pos.to(torch.float32).contiguous().requires_grad_(True)
        if graph is None:
            graph = EdgeGraph.from_radius(pos, batch, self.max_radius, max_num_neighbors=1000)
        atom_embedding, _, _ = self.atom_embed(node_atom)
        trainable = any(p.requires_grad for p in self.parameters())
        second_order = (self.training or self.differentiable_forces_in_eval) and trainable
        self.__dict__["_second_order_pass"] = second_order  # read by _trunk_forward (radial bank: first-order only)
        energy = self._trunk_forward(atom_embedding, pos, graph)
        if self.scale is not None:
            energy = self.scale * energy
        with ops.input_grads_only():  # only d E / d pos is wanted here: no parameter gradients in this pass
            forces = -1 * torch.autograd.grad(energy, pos, grad_outputs=torch.ones_like(energy),
                                              create_graph=second_order, retain_graph=trainable)[0]
        return energy, forces


def _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, **over):
    kw = dict(irreps_in=irreps_in, irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_node_attr="1x0e",
              irreps_sh="1x0e+1x1e+1x2e", max_radius=radius, number_of_basis=num_basis, fc_neurons=[64, 64],
              basis_type="exp", irreps_feature="512x0e", irreps_head="32x0e+16x1e+8x2e", num_heads=4,
              irreps_pre_attn=None, rescale_degree=False, nonlinear_message=True,
              irreps_mlp_mid="384x0e+192x1e+96x2e", norm_layer="layer", alpha_drop=0.0, proj_drop=0.0, out_drop=0.0,
              drop_path_rate=0.0, mean=task_mean, std=task_std, scale=None, atomref=atomref)
    cls = over.pop("cls", GraphAttentionTransformerMD17)
    kw.update(over)
    return cls(**kw)


@register_model
def graph_attention_transformer_l2_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None, task_std=None,
                                        **kwargs):
    """[ref: nets/graph_attention_transformer_md17.py:330-347] linear messages, Gaussian basis, alpha_drop 0.2"""
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, basis_type="gaussian", alpha_drop=0.2,
                 nonlinear_message=False)


@register_model
def graph_attention_transformer_nonlinear_bessel_l2_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                         task_std=None, **kwargs):
    """[ref: nets/graph_attention_transformer_md17.py:387-404]"""
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, basis_type="bessel")


@register_model
def graph_attention_transformer_nonlinear_bessel_l3_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                         task_std=None, **kwargs):
    """[ref: nets/graph_attention_transformer_md17.py:484-501]"""
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, basis_type="bessel",
                 irreps_node_embedding="128x0e+64x1e+64x2e+32x3e", irreps_sh="1x0e+1x1e+1x2e+1x3e",
                 irreps_head="32x0e+16x1e+16x2e+8x3e", irreps_mlp_mid="384x0e+192x1e+192x2e+96x3e")


@register_model
def graph_attention_transformer_nonlinear_l2_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                  task_std=None, **kwargs):
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, basis_type="gaussian", alpha_drop=0.2)


@register_model
def graph_attention_transformer_nonlinear_exp_l2_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                      task_std=None, **kwargs):
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref)


@register_model
def graph_attention_transformer_nonlinear_exp_l3_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                      task_std=None, **kwargs):
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref,
                 irreps_node_embedding="128x0e+64x1e+64x2e+32x3e", irreps_sh="1x0e+1x1e+1x2e+1x3e",
                 irreps_head="32x0e+16x1e+16x2e+8x3e", irreps_mlp_mid="384x0e+192x1e+192x2e+96x3e")


@register_model
def graph_attention_transformer_nonlinear_attn_exp_l3_md17(irreps_in, radius, num_basis=128, atomref=None,
                                                           task_mean=None, task_std=None, **kwargs):
    """[ref: nets/graph_attention_transformer_md17.py:445-462] L_max = 3 feature, GraphAttention energy head"""
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref,
                 irreps_node_embedding="128x0e+64x1e+64x2e+32x3e", irreps_sh="1x0e+1x1e+1x2e+1x3e",
                 irreps_feature="128x0e+64x1e+64x2e+32x3e", irreps_head="32x0e+16x1e+16x2e+8x3e",
                 irreps_mlp_mid="384x0e+192x1e+192x2e+96x3e", use_attn_head=True)


_E3_L2 = dict(irreps_node_embedding="128x0e+32x0o+32x1e+32x1o+16x2e+16x2o", irreps_sh="1x0e+1x1o+1x2e",
              irreps_head="32x0e+8x0o+8x1e+8x1o+4x2e+4x2o", irreps_mlp_mid="384x0e+96x0o+96x1e+96x1o+48x2e+48x2o")
_E3_L3 = dict(irreps_node_embedding="128x0e+64x0o+32x1e+32x1o+32x2e+32x2o+16x3e+16x3o", irreps_sh="1x0e+1x1o+1x2e+1x3o",
              irreps_head="32x0e+16x0o+8x1e+8x1o+8x2e+8x2o+4x3e+4x3o",
              irreps_mlp_mid="384x0e+192x0o+96x1e+96x1o+96x2e+96x2o+48x3e+48x3o")


@register_model
def graph_attention_transformer_nonlinear_l2_e3_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                     task_std=None, **kwargs):
    """[ref: nets/graph_attention_transformer_md17.py:368-385] E(3) irreps (parity-aware), Gaussian basis, alpha_drop 0.2"""
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, basis_type="gaussian", alpha_drop=0.2,
                 **_E3_L2)


@register_model
def graph_attention_transformer_nonlinear_exp_l3_e3_md17(irreps_in, radius, num_basis=128, atomref=None, task_mean=None,
                                                         task_std=None, **kwargs):
    """[ref: nets/graph_attention_transformer_md17.py:465-482]"""
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, **_E3_L3)


@register_model
def graph_attention_transformer_nonlinear_bessel_l3_e3_md17(irreps_in, radius, num_basis=128, atomref=None,
                                                            task_mean=None, task_std=None, **kwargs):
    """[ref: nets/graph_attention_transformer_md17.py:503-519]"""
    return _md17(irreps_in, radius, num_basis, task_mean, task_std, atomref, basis_type="bessel", **_E3_L3)
