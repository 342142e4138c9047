"""MD17 model with the DeNS auxiliary task (denoising non-equilibrium structures), drop-in for the reference's
nets/equiformer_md17_dens.py (`Equiformer_MD17_DeNS` :55-354, factory `equiformer_md17_dens` :358-359).

`forward(data) -> (energy [B,1], forces-or-denoising vectors [N,3])` with `data.z, data.pos, data.batch` and, for
corrupted structures, `data.force` (the forces of the uncorrupted structure, encoded into the input embedding as
|F|/sqrt(3) * Y(F/|F|) on the corrupted atoms), `data.noise_mask` (atoms whose position was perturbed: their output row
is the prediction of the denoising head instead of the force) and, with `use_force_encoding=False`,
`data.denoising_mask` / `data.denoising_pos_mask`.  The trunk, the second-order force path and the attention head are
the ones of the other MD17 models; the force encoding is eqf_vec_sh (no gradient: it is input data).
"""
import math

import torch

from .. import ops
from ..graph import EdgeGraph
from ..irreps import Irreps
from .graph_attention_transformer import _RESCALE
from .graph_attention_transformer_md17 import _AVG_DEGREE, _AVG_NUM_NODES, _MAX_ATOM_TYPE, GraphAttentionTransformerMD17
from .layers import Activation, GraphAttention, LinearRS
from .registry import register_model


class Equiformer_MD17_DeNS(GraphAttentionTransformerMD17):
    def __init__(self, irreps_in="64x0e", irreps_equivariant_inputs="1x0e+1x1e+1x2e",
                 irreps_node_embedding="128x0e+64x1e+32x2e", num_layers=6, irreps_node_attr="1x0e",
                 irreps_sh="1x0e+1x1e+1x2e", max_radius=5.0, number_of_basis=32, basis_type="exp", fc_neurons=[64, 64],
                 irreps_feature="512x0e+256x1e+128x2e", irreps_head="32x0e+16x1o+8x2e", num_heads=4,
                 irreps_pre_attn="128x0e+64x1e+32x2e", rescale_degree=False, nonlinear_message=True,
                 irreps_mlp_mid="128x0e+64x1e+32x2e", norm_layer="layer", alpha_drop=0.0, proj_drop=0.0, out_drop=0.0,
                 drop_path_rate=0.0, mean=None, std=None, scale=None, atomref=None, use_force_encoding=True):
        super().__init__(irreps_in=irreps_in, irreps_node_embedding=irreps_node_embedding, num_layers=num_layers,
                         irreps_node_attr=irreps_node_attr, irreps_sh=irreps_sh, max_radius=max_radius,
                         number_of_basis=number_of_basis, basis_type=basis_type, fc_neurons=fc_neurons,
                         irreps_feature=irreps_feature, irreps_head=irreps_head, num_heads=num_heads,
                         irreps_pre_attn=irreps_pre_attn, rescale_degree=rescale_degree,
                         nonlinear_message=nonlinear_message, irreps_mlp_mid=irreps_mlp_mid, norm_layer=norm_layer,
                         alpha_drop=alpha_drop, proj_drop=proj_drop, out_drop=out_drop, drop_path_rate=drop_path_rate,
                         mean=mean, std=std, scale=scale, atomref=atomref)
        self.use_force_encoding = use_force_encoding
        self.irreps_node_equivariant_inputs = Irreps(irreps_equivariant_inputs)
        ls = [(m, ir.l, ir.p) for m, ir in self.irreps_node_equivariant_inputs]
        if ls != [(1, l, 1) for l in range(len(ls))] or len(ls) > 4:
            raise NotImplementedError("irreps_equivariant_inputs must be 1x0e+1x1e+... (<= 3): %r" % (irreps_equivariant_inputs,))
        self._lmax_force = len(ls) - 1
        if hasattr(self, "head"):  # scalar-only feature: the trunk built the plain MD17 head, which this model lacks
            del self.head
        self.force_embed = LinearRS(self.irreps_node_equivariant_inputs, self.irreps_node_embedding, rescale=_RESCALE)
        scalars = Irreps([(m, ir) for m, ir in self.irreps_feature if ir.l == 0 and ir.p == 1])
        self.energy_head = torch.nn.Sequential(LinearRS(self.irreps_feature, scalars, rescale=_RESCALE),
                                               Activation(scalars, kind="silu"),
                                               LinearRS(scalars, Irreps("1x0e"), rescale=_RESCALE))
        out = Irreps("1x1e")  # the equivariant inputs carry 1e [ref: :153-154]
        self.denoising_pos_head = GraphAttention(self.irreps_feature, self.irreps_node_attr, self.irreps_edge_attr, out,
                                                 self.fc_neurons, self.irreps_head, num_heads, irreps_pre_attn,
                                                 rescale_degree, nonlinear_message, alpha_drop, proj_drop)
        self.apply(self._init_weights)
        # registration order of the reference (parameters() order is what optimizer checkpoints index by)
        order = ["atom_embed", "rbf", "edge_deg_embed", "force_embed", "blocks", "norm", "energy_head", "scale_scatter",
                 "denoising_pos_head"]
        assert sorted(order) == sorted(self._modules), sorted(self._modules)
        self._modules = {k: self._modules[k] for k in order}

    def _attention_heads(self):
        return [self.denoising_pos_head]

    @torch.enable_grad()
    def forward(self, data):
        node_atom, batch = data.z, data.batch
        pos = data.pos.to(torch.float32).contiguous().requires_grad_(True)
        graph = EdgeGraph.from_radius(pos, batch, self.max_radius, max_num_neighbors=1000)
        atom_embedding, _, _ = self.atom_embed(node_atom)
        n = pos.shape[0]
        if hasattr(data, "force") and self.use_force_encoding:  # [ref: :276-289]
            force_sh = ops.vec_sh(data.force, data.noise_mask, self._lmax_force, 1.0 / math.sqrt(3.0))
        else:
            force_sh = torch.zeros((n, self.irreps_node_equivariant_inputs.dim), device=pos.device, dtype=torch.float32)
        force_embedding = self.force_embed(force_sh)
        trainable = any(p.requires_grad for p in self.parameters())
        second_order = self.training and trainable
        self.__dict__["_second_order_pass"] = second_order
        node_features, ectx = self._trunk_features(atom_embedding, pos, graph, extra=force_embedding)
        energy = self.energy_head(node_features)
        if hasattr(data, "denoising_mask") and not self.use_force_encoding:  # [ref: :309-311]
            energy = energy * (~data.denoising_mask).to(energy.dtype).view(-1, 1)
        energy = self.scale_scatter(energy, graph.mol_ptr, graph.batch, graph.num_graphs)
        if self.scale is not None:
            energy = self.scale * energy
        denoise = hasattr(data, "noise_mask")
        with ops.input_grads_only():
            forces = -1 * torch.autograd.grad(energy, pos, grad_outputs=torch.ones_like(energy),
                                              create_graph=second_order, retain_graph=trainable or denoise)[0]
        if not denoise:
            return energy, forces
        denoising_pos = self.denoising_pos_head(node_features, ectx=ectx)  # [ref: :328-345]
        outputs_dy = torch.where(data.noise_mask.view(-1, 1), denoising_pos, forces)
        if not self.use_force_encoding:
            outputs_dy = outputs_dy * (~data.denoising_pos_mask).to(outputs_dy.dtype).view(-1, 1)
        return energy, outputs_dy


@register_model
def equiformer_md17_dens(**kwargs):
    return Equiformer_MD17_DeNS(**kwargs)


@register_model
def equiformer_md17_dens_l2(**over):
    """md17/configs/equiformer_dens/equiformer_dens_N@6_L@2_C@128-64-32.yml"""
    cfg = dict(irreps_in="64x0e", irreps_equivariant_inputs="1x0e+1x1e+1x2e", irreps_node_embedding="128x0e+64x1e+32x2e",
               num_layers=6, irreps_node_attr="1x0e", irreps_sh="1x0e+1x1e+1x2e", max_radius=5.0, number_of_basis=32,
               basis_type="exp", fc_neurons=[64, 64], irreps_feature="512x0e+256x1e+128x2e",
               irreps_head="32x0e+16x1e+8x2e", num_heads=4, irreps_pre_attn="128x0e+64x1e+32x2e", rescale_degree=False,
               nonlinear_message=True, irreps_mlp_mid="384x0e+192x1e+96x2e", norm_layer="layer", alpha_drop=0.0,
               proj_drop=0.0, out_drop=0.0, drop_path_rate=0.0, use_force_encoding=True)
    cfg.update(over)
    return Equiformer_MD17_DeNS(**cfg)


@register_model
def equiformer_md17_dens_l3(**over):
    """md17/configs/equiformer_dens/equiformer_dens_N@6_L@3_C@128-64-64-32.yml"""
    cfg = dict(irreps_equivariant_inputs="1x0e+1x1e+1x2e+1x3e", irreps_node_embedding="128x0e+64x1e+64x2e+32x3e",
               irreps_sh="1x0e+1x1e+1x2e+1x3e", irreps_feature="512x0e+256x1e+256x2e+128x3e",
               irreps_head="32x0e+16x1e+16x2e+8x3e", irreps_pre_attn="128x0e+64x1e+64x2e+32x3e",
               irreps_mlp_mid="384x0e+192x1e+192x2e+96x3e")
    cfg.update(over)
    return equiformer_md17_dens_l2(**cfg)
