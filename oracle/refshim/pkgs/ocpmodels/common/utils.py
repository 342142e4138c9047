from functools import wraps

import torch

from oracle import pbc as _pbc


def conditional_grad(dec):
    """ocpmodels.common.utils.conditional_grad: apply `dec` (torch.enable_grad()) when self.regress_forces is set."""
    def decorator(func):
        @wraps(func)
        def cls_method(self, *args, **kwargs):
            f = func
            if getattr(self, "regress_forces", False) and not getattr(self, "direct_forces", 0):
                f = dec(func)
            return f(self, *args, **kwargs)
        return cls_method
    return decorator


def get_pbc_distances(pos, edge_index, cell, cell_offsets, neighbors, return_offsets=False, return_distance_vec=False):
    ei, dist, offsets = _pbc.get_pbc_distances(pos, edge_index, cell, cell_offsets, neighbors)
    out = {"edge_index": ei, "distances": dist}
    if return_offsets:
        out["offsets"] = offsets
    if return_distance_vec:
        out["distance_vec"] = pos[ei[0]] - pos[ei[1]] + offsets
    return out


def radius_graph_pbc(data, radius, max_num_neighbors_threshold):
    return _pbc.radius_graph_pbc(data.pos.detach(), data.cell, data.natoms, radius, max_num_neighbors_threshold)
