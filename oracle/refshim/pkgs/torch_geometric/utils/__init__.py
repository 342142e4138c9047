import torch

from torch_scatter import scatter


def softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
    """torch_geometric/utils/softmax.py (2.0.3): out = exp(src - max_seg); out / (sum_seg(out) + 1e-16)."""
    assert ptr is None and index is not None
    N = int(index.max()) + 1 if num_nodes is None else num_nodes
    src_max = scatter(src, index, dim, dim_size=N, reduce="max").index_select(dim, index)
    out = (src - src_max).exp()
    out_sum = scatter(out, index, dim, dim_size=N, reduce="sum").index_select(dim, index)
    return out / (out_sum + 1e-16)


def degree(index, num_nodes=None, dtype=None):
    N = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((N,), dtype=dtype, device=index.device)
    return out.scatter_add_(0, index, torch.ones((index.size(0),), dtype=out.dtype, device=out.device))
