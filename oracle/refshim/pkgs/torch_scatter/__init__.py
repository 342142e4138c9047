"""torch_scatter 2.0.9 stand-in: scatter(src, index, dim, dim_size, reduce) as the reference calls it
(nets/graph_attention_transformer.py:513,700: reduce='sum' / 'mean' over dim 0)."""
import torch


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    assert out is None
    dim = dim % src.dim()
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    idx = index.view([-1 if d == dim else 1 for d in range(src.dim())]).expand_as(src)
    if reduce in ("sum", "add"):
        return src.new_zeros(shape).scatter_add(dim, idx, src)
    if reduce == "mean":
        s = src.new_zeros(shape).scatter_add(dim, idx, src)
        cnt = src.new_zeros(shape).scatter_add(dim, idx, torch.ones_like(src)).clamp_(min=1)
        return s / cnt
    if reduce == "max":
        return src.new_full(shape, float("-inf")).scatter_reduce(dim, idx, src, reduce="amax", include_self=True)
    raise NotImplementedError(reduce)
