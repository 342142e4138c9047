"""torch_cluster 1.6.0 stand-in: radius_graph(pos, r, batch, max_num_neighbors) -> [2, E] = (neighbour, centre)
(flow='source_to_target', loop=False, strict d < r; nets/graph_attention_transformer.py:866-867) -- oracle.nets.radius_graph."""
import torch

from oracle.nets import radius_graph as _rg


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target", num_workers=1):
    assert not loop and flow == "source_to_target"
    if batch is None:
        batch = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
    src, dst = _rg(x.detach(), r, batch, max_num_neighbors)
    return torch.stack([src, dst])
