"""Destination-sorted edge lists (CSR) for the segmented HIP kernels.

The reference builds `edge_index` with torch_cluster.radius_graph (nets/graph_attention_transformer.py:866-867) and
scatters with atomics; here the graph is produced already grouped by destination node (the order torch_cluster
emits as well), together with the CSR offsets and the by-source permutation that turn every scatter / gather
backward into an atomics-free segmented reduction.
"""
import ctypes

import torch

from . import ops
from .lib import call


def _i32(t):
    return t.to(torch.int32).contiguous()


def _ptr_from_counts(counts):
    z = torch.zeros(1, dtype=torch.int64, device=counts.device)
    return _i32(torch.cat([z, torch.cumsum(counts.to(torch.int64), 0)]))


class EdgeGraph:
    """N nodes, E directed edges sorted by dst.  All index tensors are int32 on the GPU."""

    def __init__(self, N, src, dst, row_ptr, batch=None, num_graphs=None):
        self.N = int(N)
        self.src, self.dst, self.row_ptr = src, dst, row_ptr
        self.E = int(src.shape[0])
        # by-source view: edges grouped by src (for gradients that flow back to the source node)
        order = torch.argsort(src.to(torch.int64), stable=True)
        self.src_perm = _i32(order)
        self.src_ptr = _ptr_from_counts(torch.bincount(src.to(torch.int64), minlength=self.N))
        self.batch = None
        if batch is not None:
            self.set_batch(batch, num_graphs)

    def set_batch(self, batch, num_graphs=None):
        self.batch = _i32(batch)
        if num_graphs is None:
            num_graphs = int(batch[-1].item()) + 1 if batch.numel() else 0
        self.num_graphs = int(num_graphs)
        self.mol_ptr = _ptr_from_counts(torch.bincount(batch.to(torch.int64), minlength=self.num_graphs))

    @staticmethod
    def from_radius(pos, batch, r, max_num_neighbors=1000, num_graphs=None):
        """Radius graph per molecule (nodes of a molecule contiguous, `batch` ascending)."""
        if not pos.is_cuda:
            raise ops.HipOnlyError("radius graph construction runs on the GPU only")
        pos = pos.detach().to(torch.float32).contiguous()
        N = pos.shape[0]
        if num_graphs is None:
            num_graphs = int(batch[-1].item()) + 1
        mol_ptr = _ptr_from_counts(torch.bincount(batch.to(torch.int64), minlength=num_graphs))
        deg = torch.empty(N, dtype=torch.int32, device=pos.device)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        call("eqf_radius_graph_count", P(pos), P(mol_ptr), num_graphs, float(r), int(max_num_neighbors), P(deg), st)
        row_ptr = _ptr_from_counts(deg)
        E = int(row_ptr[-1].item())  # the one host sync of graph construction (the reference syncs here as well)
        src = torch.empty(E, dtype=torch.int32, device=pos.device)
        dst = torch.empty(E, dtype=torch.int32, device=pos.device)
        call("eqf_radius_graph_fill", P(pos), P(mol_ptr), num_graphs, float(r), int(max_num_neighbors), P(row_ptr),
             P(src), P(dst), st)
        g = EdgeGraph(N, src, dst, row_ptr)
        g.batch = _i32(batch)
        g.num_graphs = int(num_graphs)
        g.mol_ptr = mol_ptr
        return g

    @staticmethod
    def from_edges(edge_src, edge_dst, N, batch=None, num_graphs=None):
        """Arbitrary edge list (e.g. periodic-boundary edges computed upstream); sorted by dst here.
        Returns (graph, order) with order = permutation applied to the caller's per-edge data."""
        order = torch.argsort(edge_dst.to(torch.int64), stable=True)
        src = _i32(edge_src[order])
        dst = _i32(edge_dst[order])
        row_ptr = _ptr_from_counts(torch.bincount(dst.to(torch.int64), minlength=N))
        return EdgeGraph(N, src, dst, row_ptr, batch, num_graphs), order
