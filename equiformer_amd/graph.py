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


def _P(t, byte_offset=0):
    return ctypes.c_void_p(ops._nonnull(t) + byte_offset) if t is not None else None


def _stream():
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return ctypes.c_void_p(raw(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


CSR_MAX_NODES = 16384  # nodes per molecule the by-source kernel keeps cursors for (csrc/graph.hip)


class EdgeGraph:
    """N nodes, E directed edges sorted by dst.  All index tensors are int32 on the GPU."""

    def __init__(self, N, src, dst, row_ptr, batch=None, num_graphs=None, mol_ptr=None, max_mol_nodes=None, by_source_into=None):
        self.N = int(N)
        self.src, self.dst, self.row_ptr = src, dst, row_ptr
        self.E = int(src.shape[0])
        # by-source view: edges grouped by src (for gradients that flow back to the source node)
        if mol_ptr is not None and max_mol_nodes is not None and max_mol_nodes <= CSR_MAX_NODES and src.is_cuda:
            # radius graphs: molecule-blocked, no multi-edges -> one HIP launch instead of a device sort
            if by_source_into is not None:  # (a captured step's static graph: the same buffers every step)
                self.src_perm, self.src_ptr = by_source_into
            else:
                self.src_perm = torch.empty(self.E, dtype=torch.int32, device=src.device)
                self.src_ptr = torch.empty(self.N + 1, dtype=torch.int32, device=src.device)
            call("eqf_csr_by_source", _P(src), _P(row_ptr), _P(mol_ptr), int(mol_ptr.shape[0]) - 1, int(max_mol_nodes),
                 _P(self.src_perm), _P(self.src_ptr), _stream())
        else:
            # arbitrary edge lists (periodic images can repeat a source inside a row): stable device sort
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
    def from_radius(pos, batch, r, max_num_neighbors=1000, num_graphs=None, into=None):
        """Radius graph per molecule (nodes of a molecule contiguous, `batch` ascending).  into: a graph of an earlier call; when
        this call finds the same node and edge counts its index tensors are REWRITTEN in place and `into` is returned -- the
        launches of a HIP-graph-captured step (equiformer_amd/capture.py) read those addresses."""
        if not pos.is_cuda:
            raise ops.HipOnlyError("radius graph construction runs on the GPU only")
        pos = pos.detach().to(torch.float32).contiguous()
        N = pos.shape[0]
        if num_graphs is None:
            num_graphs = int(batch[-1].item()) + 1
        dev = pos.device
        b32 = _i32(batch)
        st = _stream()
        stats = torch.empty(2, dtype=torch.int32, device=dev)  # [E, nodes of the largest molecule]
        mol_ptr = torch.empty(num_graphs + 1, dtype=torch.int32, device=dev)
        call("eqf_segment_ptr", _P(b32), N, int(num_graphs), _P(mol_ptr), _P(stats, 4), st)
        deg = torch.empty(N, dtype=torch.int32, device=dev)
        call("eqf_radius_graph_count", _P(pos), _P(mol_ptr), num_graphs, float(r), int(max_num_neighbors), _P(deg), st)
        reuse = (into is not None and into.N == N and into.num_graphs == int(num_graphs) and into.src.device == dev
                 and getattr(into, "_radius_static", False))
        # (a candidate for in-place reuse gets the scan written straight into its row_ptr: if the edge count then differs the graph
        # is abandoned by its owner anyway -- equiformer_amd/capture.py falls back to an eager step on a fresh graph)
        row_ptr = into.row_ptr if reuse else torch.empty(N + 1, dtype=torch.int32, device=dev)
        call("eqf_exclusive_scan_i32", _P(deg), N, _P(row_ptr), _P(stats), st)
        E, max_mol_nodes = stats.tolist()  # the one host sync of graph construction (the reference syncs here as well)
        if reuse and into.E != E:
            row_ptr = row_ptr.clone()  # (the caller's tensors must not alias the abandoned graph's)
            into._radius_static = False
        if reuse and into.E == E:
            into.mol_ptr.copy_(mol_ptr)
            if into.batch.data_ptr() != b32.data_ptr():
                into.batch.copy_(b32)
            call("eqf_radius_graph_fill", _P(pos), _P(into.mol_ptr), num_graphs, float(r), int(max_num_neighbors),
                 _P(into.row_ptr), _P(into.src), _P(into.dst), st)
            call("eqf_csr_by_source", _P(into.src), _P(into.row_ptr), _P(into.mol_ptr), int(num_graphs), int(max_mol_nodes),
                 _P(into.src_perm), _P(into.src_ptr), st)
            return into
        src = torch.empty(E, dtype=torch.int32, device=dev)
        dst = torch.empty(E, dtype=torch.int32, device=dev)
        call("eqf_radius_graph_fill", _P(pos), _P(mol_ptr), num_graphs, float(r), int(max_num_neighbors), _P(row_ptr),
             _P(src), _P(dst), st)
        g = EdgeGraph(N, src, dst, row_ptr, mol_ptr=mol_ptr, max_mol_nodes=max_mol_nodes)
        g._radius_static = max_mol_nodes <= CSR_MAX_NODES  # (its by-source view came from eqf_csr_by_source: refillable in place)
        g.batch = b32
        g.num_graphs = int(num_graphs)
        g.mol_ptr = mol_ptr
        return g

    @staticmethod
    def from_radius_pbc(pos, cell, batch, r, max_num_neighbors=50, num_graphs=None):
        """Periodic radius graph (ocpmodels radius_graph_pbc + get_pbc_distances semantics, see csrc/graph.hip).
        Returns (graph, offsets[E,3] Cartesian, cell_offsets[E,3] int32); edges are dst-sorted."""
        if not pos.is_cuda:
            raise ops.HipOnlyError("radius graph construction runs on the GPU only")
        pos = pos.detach().to(torch.float32).contiguous()
        cell = cell.detach().to(torch.float32).contiguous().view(-1, 3, 3)
        N = pos.shape[0]
        if num_graphs is None:
            num_graphs = int(cell.shape[0])
        dev = pos.device
        b32 = _i32(batch)
        st = _stream()
        stats = torch.empty(3, dtype=torch.int32, device=dev)  # [E, nodes of the largest structure, candidates]
        mol_ptr = torch.empty(num_graphs + 1, dtype=torch.int32, device=dev)
        call("eqf_segment_ptr", _P(b32), N, int(num_graphs), _P(mol_ptr), _P(stats, 4), st)
        cand = torch.empty(N, dtype=torch.int32, device=dev)
        deg = torch.empty(N, dtype=torch.int32, device=dev)
        call("eqf_radius_graph_pbc_count", _P(pos), _P(cell), _P(mol_ptr), num_graphs, float(r), int(max_num_neighbors),
             _P(cand), _P(deg), st)
        row_ptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        cand_ptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        call("eqf_exclusive_scan_i32", _P(deg), N, _P(row_ptr), _P(stats), st)
        call("eqf_exclusive_scan_i32", _P(cand), N, _P(cand_ptr), _P(stats, 8), st)
        E, _, C = stats.tolist()  # the one host sync
        src = torch.empty(E, dtype=torch.int32, device=dev)
        dst = torch.empty(E, dtype=torch.int32, device=dev)
        cell_offsets = torch.empty((E, 3), dtype=torch.int32, device=dev)
        offsets = torch.empty((E, 3), dtype=torch.float32, device=dev)
        scratch = torch.empty(max(C, 1), dtype=torch.float32, device=dev)
        call("eqf_radius_graph_pbc_fill", _P(pos), _P(cell), _P(mol_ptr), num_graphs, float(r), int(max_num_neighbors),
             _P(row_ptr), _P(cand_ptr), _P(scratch), _P(src), _P(dst), _P(cell_offsets), _P(offsets), st)
        # a source can occur several times in a row (different images): generic by-source path
        g = EdgeGraph(N, src, dst, row_ptr)
        g.batch = b32
        g.num_graphs = int(num_graphs)
        g.mol_ptr = mol_ptr
        return g, offsets, cell_offsets

    @staticmethod
    def from_edges(edge_src, edge_dst, N, batch=None, num_graphs=None):
        """Arbitrary edge list (e.g. periodic-boundary edges computed upstream); sorted by dst here.
        Returns (graph, order) with order = permutation applied to the caller's per-edge data."""
        order = torch.argsort(edge_dst.to(torch.int64), stable=True)
        src = _i32(edge_src[order])
        dst = _i32(edge_dst[order])
        row_ptr = _ptr_from_counts(torch.bincount(dst.to(torch.int64), minlength=N))
        return EdgeGraph(N, src, dst, row_ptr, batch, num_graphs), order
