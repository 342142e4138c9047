"""TEST INFRASTRUCTURE (oracle): CPU restatement of the periodic neighbour search the OC20 model calls.

The reference calls `radius_graph_pbc(data, max_radius, max_neighbors)` and `get_pbc_distances(pos, edge_index, cell,
cell_offsets, neighbors, return_offsets=True)` (nets/graph_attention_transformer_oc20.py:267-293).  Both live in the
un-vendored dependency ocpmodels 0.0.3 @ d2aaaeb (docs/env_setup.md:18-26 of the reference), file
ocpmodels/common/utils.py, which is absent from /root/reference and cannot be installed here.  Dependency restatement (ocpmodels itself has never run here):
what follows restates that library's published algorithm; it is anchored on the reference's own call sites (argument
meaning, the edge_vec = pos[src] - pos[dst] + offsets convention of :290-291) and on the known-answer tests in
tests/test_oracle_kat.py (coordination numbers of simple lattices).

Algorithm (ocpmodels.common.utils.radius_graph_pbc):
  * candidate pairs: every ordered pair (centre i, neighbour j) of atoms of the same structure, including i == j;
  * periodic images: n = (n1, n2, n3), |n_k| <= rep_k, rep_k = ceil(radius * |a_l x a_m| / |det cell|)  (the number
    of cells needed along a_k to cover the cut-off sphere); the library uses the batch maximum of rep_k for every
    structure, which only adds images that the distance test rejects;
  * keep pairs with 1e-4 < |pos_j + n . cell - pos_i|^2 <= radius^2   (<=, unlike torch_cluster's strict <);
  * per centre keep the `max_neighbors` nearest (get_max_neighbors_mask: sort by distance, ties in candidate order);
  * returns edge_index = [neighbour j, centre i], cell_offsets n (integers), neighbours per structure.
get_pbc_distances: offsets = cell_offsets @ cell (Cartesian), distance vector = pos[j] - pos[i] + offsets, edges of
zero length dropped.
"""
import math

import torch


def cell_repeats(cell, radius):
    """rep_k of one [3,3] cell (rows are the lattice vectors a1, a2, a3)."""
    a1, a2, a3 = cell[0].double(), cell[1].double(), cell[2].double()
    c23, c31, c12 = torch.linalg.cross(a2, a3), torch.linalg.cross(a3, a1), torch.linalg.cross(a1, a2)
    vol = torch.dot(a1, c23)
    reps = []
    for c in (c23, c31, c12):
        reps.append(int(math.ceil(radius * float(torch.linalg.norm(c / vol)) - 1e-12)))
    return reps


def radius_graph_pbc(pos, cell, natoms, radius, max_neighbors):
    """pos [N,3], cell [B,3,3], natoms [B] -> (edge_index [2,E] = (neighbour, centre), cell_offsets [E,3] int64,
    neighbors [B]); edges grouped by centre (ascending), then neighbour, then image (n1, n2, n3 lexicographic)."""
    src, dst, offs, per_structure = [], [], [], []
    start = 0
    r2 = float(radius) * float(radius)
    for b, n in enumerate(int(v) for v in natoms):
        p = pos[start:start + n].double()
        c = cell[b].double()
        r1, r2_, r3 = cell_repeats(c, radius)
        imgs = torch.tensor([(i, j, k) for i in range(-r1, r1 + 1) for j in range(-r2_, r2_ + 1)
                             for k in range(-r3, r3 + 1)], dtype=torch.float64)
        shift = imgs @ c  # [I,3]
        # d[i, j, I] = |p_j + shift_I - p_i|^2
        d = (p[None, :, None, :] + shift[None, None, :, :] - p[:, None, None, :]).pow(2).sum(-1)
        keep = (d <= r2) & (d > 1e-4)
        count = 0
        for i in range(n):
            jj, ii = keep[i].nonzero(as_tuple=True)
            if jj.numel() > max_neighbors:
                order = torch.sort(d[i][jj, ii], stable=True).indices[:max_neighbors]
                order = torch.sort(order).values  # keep candidate order among the survivors
                jj, ii = jj[order], ii[order]
            src.append(jj + start)
            dst.append(torch.full_like(jj, i + start))
            offs.append(imgs[ii].long())
            count += jj.numel()
        per_structure.append(count)
        start += n
    edge_index = torch.stack([torch.cat(src), torch.cat(dst)])
    return edge_index, torch.cat(offs), torch.tensor(per_structure, dtype=torch.long)


def get_pbc_distances(pos, edge_index, cell, cell_offsets, neighbors):
    """-> (edge_index, distances, Cartesian offsets) with zero-length edges dropped."""
    row, col = edge_index
    cell_per_edge = torch.repeat_interleave(cell, neighbors, dim=0).to(pos.dtype)
    offsets = torch.bmm(cell_offsets.to(pos.dtype).view(-1, 1, 3), cell_per_edge).view(-1, 3)
    vec = pos[row] - pos[col] + offsets
    dist = vec.norm(dim=-1)
    nz = dist != 0
    return edge_index[:, nz], dist[nz], offsets[nz]
