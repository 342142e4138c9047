"""Row layouts and depth-wise tensor-product path tables handed to the HIP library.

Internal feature layout ("CF", channel-fastest): a row holds its irreps segments one after the other and the
segment of degree l is stored as [2l+1][mul].  e3nn (and therefore the reference's tensors) use [mul][2l+1];
the two only meet at the model boundary, where every tensor is scalar (0e), so no conversion is ever executed in
the hot path -- `perm_from_e3nn` exists for tests and for users who want to inspect intermediate features.
"""
import ctypes

import numpy as np
import torch

from . import lib, so3
from .irreps import Irrep, Irreps


class RowLayout:
    """Layout of one feature row described by simplified, degree-sorted, even irreps (at most one segment per l)."""

    def __init__(self, irreps):
        irreps = Irreps(irreps).require_even()
        ls = [ir.l for _, ir in irreps]
        if ls != sorted(set(ls)):
            raise NotImplementedError("row layouts need one segment per degree in ascending order, got %r" % irreps)
        self.irreps = irreps
        self.segs = [(mul, ir.l) for mul, ir in irreps]
        self.offsets, off = [], 0
        for mul, l in self.segs:
            self.offsets.append(off)
            off += mul * (2 * l + 1)
        self.dim = off
        self.c = lib.make_irreps(self.segs)
        self.num_irreps = irreps.num_irreps

    @property
    def c_ref(self):  # not stored: a byref object cannot be deep-copied / pickled (ModelEma deep-copies the model)
        return ctypes.byref(self.c)

    def seg_index(self, l):
        for i, (_, ll) in enumerate(self.segs):
            if ll == l:
                return i
        return None

    def mul_of(self, l):
        i = self.seg_index(l)
        return 0 if i is None else self.segs[i][0]

    def perm_from_e3nn(self):
        """idx with x_cf = x_e3nn[..., idx]."""
        idx = np.empty(self.dim, dtype=np.int64)
        for (mul, l), off in zip(self.segs, self.offsets):
            d = 2 * l + 1
            for m in range(d):
                for u in range(mul):
                    idx[off + m * mul + u] = off + u * d + m
        return torch.from_numpy(idx)

    def perm_to_e3nn(self):
        """idx with x_e3nn = x_cf[..., idx]."""
        p = self.perm_from_e3nn()
        inv = torch.empty_like(p)
        inv[p] = torch.arange(p.numel())
        return inv

    def __repr__(self):
        return "RowLayout(%r)" % (self.irreps,)


class DtpTable:
    """Path table of DepthwiseTensorProduct(irreps_in, irreps_sh, irreps_node_output)
    [ref: nets/graph_attention_transformer.py:157-183]: 'uvu' paths (i, j, k) for every output degree that is in
    irreps_node_output or is 0e, weights in creation order, output segments sorted by degree (stable)."""

    def __init__(self, irreps_in, irreps_sh, irreps_node_output):
        self.layout_in = RowLayout(Irreps(irreps_in).simplify())
        irreps_sh = Irreps(irreps_sh).require_even()
        if [(m, ir.l) for m, ir in irreps_sh] != [(1, l) for l in range(len(irreps_sh))]:
            raise NotImplementedError("edge attributes must be the full spherical harmonics 1x0e+1x1e+...: %r" % irreps_sh)
        self.lmax_sh = len(irreps_sh) - 1
        node_out = Irreps(irreps_node_output).require_even()
        paths = []  # creation order
        w_off = 0
        for (mul, l1), in_off in zip(self.layout_in.segs, self.layout_in.offsets):
            for l2 in range(self.lmax_sh + 1):
                for ir3 in Irrep(l1, 1).couple(Irrep(l2, 1)):
                    if ir3 in node_out or ir3.l == 0:
                        paths.append(dict(l1=l1, l2=l2, l3=ir3.l, mul=mul, in_off=in_off, w_off=w_off))
                        w_off += mul
        self.weight_numel = w_off
        # sorted output irreps (stable by creation index) -> channel offsets inside each output degree
        out_k = {}
        for p in paths:
            p["out_ch"] = out_k.get(p["l3"], 0)
            out_k[p["l3"]] = p["out_ch"] + p["mul"]
        self.irreps_out = Irreps([(out_k[l], Irrep(l, 1)) for l in sorted(out_k)])
        self.layout_out = RowLayout(self.irreps_out)
        # instruction list in the unsimplified, sorted form (what e3nn / the reference's state_dict sees)
        self.irreps_out_unsimplified = Irreps(
            [(p["mul"], Irrep(p["l3"], 1)) for p in sorted(paths, key=lambda q: (q["l3"], paths.index(q)))])
        cg_chunks, cg_off = [], 0
        for p in paths:
            i = self.layout_out.seg_index(p["l3"])
            p["out_off"] = self.layout_out.offsets[i]
            p["out_k"] = out_k[p["l3"]]
            t = so3.path_table(p["l1"], p["l2"], p["l3"]).astype(np.float32).reshape(-1)
            p["cg_off"] = cg_off
            cg_chunks.append(t)
            cg_off += t.size
        # coupling row: degree-major (all matrices of the paths feeding l3 are contiguous), creation order inside
        m_off = 0
        for l3 in sorted(out_k):
            for p in paths:
                if p["l3"] == l3:
                    p["m_off"] = m_off
                    m_off += (2 * p["l1"] + 1) * (2 * p["l3"] + 1)
        self.paths = paths
        self.m_numel = m_off
        self.cg_host = torch.from_numpy(np.concatenate(cg_chunks))
        if len(paths) > lib.EQF_MAX_PATHS:
            raise NotImplementedError("too many DTP paths")
        c = lib.EqfDtpPaths()
        c.npaths = len(paths)
        c.sh_dim = (self.lmax_sh + 1) ** 2
        c.in_dim = self.layout_in.dim
        c.out_dim = self.layout_out.dim
        c.w_numel = self.weight_numel
        c.m_numel = self.m_numel
        for i, p in enumerate(paths):
            for k in ("l1", "l2", "l3", "mul", "in_off", "out_off", "out_ch", "out_k", "w_off", "cg_off", "m_off"):
                getattr(c, k)[i] = p[k]
        self.c = c
        self._cg_dev = {}
        self.key = (repr(self.layout_in.irreps), self.lmax_sh, repr(self.irreps_out))
        self.fusable = all(p["mul"] % 32 == 0 for p in paths)
        # every input segment is read by at least one path (then the backward writes all of dx)
        self.in_covered = {p["in_off"] for p in paths} == set(self.layout_in.offsets)

    @property
    def c_ref(self):
        return ctypes.byref(self.c)

    def cg(self, device):
        t = self._cg_dev.get(device)
        if t is None:
            t = self.cg_host.to(device)
            self._cg_dev[device] = t
        return t
