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
    """Layout of one feature row described by simplified irreps sorted by degree, even before odd (at most one segment
    per (l, p); the SE(3) models only have even segments, the E(3) ones interleave 0e, 0o, 1e, 1o, ...).  `segs` holds
    (mul, l) per segment, `par` the parities."""

    def __init__(self, irreps):
        irreps = Irreps(irreps)
        keys = [(ir.l, -ir.p) for _, ir in irreps]
        if keys != sorted(set(keys)):
            raise NotImplementedError("row layouts need one segment per irrep, sorted by degree (even first), got %r"
                                      % irreps)
        self.irreps = irreps
        self.segs = [(mul, ir.l) for mul, ir in irreps]
        self.par = [ir.p for _, ir in irreps]
        self.has_odd = any(p == -1 for p in self.par)
        self.offsets, off = [], 0
        for mul, l in self.segs:
            self.offsets.append(off)
            off += mul * (2 * l + 1)
        self.dim = off
        self.c = lib.make_irreps(self.segs, self.par)
        self.num_irreps = irreps.num_irreps

    @property
    def c_ref(self):  # not stored: a byref object cannot be deep-copied / pickled (ModelEma deep-copies the model)
        return ctypes.byref(self.c)

    def seg_index(self, l, p=1):
        for i, (_, ll) in enumerate(self.segs):
            if ll == l and self.par[i] == p:
                return i
        return None

    def mul_of(self, l, p=1):
        i = self.seg_index(l, p)
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
        irreps_sh = Irreps(irreps_sh)
        if [(m, ir.l) for m, ir in irreps_sh] != [(1, l) for l in range(len(irreps_sh))]:
            raise NotImplementedError("edge attributes must be the full spherical harmonics 1x0e+1x1e+...: %r" % irreps_sh)
        self.lmax_sh = len(irreps_sh) - 1
        sh_par = [ir.p for _, ir in irreps_sh]  # SE(3) models: all even; E(3) models: (-1)^l
        node_out = Irreps(irreps_node_output)
        paths = []  # creation order
        w_off = 0
        for (mul, l1), p1, in_off in zip(self.layout_in.segs, self.layout_in.par, self.layout_in.offsets):
            for l2 in range(self.lmax_sh + 1):
                for ir3 in Irrep(l1, p1).couple(Irrep(l2, sh_par[l2])):
                    if ir3 in node_out or ir3 == Irrep(0, 1):
                        paths.append(dict(l1=l1, l2=l2, l3=ir3.l, p3=ir3.p, mul=mul, in_off=in_off, w_off=w_off))
                        w_off += mul
        self.weight_numel = w_off
        # sorted output irreps (even first, stable by creation index) -> channel offsets inside each output irrep
        okey = lambda q: (q["l3"], -q["p3"])  # noqa: E731
        out_k = {}
        for p in paths:
            p["out_ch"] = out_k.get(okey(p), 0)
            out_k[okey(p)] = p["out_ch"] + p["mul"]
        self.irreps_out = Irreps([(out_k[k], Irrep(k[0], -k[1])) for k in sorted(out_k)])
        self.layout_out = RowLayout(self.irreps_out)
        # instruction list in the unsimplified, sorted form (what e3nn / the reference's state_dict sees)
        self.irreps_out_unsimplified = Irreps(
            [(p["mul"], Irrep(p["l3"], p["p3"])) for p in sorted(paths, key=lambda q: (okey(q), paths.index(q)))])
        cg_chunks, cg_off = [], 0
        for p in paths:
            i = self.layout_out.seg_index(p["l3"], p["p3"])
            p["out_off"] = self.layout_out.offsets[i]
            p["out_k"] = out_k[okey(p)]
            t = so3.path_table(p["l1"], p["l2"], p["l3"]).astype(np.float32).reshape(-1)
            p["cg_off"] = cg_off
            cg_chunks.append(t)
            cg_off += t.size
        # coupling row: degree-major (all matrices of the paths feeding l3 are contiguous), creation order inside
        m_off = 0
        for k3 in sorted(out_k):
            for p in paths:
                if okey(p) == k3:
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
        # the DTP-generating GEMMs (eqf_dtp_linear_*, eqf_sfc_*) index their per-degree tables by l3: SE(3) models only
        self.has_odd = self.layout_in.has_odd or self.layout_out.has_odd
        self.fusable = all(p["mul"] % 32 == 0 for p in paths) and not self.has_odd
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
