"""e3nn.o3 stand-in: Irrep / Irreps algebra with e3nn's public surface (the reference iterates `for mul, ir in irreps`,
reads `.mul`, `.ir`, `ir.l`, `ir.p`, `ir.dim`, `ir.is_scalar()`, parses `str(irreps)`, calls `.simplify()`, `.sort()`,
`.slices()`, `.dim`, `.num_irreps`, `.lmax`, `ir1 * ir2`, `ir in irreps`), TensorProduct with e3nn's Instruction
tuples, ElementwiseTensorProduct, spherical_harmonics.  Contractions / tables / harmonics come from oracle.e3."""
import collections
import math as _math

import torch

from oracle import e3 as _e3


class Irrep(tuple):
    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                s = l.strip()
                l, p = int(s[:-1]), {"e": 1, "o": -1, "y": None}[s[-1]]
                if p is None:
                    p = (-1) ** l
            else:
                l, p = l
        assert isinstance(l, int) and l >= 0 and p in (-1, 1), (l, p)
        return tuple.__new__(cls, (l, p))

    l = property(lambda self: self[0])
    p = property(lambda self: self[1])
    dim = property(lambda self: 2 * self[0] + 1)

    def is_scalar(self):
        return self[0] == 0 and self[1] == 1

    def __mul__(self, other):
        other = Irrep(other)
        return [Irrep(l, self.p * other.p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]

    def __repr__(self):
        return "%d%s" % (self[0], "e" if self[1] == 1 else "o")


class _MulIr(tuple):
    def __new__(cls, mul, ir):
        return tuple.__new__(cls, (int(mul), Irrep(ir)))

    mul = property(lambda self: self[0])
    ir = property(lambda self: self[1])
    dim = property(lambda self: self[0] * self[1].dim)

    def __repr__(self):
        return "%dx%r" % (self[0], self[1])


class Irreps(tuple):
    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return irreps
        out = []
        if irreps is None:
            pass
        elif isinstance(irreps, Irrep):
            out.append(_MulIr(1, irreps))
        elif isinstance(irreps, str):
            if irreps.strip():
                for tok in irreps.split("+"):
                    tok = tok.strip()
                    mul, ir = tok.split("x") if "x" in tok else (1, tok)
                    out.append(_MulIr(int(mul), Irrep(ir.strip())))
        else:
            for item in irreps:
                if isinstance(item, (Irrep, str)):
                    out.append(_MulIr(1, Irrep(item)))
                else:
                    mul, ir = item
                    out.append(_MulIr(mul, Irrep(ir)))
        return tuple.__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, (l, p ** l)) for l in range(lmax + 1)])

    dim = property(lambda self: sum(mi.dim for mi in self))
    num_irreps = property(lambda self: sum(mi.mul for mi in self))
    ls = property(lambda self: [mi.ir.l for mi in self for _ in range(mi.mul)])

    @property
    def lmax(self):
        if len(self) == 0:
            raise ValueError("Cannot get lmax of empty Irreps")
        return max(mi.ir.l for mi in self)

    def slices(self):
        s, i = [], 0
        for mi in self:
            s.append(slice(i, i + mi.dim))
            i += mi.dim
        return s

    def simplify(self):
        out = []
        for mul, ir in self:
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + mul, ir)
            elif mul > 0:
                out.append((mul, ir))
        return Irreps(out)

    def remove_zero_multiplicities(self):
        return Irreps([(mul, ir) for mul, ir in self if mul > 0])

    def sort(self):
        """e3nn: sorted by the Irrep TUPLE (l, p) -- odd before even within a degree -- then creation index."""
        Ret = collections.namedtuple("sort", ["irreps", "p", "inv"])
        out = sorted((tuple(ir), i, mul) for i, (mul, ir) in enumerate(self))
        inv = tuple(i for _, i, _ in out)
        p = [0] * len(inv)
        for new, old in enumerate(inv):
            p[old] = new
        return Ret(Irreps([(mul, ir) for ir, _, mul in out]), tuple(p), inv)

    def count(self, ir):
        ir = Irrep(ir)
        return sum(mul for mul, i2 in self if i2 == ir)

    def randn(self, *size, **kw):
        lead = [s for s in size if s != -1]
        return torch.randn(*lead, self.dim, **kw)

    def __contains__(self, ir):
        ir = Irrep(ir)
        return any(ir == i2 for _, i2 in self)

    def __getitem__(self, i):
        x = tuple.__getitem__(self, i)
        return Irreps(x) if isinstance(i, slice) else x

    def __add__(self, other):
        return Irreps(tuple.__add__(self, Irreps(other)))

    def __mul__(self, n):
        return Irreps(tuple.__mul__(self, n))

    __rmul__ = __mul__

    def __repr__(self):
        return "+".join(repr(mi) for mi in self)


Instruction = collections.namedtuple("Instruction", "i_in1 i_in2 i_out connection_mode has_weight path_weight")


class TensorProduct(_e3.TensorProduct):
    """o3.TensorProduct(..., path_normalization='none', irrep_normalization='component' [the default behind
    normalization=None]): the contraction itself is oracle.e3.TensorProduct (out += sqrt(2 l_out + 1) * path_weight *
    w * C_ijk x1 x2, flat weights in instruction order); this class only adds e3nn's attribute surface."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, in1_var=None, in2_var=None, out_var=None,
                 irrep_normalization=None, path_normalization=None, internal_weights=None, shared_weights=None,
                 normalization=None, **kw):
        assert path_normalization == "none", "the reference only builds path_normalization='none' products"
        assert (irrep_normalization or normalization or "component") == "component"
        assert not kw, kw
        ins = [Instruction(*(tuple(x) + (1.0,))[:6]) for x in instructions]
        super().__init__(Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out), ins,
                         internal_weights=internal_weights, shared_weights=shared_weights)
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        self.instructions = ins

    def weight_views(self, weight=None, yield_instruction=False):
        views = super().weight_views(weight)
        if not yield_instruction:
            return views
        have = [k for k, i in enumerate(self.instructions) if i.has_weight]
        return [(k, self.instructions[k], v) for k, v in zip(have, views)]


class ElementwiseTensorProduct(TensorProduct):
    """e3nn/o3/_tensor_product/_sub.py: channel-wise ('uuu', no weights) product of two inputs with the same number of
    irreps; multiplicities are aligned by splitting (reference call sites: nets/fast_activation.py:122, nets/drop.py:75)."""

    def __init__(self, irreps_in1, irreps_in2, filter_ir_out=None, irrep_normalization=None, **kw):
        a, b = list(Irreps(irreps_in1).simplify()), list(Irreps(irreps_in2).simplify())
        assert sum(m for m, _ in a) == sum(m for m, _ in b), (irreps_in1, irreps_in2)
        i = 0
        while i < len(a):
            (m1, ir1), (m2, ir2) = a[i], b[i]
            if m1 < m2:
                b[i] = (m1, ir2)
                b.insert(i + 1, (m2 - m1, ir2))
            if m2 < m1:
                a[i] = (m2, ir1)
                a.insert(i + 1, (m1 - m2, ir1))
            i += 1
        out, ins = [], []
        if filter_ir_out is not None:
            filter_ir_out = [Irrep(ir) for ir in filter_ir_out]
        for i, ((mul, ir1), (mul2, ir2)) in enumerate(zip(a, b)):
            assert mul == mul2
            for ir in ir1 * ir2:
                if filter_ir_out is not None and ir not in filter_ir_out:
                    continue
                ins.append((i, i, len(out), "uuu", False))
                out.append((mul, ir))
        super().__init__(Irreps(a), Irreps(b), Irreps(out), ins, path_normalization="none",
                         irrep_normalization=irrep_normalization, **kw)


class FullyConnectedTensorProduct(TensorProduct):
    """Only built by a __main__ print-check of the reference (nets/tensor_product_rescale.py:234-291); not on the path.
    e3nn's default path normalisation ('element') is not restated."""

    def __init__(self, *a, **kw):
        raise NotImplementedError("FullyConnectedTensorProduct with e3nn's default path normalisation is not restated")


def spherical_harmonics(l, x, normalize, normalization="integral"):
    """e3nn.o3.spherical_harmonics; the reference always passes an Irreps 1x0e+1x1?+...+1xL? covering every degree
    0..L once, normalize=True, normalization='component' (nets/graph_attention_transformer.py:869-870)."""
    if isinstance(l, int):
        ls = [l]
    elif isinstance(l, (str, Irreps)):
        ls = Irreps(l).ls
    else:
        ls = list(l)
    full = _e3.spherical_harmonics(max(ls), x, normalize=normalize, normalization=normalization)
    if ls == list(range(max(ls) + 1)):
        return full
    return torch.cat([full[..., d * d:(d + 1) * (d + 1)] for d in ls], dim=-1)


def wigner_3j(l1, l2, l3, dtype=None, device=None):
    return _e3.wigner_3j(l1, l2, l3).to(dtype=dtype or torch.get_default_dtype(), device=device)
